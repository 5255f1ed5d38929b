#!/usr/bin/env python
"""Round 6 check (GPU box): the one-dictionary-row shortcut of k_sparse_large (constant feature rows) must not change a bit.  Runs a BA-House x100k
XL batch + a route-7 batch with the library given by GNNX_LIBRARY_PATH (or the shipped one) and writes the edge results; called twice, compared by --compare."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    ok = all(np.array_equal(a[k], b[k]) for k in a.files) and sorted(a.files) == sorted(b.files)
    print("one-row shortcut vs previous library:", "BIT-IDENTICAL" if ok else "DIFFERENT", {k: a[k].shape for k in a.files})
    sys.exit(0 if ok else 1)
import torch, helpers
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper
from gnn_model_explainer_amd.pipeline import BatchPipeline
wl = bench.Workload("ba100k", 64)
ck = wl.ck
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
N = int(wl.idx.csr.shape[0])
rng = np.random.default_rng(3)
sizes = None
hy = Hyper(num_iters=60, edge_results_only=True)
out = {}
# targets of all sizes: a few BA nodes (large sub-graphs -> XL route) and motif nodes (resident / route 7)
cands = np.concatenate([rng.choice(np.arange(0, 42857), 24, replace=False), rng.choice(np.arange(42857, N), 40, replace=False)]).astype(np.int64)
for xl_min, tag in ((512, "xl"), (10 ** 9, "dense")):
    os.environ["GNNX_XL_MIN_N"] = str(xl_min)
    keep = cands if tag == "xl" else cands[24:]
    pipe = BatchPipeline(graph, ck["sd"], wl.label, hy, prepare_workers=1)
    em = list(pipe.run([keep]))[0]
    out[tag + "_vals"] = np.asarray(em.masked_adj); out[tag + "_rc"] = np.asarray(em.rc); out[tag + "_feat"] = np.asarray(em.feat_mask)
    print(tag, "routes", em.routes, "edges", len(em.masked_adj), "n max", int(em.n.max()))
np.savez(sys.argv[1], **out)
