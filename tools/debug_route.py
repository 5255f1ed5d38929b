import os, sys
os.environ["GNNX_DEBUG_ROUTE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench
from gnn_model_explainer_amd import engine
wl = bench.Workload("ba100k", 16384)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
dn = engine.khop_device(graph, wl.targets, 3)
job = engine.MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"])
r = job.route()
print("routes", dict(zip(*np.unique(r, return_counts=True))))
big = np.nonzero(r == 7)[0]
print("k_sparse_large sizes: max n", dn.sizes[big].max(), "p50", np.percentile(dn.sizes[big], 50))
