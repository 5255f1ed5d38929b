#!/usr/bin/env python
"""Measurement tool (GPU box): `--method att` on k_att - all 400 syn1 motif nodes as one batch (the attention encoder of the golden
fixture), 300 iterations: wall time of the run, and the same explanation for two nodes on the PyTorch-ROCm route for scale."""
import os, sys, time, argparse
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import helpers
from gnn_model_explainer_amd import engine, models
from gnn_model_explainer_amd.engine import MaskOptimJob
from gnn_model_explainer_amd.explainer import explain, torch_route
Z = np.load(os.path.join(helpers.GOLDEN, "options_explain.npz"))
ck = helpers.load_ckpt("syn1")
sd = {k[len("route:att:w:"):]: Z[k] for k in Z.files if k.startswith("route:att:w:")}
from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
pred = Z["route:att:pred"]
graph = engine.device_graph(idx.csr, ck["feat"], pred)
targets = np.arange(300, 700, dtype=np.int64)
args = argparse.Namespace(lr=0.1, opt="adam", opt_scheduler="none", num_epochs=300, method="att", bias=True, num_gc_layers=3, mask_act="sigmoid")
hy = explain._hyper(args, edge_results_only=True)
for rep in range(3):
    t0 = time.perf_counter()
    dn = engine.khop_device(graph, targets, 3)
    job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], sd)
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets, pin=True))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    job.launch(hy); torch.cuda.synchronize(); t2 = time.perf_counter()
    em = job.fetch_edges(); t3 = time.perf_counter()
    print(f"k_att, 400 targets x 300 iterations: prepare {1e3 * (t1 - t0):.1f} ms, optimise {1e3 * (t2 - t1):.1f} ms, fetch {1e3 * (t3 - t2):.1f} ms "
          f"-> {400 / (t3 - t0):.0f} nodes/s end to end", flush=True)
    job.close()
model = models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args).cuda()
model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
for t in (300, 302):
    nb = idx.neighbors(t) if hasattr(idx, "neighbors") else idx.neighbors_batch(np.asarray([t]))[0]
    A = idx.sub_adjacency(nb); X = ck["feat"][nb]; new = int(np.searchsorted(nb, t))
    t0 = time.perf_counter()
    torch_route.explain_one(model, A, X, ck["label"][nb], np.argmax(pred[nb], 1), new, args, explain.COEFFS, explain._torch_optimizer if False else (lambda a, p: (None, torch.optim.Adam(p, lr=0.1))),
                            helpers.seeded_mask0(t, len(nb)).numpy(), device=torch.device("cuda"), reason="probe")
    torch.cuda.synchronize()
    print(f"PyTorch-ROCm route, node {t} (n = {len(nb)}), 300 iterations: {time.perf_counter() - t0:.2f} s")
