#!/usr/bin/env python
"""Measurement tool (GPU box): syn1 (400 targets) split by row-block count; each group alone (resident and streaming)
and the whole batch, single launches with a host sync in between."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import bench
from gnn_model_explainer_amd.engine import MaskOptimJob, Hyper, device_graph

wl = bench.Workload("syn1")
wl.prepare()
graph = device_graph(wl.idx.csr, wl.feat, wl.pred)
nb = np.asarray([(len(x) + 31) // 32 for x in wl.nbs])
label = wl.label[np.asarray(wl.targets)]


def timed(sel, back_to_back=1, **kw):
    sel = np.flatnonzero(sel)
    job = MaskOptimJob.from_csr(graph, [wl.nbs[k] for k in sel], [wl.rows[k] for k in sel], label[sel], wl.ck["sd"])
    hy = Hyper(num_iters=300, use_graph=True, **kw)
    job.set_masks([wl.masks[k] for k in sel])
    M0 = job.M.clone()
    job.launch(hy)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(back_to_back):
            job.M.copy_(M0)
            job.launch(hy)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / back_to_back)
    return best * 1e3


for name, sel in (("nb=1", nb == 1), ("nb=2", nb == 2), ("nb=3", nb == 3), ("nb>=4", nb >= 4), ("nb<=3", nb <= 3), ("all", nb >= 1)):
    print(f"{name:6s} count={int(sel.sum()):3d}  default: {timed(sel):6.2f} ms   streaming: {timed(sel, use_resident=False):6.2f} ms"
          f"   default x5 back-to-back: {timed(sel, back_to_back=5):6.2f} ms/step", flush=True)
