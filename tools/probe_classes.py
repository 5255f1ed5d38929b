#!/usr/bin/env python
"""Measurement tool (GPU box): the BA-House x100k target set split by sub-graph size, each part optimised as its own batch
(300 iterations) - which size class the batch time comes from, and how far each is from its one-target latency."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob

wl = bench.Workload("ba100k", int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
sz = np.asarray(wl.idx.sizes(wl.targets))
bins = [(0, 8), (8, 16), (16, 32), (0, 32), (32, 64), (64, 128), (128, 256), (256, 512), (32, 512), (0, 512), (512, 100000), (0, 100000)]
for lo, hi in bins:
    sel = wl.targets[(sz > lo) & (sz <= hi)]
    if len(sel) == 0:
        continue
    dn = engine.khop_device(graph, sel, 3)
    job = MaskOptimJob.from_csr(graph, dn, None, wl.label[sel], wl.ck["sd"])
    route = np.asarray(job.route())
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + sel, threads=4))
    hy = Hyper(num_iters=300)
    job.launch(hy)
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        job.set_masks_raw_resident()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.launch(hy)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    r, c = np.unique(route, return_counts=True)
    print("n in (%5d, %6d]: %6d targets  routes %s  batch %8.2f ms  -> %.1f us per target" %
          (lo, hi, len(sel), dict(zip(r.tolist(), c.tolist())), min(ts), min(ts) * 1e3 / len(sel)), flush=True)
    del job
