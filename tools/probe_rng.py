#!/usr/bin/env python
"""Measurement tool (GPU box host): how the seeded initial-mask generation scales with host threads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from gnn_model_explainer_amd import engine
for name, T in (("syn1", 0), ("ba100k", 2048)):
    wl = bench.Workload(name, T) if T else bench.Workload(name)
    sz = np.asarray(wl.idx.sizes(wl.targets))
    print(name, len(sz), "targets", "%.3g normals" % float((sz.astype(float) ** 2).sum()), "cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
    for th in (1, 2, 4, 8, 16, 32):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            engine.init_edge_masks_raw(sz, seeds=1000 + wl.targets, pin=True, threads=th)
            ts.append(time.perf_counter() - t0)
        print("  threads %2d: %.2f ms (median %.2f)" % (th, min(ts) * 1e3, sorted(ts)[2] * 1e3), flush=True)
