#!/usr/bin/env python
"""Measurement tool (GPU box host, round 4): the seeded initial masks of the 16 384-target BA-House x100k set (1.0e9 normals, the largest
target 31 M) - host threads, slice length of the large targets (gnnx_host_draw_masks_sliced), pinned vs pageable destination."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from gnn_model_explainer_amd import engine
wl = bench.Workload("ba100k", 16384)
sz = np.asarray(wl.idx.sizes(wl.targets))
total = int((sz.astype(np.int64) ** 2).sum())
print(len(sz), "targets", "%.3g normals" % total, "largest", int(sz.max()), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
bufs = {"pinned": torch.empty(total, dtype=torch.float32, pin_memory=True), "pageable": torch.empty(total, dtype=torch.float32)}
for b in bufs.values():
    b.zero_()       # first touch before anything is timed
for kind, buf in bufs.items():
    for th in (32, 64, 96, 128):
        for sl in (1 << 40, 1 << 21, 1 << 19):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                engine.init_edge_masks_raw(sz, seeds=1000 + wl.targets, threads=th, out=buf, slice_values=sl)
                ts.append(time.perf_counter() - t0)
            print("  %-8s threads %3d slice %-13s: %.1f ms (median %.1f)" % (kind, th, "off" if sl > 1 << 30 else sl, min(ts) * 1e3, sorted(ts)[1] * 1e3), flush=True)
# the largest target alone
big = np.asarray([int(sz.max())])
for th, sl in ((1, 1 << 40), (64, 1 << 21), (64, 1 << 19), (64, 1 << 17)):
    t0 = time.perf_counter()
    engine.init_edge_masks_raw(big, seeds=[7], threads=th, out=bufs["pageable"][:int(big[0]) ** 2], slice_values=sl)
    print("  largest target alone: threads %3d slice %-13s: %.1f ms" % (th, "off" if sl > 1 << 30 else sl, (time.perf_counter() - t0) * 1e3), flush=True)
