#!/usr/bin/env python
"""Measurement tool (GPU box): the host draw of the 16 384-target BA-House x100k set standalone - full stream into a pinned buffer vs the
values on the edges only (gnnx_host_draw_edge_masks) - by threads and slice length."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import MaskOptimJob
wl = bench.Workload("ba100k", 16384)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
dn = engine.khop_device(graph, wl.targets, 3)
job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"])
job._edge_layout()
E = int(job._eoff[-1])
rc = job._rc[:E].cpu().numpy()
sz = dn.sizes
total = int((sz.astype(np.int64) ** 2).sum())
print(len(sz), "targets", "%.3g normals" % total, E, "edges", "cpus", os.cpu_count(), flush=True)
full = torch.empty(total, dtype=torch.float32, pin_memory=True); full.zero_()
out = torch.empty(E, 2, dtype=torch.float32, pin_memory=True); out.zero_()
for th in (32, 64, 96, 128):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); engine.init_edge_masks_raw(sz, seeds=1000 + wl.targets, threads=th, out=full); ts.append(time.perf_counter() - t0)
    print("  full stream, threads %3d: %.1f ms (median %.1f)" % (th, min(ts) * 1e3, sorted(ts)[1] * 1e3), flush=True)
    for sl in (1 << 15, 1 << 17, 1 << 19):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); engine.init_edge_masks_on_edges(sz, 1000 + wl.targets, job._eoff, rc, threads=th, out=out, slice_values=sl); ts.append(time.perf_counter() - t0)
        print("  edges only,  threads %3d slice %7d: %.1f ms (median %.1f)" % (th, sl, min(ts) * 1e3, sorted(ts)[1] * 1e3), flush=True)
pos = job._epos[:E].cpu().numpy()
M = torch.zeros(job.Q); 
