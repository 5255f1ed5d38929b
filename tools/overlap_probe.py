#!/usr/bin/env python
"""Measurement tool: does a long-running kernel on a second stream overlap with the hipGraph replay of the
mask-optimisation job?  (Decides whether an on-chip-resident kernel for small targets can run beside the
multi-launch chain of the large ones.)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob

ck, subs, _ = bench.build_workload("syn1", 300)
job = MaskOptimJob(subs, ck["sd"]); hy = Hyper(num_iters=300, use_graph=True)
job.set_masks([s.mask0 for s in subs]); M0 = job.M.clone()
def step():
    job.M.copy_(M0); job.launch(hy)
step(); torch.cuda.synchronize()
def timed(fn, n=3):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("graph alone           %.2f ms" % timed(step))
side = torch.cuda.Stream()
x = torch.randn(6144, 6144, device="cuda")            # ONE long kernel (fp32 GEMM) on the side stream
def busy():
    with torch.cuda.stream(side):
        return torch.mm(x, x)
busy(); torch.cuda.synchronize()
print("side work alone       %.2f ms" % timed(busy))
def both():
    busy(); step()
print("graph + side together %.2f ms" % timed(both))
