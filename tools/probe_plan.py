#!/usr/bin/env python
"""Measurement tool (GPU box): where the plan / pack / analyze time of a BA-House x100k batch goes (wall clock with device
syncs between the steps of MaskOptimJob.from_csr)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import MaskOptimJob

wl = bench.Workload(sys.argv[2] if len(sys.argv) > 2 else "ba100k", int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
for rep in range(4):
    stamps = []
    def mark(name):
        torch.cuda.synchronize()
        stamps.append((name, time.perf_counter()))
    mark("start")
    dn = engine.khop_device(graph, wl.targets, 3)
    mark("khop")
    orig = {k: getattr(MaskOptimJob, k) for k in ("_create_plan", "_alloc_device", "analyze")}
    def wrap(name):
        f = orig[name]
        def g(self, *a, **kw):
            mark("before " + name)
            r = f(self, *a, **kw)
            mark(name)
            return r
        return g
    for k in orig:
        setattr(MaskOptimJob, k, wrap(k))
    job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"])
    mark("end")
    for k, f in orig.items():
        setattr(MaskOptimJob, k, f)
    print("rep", rep, " ".join("%s %.1f ms |" % (n, (t - stamps[i][1]) * 1e3) for i, (n, t) in enumerate(stamps[1:])), flush=True)
    job.close()
    del job
