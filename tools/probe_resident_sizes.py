#!/usr/bin/env python
"""Measurement tool (GPU box): time per iteration of the resident kernels vs the streaming path for batches of
equal-size random targets (n = 30 / 60 / 90 -> 1 / 2 / 3 row blocks)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers
from gnn_model_explainer_amd.engine import MaskOptimJob, Subgraph, Hyper

rng = np.random.default_rng(0)
sd = helpers.random_model(rng, 10, 20, 20, 4)
iters = 300
for n in (30, 60, 90):
    for count in (64, 256, 512):
        subs = []
        for _ in range(count):
            A, X = helpers.random_graph(rng, n, 10, density=0.1)
            m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
            subs.append(Subgraph(A, X, 1, 3, rng.integers(0, 4, n), m0))
        for resident in (True, False):
            job = MaskOptimJob(subs, sd)
            hy = Hyper(num_iters=iters, use_graph=True, use_resident=resident)
            m0s = [s.mask0 for s in subs]
            job.run(m0s, hy)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                job.set_masks(m0s)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                job.launch(hy)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            dt = best
            print(f"n={n} count={count} resident={resident}: {dt*1e3:.2f} ms  {dt/iters*1e6:.1f} us/iter", flush=True)
