#!/usr/bin/env python
"""Measurement tool (GPU box): does the resident kernel (side stream) overlap the streaming chain (hipGraph on the job
stream)?  syn1-like batch: S small targets (n = 20, resident) + B big targets (n = 150, streaming); times the two
parts alone and together."""
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


def make(n, count, density):
    out = []
    for _ in range(count):
        A, X = helpers.random_graph(rng, n, 10, density=density)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        out.append(Subgraph(A, X, 1, 3, rng.integers(0, 4, n), m0))
    return out


def timed(subs, **kw):
    job = MaskOptimJob(subs, sd)
    hy = Hyper(num_iters=iters, use_graph=True, **kw)
    m0s = [s.mask0 for s in subs]
    job.run(m0s, hy)
    best = 1e9
    for _ in range(3):
        job.set_masks(m0s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.launch(hy)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


small, mid, big = make(20, 300, 0.15), make(80, 30, 0.08), make(150, 70, 0.04)
print(f"small only (resident): {timed(small):.2f} ms")
print(f"mid only   (resident): {timed(mid):.2f} ms")
print(f"mid only  (streaming): {timed(mid, use_resident=False):.2f} ms")
print(f"big only  (streaming): {timed(big):.2f} ms")
print(f"small + big  (hybrid): {timed(small + big):.2f} ms")
print(f"small + big (all streaming): {timed(small + big, use_resident=False):.2f} ms")
print(f"mid + big    (hybrid): {timed(mid + big):.2f} ms")
print(f"mid + big (all streaming): {timed(mid + big, use_resident=False):.2f} ms")
print(f"small + mid + big (hybrid): {timed(small + mid + big):.2f} ms")
print(f"small + mid + big (all streaming): {timed(small + mid + big, use_resident=False):.2f} ms")
