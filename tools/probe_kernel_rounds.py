#!/usr/bin/env python
"""Measurement tool (GPU box): per-launch time of each streaming kernel vs number of tiles (targets of n = 150 ->
5 row blocks, 15 tile pairs each): looks for the step where the grid no longer fits one round of workgroups."""
import os, sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import helpers
from gnn_model_explainer_amd.engine import MaskOptimJob, Subgraph, Hyper

rng = np.random.default_rng(0)
sd = helpers.random_model(rng, 10, 20, 20, 4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
nb = (n + 31) // 32
pool = []
for _ in range(160):
    A, X = helpers.random_graph(rng, n, 10, density=0.04)
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    pool.append(Subgraph(A, X, 1, 3, rng.integers(0, 4, n), m0))
hy = Hyper(num_iters=20, use_graph=False, use_resident=False)
names = {0: "mask", 1: "FWD1", 2: "FWD2", 3: "head", 4: "BWD1"}
print("targets pairs rowblocks | " + " ".join(f"{v:>7s}" for v in names.values()) + " | sum us")
for count in (8, 20, 40, 60, 66, 68, 70, 72, 80, 100, 120, 136, 140, 160):
    job = MaskOptimJob(pool[:count], sd)
    job.run([s.mask0 for s in pool[:count]], hy)
    t = [job.time_kernel(hy, k, 200)[0] * 1e3 for k in names]
    print(f"{count:7d} {count * nb * (nb + 1) // 2:5d} {count * nb:9d} | " + " ".join(f"{x:7.2f}" for x in t) + f" | {sum(t):6.1f}", flush=True)
    job.close()
