#!/usr/bin/env python
"""BASELINE.json configs[3] for the record (not a bench line): graph-level explanation, per-graph edge masks batched
across all molecules on 1 GPU.  The real Mutagenicity files are not available offline, so this uses 4337 synthetic
molecule-like graphs (random trees + ring closures, 10..100 atoms, 14 one-hot atom types, padded to 100 x 100 like
the reference's GraphSampler) and a random-init GcnEncoderGraph (D=14, H=O=20, C=2).
    python tools/config4_mutag_like.py [--graphs 4337] [--iters 300]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_model_explainer_amd import models  # noqa: E402
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph, init_edge_mask  # noqa: E402


def molecule_like(rng, max_nodes=100, num_feat=14):
    n = int(rng.integers(10, max_nodes + 1))
    A = np.zeros((max_nodes, max_nodes), np.float32)
    for v in range(1, n):
        u = int(rng.integers(max(0, v - 4), v))
        A[u, v] = A[v, u] = 1
    for _ in range(max(1, n // 8)):
        u, v = rng.integers(0, n, 2)
        if u != v:
            A[u, v] = A[v, u] = 1
    X = np.zeros((max_nodes, num_feat), np.float32)
    X[np.arange(n), rng.integers(0, num_feat, n)] = 1
    return A, X


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4337)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    model = models.GcnEncoderGraph(14, 20, 20, 2, 3, bn=False, args=None)
    subs = []
    for g in range(a.graphs):
        A, X = molecule_like(rng)
        subs.append(Subgraph(A, X, int(rng.integers(0, 2)), 0, None, init_edge_mask(100)))
    job = MaskOptimJob(subs, model.state_dict(), graph_mode=True)
    hy = Hyper(num_iters=a.iters, use_graph=True)
    job.set_masks([s.mask0 for s in subs])
    M0 = job.M.clone()
    for _ in range(1):
        job.M.copy_(M0)
        job.launch(hy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        job.M.copy_(M0)
        job.launch(hy)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    res = job.fetch(hy)
    ok = all(np.isfinite(m).all() and np.array_equal(m, m.T) for m in res.masked_adj[:64])
    print(json.dumps({"config": f"Mutagenicity-like graph mode: {a.graphs} graphs x 100 (padded), {a.iters} iters",
                      "explained_graphs_per_s": a.graphs / dt, "ms_per_batch": dt * 1e3, "sum_n2": job.sum_n2,
                      "alg_hbm_TBps": 28.0 * job.sum_n2 * a.iters / dt / 1e12, "sane": bool(ok)}))


if __name__ == "__main__":
    main()
