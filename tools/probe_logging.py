#!/usr/bin/env python
"""Measurement tool (GPU box): the reference CLI's single-node invocation (explainer_main.py --explain-node, print_training=True,
explain.py:149-159) on the mirror - which kernel the logging run takes, its loss trace against the live reference's, its wall time."""
import io, os, sys, time, contextlib, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
targets = [int(t) for t in gx["targets"]]
subs = []
for t in targets:
    nb = gx[f"{t}:neighbors"]
    A, X, lab, yhat = helpers.subgraph(ck, nb)
    new = int(gx[f"{t}:node_idx_new"])
    subs.append(Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy()))
iters = int(gx["epochs"])
for use_resident, what in ((True, "resident kernel, logging form + k_dead_entries"), (False, "dense streaming kernels (round 3's only logging route)")):
    job = MaskOptimJob(subs, ck["sd"])
    hy = Hyper(num_iters=iters, record_loss=True, use_resident=use_resident)
    job.run([s.mask0 for s in subs], hy)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = job.run([s.mask0 for s in subs], hy)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    worst = 0.0
    for i, t in enumerate(targets):
        got, want = res.loss[i][:, :5].sum(1), gx[f"{t}:loss"]
        worst = max(worst, float(np.abs(got / want - 1).max()))
    print(f"{what}: routes {sorted(set(int(r) for r in job.route()))}, {len(targets)} golden syn1 targets (n = {[s.adj.shape[0] for s in subs]}), {iters} epochs with loss logging: "
          f"{ms:.1f} ms; loss trace vs the live reference's: worst relative deviation over all epochs {worst:.2e}")
