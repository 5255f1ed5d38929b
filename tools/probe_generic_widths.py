#!/usr/bin/env python
"""Speed of the generic-width instantiations (`<16, 16>`: any encoder whose hidden / output width is not 20 or whose input width exceeds 14) next
to the reference's widths, on the headline workload's graph: syn1, all 400 motif nodes, 300 iterations, random encoders of
hidden = output = 20 (compile-time widths) / 32 / 16 (parity is tested on those shapes, speed never was).

    python tools/probe_generic_widths.py        (GPU box)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from gnn_model_explainer_amd import engine  # noqa: E402
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob  # noqa: E402
from gnn_model_explainer_amd.utils.graph_utils import KHopIndex  # noqa: E402
from oracle import closed_form  # noqa: E402


def main():
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    targets = np.arange(300, ck["num_nodes"], dtype=np.int64)
    for hidden in (20, 32, 16):
        rng = np.random.default_rng(hidden)
        sd = helpers.random_model(rng, 10, hidden, hidden, 4)
        pred = np.zeros((ck["num_nodes"], 4), np.float32)
        pred[np.arange(ck["num_nodes"]), rng.integers(0, 4, ck["num_nodes"])] = 1.0
        graph = engine.device_graph(idx.csr, ck["feat"], pred)
        dn = engine.khop_device(graph, targets, 3)
        job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], sd)
        raw = engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets)
        job.set_masks_raw(raw)
        hy = Hyper(num_iters=300, edge_results_only=True)
        job.launch(hy)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            job.set_masks_raw_resident()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.launch(hy)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        # a spot check against the closed form (12 iterations, three targets)
        k3 = [0, 150, 399]
        lists = dn.lists()
        errs = []
        for k in k3:
            nb = lists[k]
            A = idx.sub_adjacency(nb)
            m0 = helpers.seeded_mask0(int(targets[k]), len(nb)).numpy()
            sg = engine.Subgraph(A, ck["feat"][nb], int(ck["label"][targets[k]]), int(dn.rows[k]), np.argmax(pred[nb], 1), m0)
            j1 = MaskOptimJob([sg], sd)
            got = j1.run([m0], Hyper(num_iters=12)).masked_adj[0]
            o = closed_form.ClosedFormOracle(sg.adj, sg.feat, sd, sg.gt_label, sg.pred_label, sg.target_row, m0)
            errs.append(float(np.abs(got - o.run(12)).max()))
        r = job.route()
        print(f"hidden = output = {hidden}: routes {dict(zip(*np.unique(r, return_counts=True)))}, loop {np.median(ts):.3f} ms per 400-target batch "
              f"({400 / np.median(ts) :.1f} k nodes/s), vs the closed form after 12 iterations on 3 targets: {max(errs):.1e}", flush=True)


if __name__ == "__main__":
    main()
