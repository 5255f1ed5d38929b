#!/usr/bin/env python
"""Measurement probe of the XL route on BA-House x100k (GPU): a seed-fixed sample of ALL nodes, sub-graph sizes from the device k-hop pass, then
  (a) XL targets (n > 16 383): prepare stages (count / build / seeded masks on the host / upload) and the 300-iteration launch, per batch;
  (b) targets BOTH forms take (512 < n <= 16 383, route 7): the XL launch against the k_sparse_large launch on the same targets + bit identity.
    python tools/probe_xl.py [--xl 16] [--mid 32] [--iters 300] [--out gpurun_out/probe_xl.json]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnn_model_explainer_amd as pkg
pkg.tune_process()
import numpy as np
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper
from gnn_model_explainer_amd.utils import synthetic


def timed(dev, fn):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(dev)
    return out, (time.perf_counter() - t0) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--xl", type=int, default=16)
    ap.add_argument("--mid", type=int, default=32)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--sample", type=int, default=4000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    ck = helpers.load_ckpt("syn1")
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, ck["sd"])
    g = engine.device_graph(csr, feat, pred, device=dev)
    rng = np.random.default_rng(5)
    samp = np.sort(rng.choice(N, a.sample, replace=False)).astype(np.int64)
    dn, ms = timed(dev, lambda: engine.khop_device(g, samp, 3))
    sizes = dn.sizes.astype(np.int64)
    print(f"k-hop sizes of {len(samp)} nodes: {ms:.1f} ms; n > 16383: {(sizes > 16383).sum()}, 512 < n <= 16383: {((sizes > 512) & (sizes <= 16383)).sum()}", flush=True)
    res = {"sample": int(a.sample), "n_gt_16383": int((sizes > 16383).sum()), "iters": a.iters}
    hy = Hyper(num_iters=a.iters)

    def xl_batch(targets, tag, device_walk=False):
        targets = np.asarray(targets, np.int64)
        dnb, t_khop = timed(dev, lambda: engine.khop_device(g, targets, 3))
        xj, t_plan = timed(dev, lambda: engine.XLJob(g, dnb, None, label[targets], ck["sd"]))
        _, t_mask = timed(dev, lambda: (xj.set_masks_seeded_device if device_walk else xj.set_masks_seeded)(1000 + targets))
        xj.record_clocks()
        _, t_run = timed(dev, lambda: xj.launch(hy))
        xj.reset_masks()
        _, t_run2 = timed(dev, lambda: xj.launch(hy))
        tms = xj.target_ms()
        em = xj.fetch_edges(with_mask=True)
        nanf = float(np.isnan(em.masked_adj).mean())
        per = [dict(n=int(dnb.sizes[k]), edges=int(em.eoff[k + 1] - em.eoff[k]), setup_ms=round(float(tms[k, 0]), 3), loop_ms=round(float(tms[k, 1]), 3),
                    results_ms=round(float(tms[k, 2]), 3)) for k in range(len(targets))]
        info = dict(tag=tag, targets=len(targets), n_min=int(dnb.sizes.min()), n_max=int(dnb.sizes.max()), edges=int(xj.E), khop_ms=t_khop, count_build_ms=t_plan,
                    masks_ms=t_mask, device_walk=device_walk, run_ms=t_run, run2_ms=t_run2, ms_per_target=t_run2 / len(targets), nan_frac=nanf,
                    ws_rows_MB=xj.ws_rows.numel() / 2 ** 20, ws_entries_MB=xj.ws_entries.numel() / 2 ** 20, sum_n2=float((dnb.sizes.astype(np.float64) ** 2).sum()),
                    sum_target_ms=float(tms.sum()), per_target=per)
        print(json.dumps({k: v for k, v in info.items() if k != "per_target"}), flush=True)
        for q in sorted(per, key=lambda q: q["n"])[:: max(1, len(per) // 8)]:
            print("   ", q, flush=True)
        return xj, em, dnb, info

    # (a) the XL class
    big = samp[sizes > 16383]
    big = big[np.argsort(sizes[sizes > 16383])]
    pick = big[np.linspace(0, len(big) - 1, min(a.xl, len(big))).astype(int)]
    xj, em, dnb, info = xl_batch(pick, "xl n > 16383")
    res["xl"] = info
    del xj
    xj, em_dw, _, info_dw = xl_batch(pick, "xl n > 16383, device engine walk", device_walk=True)
    res["xl_device_walk"] = info_dw
    res["xl_device_walk"]["masks_equal_host_walk"] = bool(np.array_equal(em_dw.masked_adj, em.masked_adj))
    print("device walk == host walk:", res["xl_device_walk"]["masks_equal_host_walk"], flush=True)
    del xj
    # (b) both forms: the mid-size targets route 7 takes
    mid = samp[(sizes > 600) & (sizes <= 16383)]
    midp = mid[np.linspace(0, len(mid) - 1, min(a.mid, len(mid))).astype(int)]
    dnm = engine.khop_device(g, midp, 3)
    job0 = engine.MaskOptimJob.from_csr(g, dnm, None, label[midp], ck["sd"])
    r0 = job0.route()
    res["mid_routes"] = {int(k): int((r0 == k).sum()) for k in np.unique(r0)}
    print("routes of the mid sample:", res["mid_routes"], flush=True)
    job0.close()
    del job0
    mid7 = midp[r0 == 7]
    xj, em, dnb, info = xl_batch(mid7, "xl on route-7 targets")
    res["xl_mid"] = info
    job, t_plan = timed(dev, lambda: engine.MaskOptimJob.from_csr(g, dnb, None, label[mid7], ck["sd"]))
    job._edge_layout()
    E = int(job._eoff[-1])
    rc = job._rc[:E].cpu()
    vals = engine.init_edge_masks_on_edges(dnb.sizes, 1000 + mid7, job._eoff, rc, threads=8)
    job.set_masks_on_edges(vals)
    ehy = Hyper(num_iters=a.iters, edge_results_only=True)
    _, t7 = timed(dev, lambda: job.launch(ehy))
    job.reset_masks()
    _, t7b = timed(dev, lambda: job.launch(ehy))
    e7 = job.fetch_edges(with_mask=True)
    same = bool(np.array_equal(e7.masked_adj, em.masked_adj) and np.array_equal(e7.mask_rc, em.mask_rc) and np.array_equal(e7.feat_mask, em.feat_mask))
    res["route7_mid"] = dict(plan_pack_ms=t_plan, run_ms=t7, run2_ms=t7b, bit_identical_to_xl=same, max_abs_diff=float(np.abs(e7.masked_adj - em.masked_adj).max()),
                             resident_ms=job.resident_times())
    print(json.dumps(res["route7_mid"]), flush=True)
    if (r0 == 0).any():
        xj2, em2, dnb2, info2 = xl_batch(midp[r0 == 0], "xl on the mid targets route 7 cannot take")
        res["xl_mid_streaming"] = info2
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
