#!/usr/bin/env python
"""Windowed fixtures for BASELINE config 5 (BA-House x100k): the LIVE reference's optimiser state every 50 epochs - and its decisions
at every epoch - on route-stratified targets of the 99 997-node graph, n = 6 ... > 4095.

The reference cannot extract these sub-graphs itself (its `neighborhoods` builds a dense 100k x 100k matrix, graph_utils.py:147-158), but
its ExplainModule - the whole hot path, explain.py:582-820 - runs unmodified on a sub-graph extracted by sparse BFS (BASELINE.md section 3;
the BFS is element-wise identical to `neighborhoods` on syn1).  So: reference_explain_subgraph of make_golden_full.py (the body of
Explainer.explain around the reference's own ExplainModule), with the optimiser snapshots of make_golden_windows.py and the forward hooks
of make_golden_decisions.py installed.  This pins k_sparse_large - the kernel of the scaling workload's largest targets, which round 3
could only pin to the dense streaming kernels - to the reference's own state, window by window.

    python tests/golden/make_golden_ba100k_windows.py --procs 8            # about 15 minutes

Written: tests/golden/ba100k_windows.npz - targets [T], size [T], nb_off / nb_flat (sub-graph node ids), node_idx_new, eoff (upper-triangle
edges), epochs = 50..300, M / m / v [6][E][2], f / mf / vf [6][T][D], vals / feat_sig (the 300-epoch output), cond50 / sens50 / noise50
[T][6] (the three CPU-only conditioning probes, computed for n <= PROBE_N_MAX; 0 = not probed), probed [T] - and
tests/golden/ba100k_decisions.npz in the layout of make_golden_decisions.py.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import make_golden_decisions as mgd  # noqa: E402
import make_golden_windows as mgw  # noqa: E402

EPOCHS, WIN = 300, 50
PROBE_N_MAX = 700          # the closed-form probes are dense n x n numpy: affordable up to here


def pick_targets():
    """route-stratified motif nodes of the BA-House x100k graph: the 13 of ba100k_explain.npz plus a spread over every size class"""
    from gnn_model_explainer_amd.utils import synthetic
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    idx = KHopIndex(csr, 3)
    old = [int(t) for t in np.load(os.path.join(HERE, "ba100k_explain.npz"))["targets"]]
    rng = np.random.default_rng(11)
    cand = np.sort(rng.choice(np.arange(42857, N), 8000, replace=False))
    size = np.asarray([len(nb) for nb in idx.neighbors_batch(cand)])
    picks = list(old)
    for lo, hi, k in ((0, 32, 3), (32, 128, 4), (128, 512, 5), (512, 1200, 5), (1200, 2500, 5), (2500, 4095, 4), (4095, 10 ** 9, 3)):
        ids = np.nonzero((size > lo) & (size <= hi))[0]
        ids = ids[np.argsort(size[ids], kind="stable")]
        for i in ids[np.linspace(0, len(ids) - 1, min(k, len(ids))).astype(int)]:
            if int(cand[i]) not in picks:
                picks.append(int(cand[i]))
    return sorted(picks)


def _worker(job):
    work, targets = job
    mg = mgw._setup()
    import torch
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    import make_golden_full as mgf
    import make_golden_noise_probe as mnp
    from gnn_model_explainer_amd.utils import synthetic
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    from oracle import closed_form
    args = mg.explain_args("syn1", work, EPOCHS)
    with mg.quiet():
        ckpt = io_utils.load_ckpt(args)
    model = models.GcnEncoderNode(input_dim=10, hidden_dim=20, embedding_dim=20, label_dim=4, num_layers=3, bn=False, args=args)
    model.load_state_dict(ckpt["model_state"])
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in ckpt["model_state"].items()}
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, sd)
    idx = KHopIndex(csr, 3)
    snaps, rc_box = mgw.install_snapshots(explain)
    recd = mgd.Recorder(model, False)
    out = []
    for t in targets:
        t0 = time.time()
        nb = idx.neighbors(t)
        new = int(np.searchsorted(nb, t))
        sub = idx.sub_adjacency(nb)
        pl = np.argmax(pred[nb], axis=1)
        rc_box["rc"] = np.nonzero(np.triu(sub, 1))
        recd.take()
        ma, fsig, mask0, loss, maxm = mgf.reference_explain_subgraph(mg, model, sub, feat[nb], label[nb], pl, new, EPOCHS, work, 1000 + t)
        mod, rec = snaps[-1]
        del snaps[:]
        n = len(nb)
        pat = (sub != 0) & ~np.eye(n, dtype=bool)
        lvl = np.full(n, 9)
        lvl[new] = 0
        for d in (1, 2):
            lvl[(pat[lvl == d - 1].sum(0) > 0) & (lvl > d)] = d
        pre, last = recd.take()
        piece = mgd.encode(pre, last, (lvl <= 2, lvl <= 1), False)
        piece["key"] = int(t)
        r, c = rc_box["rc"]
        probes = np.zeros((3, EPOCHS // WIN), np.float32)
        if n <= PROBE_N_MAX:
            o = closed_form.ClosedFormOracle(sub.astype(np.float32), feat[nb], sd, int(label[t]), pl, new, mask0)
            o._off_edges, o._M0, o._lvl = (sub == 0), np.asarray(mask0, np.float32), lvl
            E = len(r)
            z2, zd = np.zeros((E, 2), np.float32), np.zeros(10, np.float32)
            M0 = np.stack([mask0[r, c], mask0[c, r]], 1).astype(np.float32)
            state = lambda k: (M0, z2, z2, zd, zd, zd) if k == 0 else rec[k]
            for w in range(EPOCHS // WIN):
                probes[0, w], probes[1, w], _ = mgw._oracle_dev(o, (r, c), state(WIN * w), rec[WIN * (w + 1)], WIN * w, WIN, seed=(int(t), w))
                probes[2, w] = mnp._probe(o, (r, c), state(WIN * w), WIN * w, WIN, int(t))
        out.append(dict(key=int(t), nb=nb.astype(np.int32), new=new, nedges=len(r), coarse=[rec[k] for k in range(WIN, EPOCHS + 1, WIN)],
                        vals=ma[r, c].astype(np.float32), fsig=fsig, probes=probes, probed=n <= PROBE_N_MAX, piece=piece))
        print(f"  ba100k target {t}: n={n} edges={len(r)} loss={loss:.4f} max|M|={maxm:.2f} probes max {probes.max():.1e} {time.time() - t0:.0f} s", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gw/work")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args()
    targets = pick_targets()[:a.limit]
    print(f"{len(targets)} targets", flush=True)
    # longest jobs first, one target per job: the n > 4095 targets take minutes each
    from gnn_model_explainer_amd.utils import synthetic
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    N, edges, _ = synthetic.ba_house(42857, 11428, seed=0)
    idx = KHopIndex(synthetic.csr_from_edges(N, edges), 3)
    size = {t: len(idx.neighbors(t)) for t in targets}
    jobs = [(a.work, [t]) for t in sorted(targets, key=lambda t: -size[t])]
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = [r for part in pool.map(_worker, jobs, chunksize=1) for r in part]
    res.sort(key=lambda r: r["key"])
    T = len(res)
    nck = EPOCHS // WIN
    cat = lambda i, j: np.concatenate([r["coarse"][i][j] for r in res])
    out = dict(targets=np.asarray([r["key"] for r in res], np.int64), size=np.asarray([len(r["nb"]) for r in res], np.int32),
               nb_off=np.cumsum([0] + [len(r["nb"]) for r in res]).astype(np.int64), nb_flat=np.concatenate([r["nb"] for r in res]),
               node_idx_new=np.asarray([r["new"] for r in res], np.int32), eoff=np.cumsum([0] + [r["nedges"] for r in res]).astype(np.int64),
               epochs=np.arange(WIN, EPOCHS + 1, WIN).astype(np.int64), sub=np.int64(10), vals=np.concatenate([r["vals"] for r in res]),
               feat_sig=np.stack([r["fsig"] for r in res]).astype(np.float32), probed=np.asarray([r["probed"] for r in res], bool),
               cond50=np.stack([r["probes"][0] for r in res]), sens50=np.stack([r["probes"][1] for r in res]),
               noise50=np.stack([r["probes"][2] for r in res]))
    for j, nm in enumerate(("M", "m", "v")):
        out[nm] = np.stack([cat(i, j) for i in range(nck)]).astype(np.float32)
    for j, nm in ((3, "f"), (4, "mf"), (5, "vf")):
        out[nm] = np.stack([np.stack([r["coarse"][i][j] for r in res]) for i in range(nck)]).astype(np.float32)
    # (the keys helpers.Windows expects of every <name>_windows.npz: no gate probe, no 10-epoch snapshots here)
    E, D = int(out["eoff"][-1]), out["f"].shape[2]
    out.update(gate50=np.full((T, nck), np.inf, np.float32), gate=np.float64(5e-7), fine_tw=np.zeros((0, 2), np.int32), fine_off=np.zeros(1, np.int64),
               cond10=np.zeros((0, 5), np.float32), sens10=np.zeros((0, 5), np.float32), gate10=np.zeros((0, 5), np.float32))
    for nm in ("fine_M", "fine_m", "fine_v"):
        out[nm] = np.zeros((4, 0, 2), np.float32)
    for nm in ("fine_f", "fine_mf", "fine_vf"):
        out[nm] = np.zeros((4, 0, D), np.float32)
    np.savez_compressed(os.path.join(HERE, "ba100k_windows.npz"), **out)
    dec = mgd.assemble([r["piece"] for r in res], "targets", False)
    np.savez_compressed(os.path.join(HERE, "ba100k_decisions.npz"), **dec)
    print(f"ba100k: {T} targets (n = {out['size'].min()} ... {out['size'].max()}, {int((out['size'] > 512).sum())} beyond the LDS-resident classes, "
          f"{int((out['size'] > 4095).sum())} beyond 4095) in {time.time() - t0:.0f} s; {len(dec['ev'])} gate-word changes", flush=True)


if __name__ == "__main__":
    main()
