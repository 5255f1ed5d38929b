#!/usr/bin/env python
"""`--method att` at scale: the LIVE reference's explanations of the CLI's default syn1 node list `range(400, 700, 5)` and of syn1's largest
motif neighbourhood (target 300, n = 310), 300 epochs each, for the attention encoder its own train.py trains (`train.syn_task1` with
--method att, seeds 0: the encoder of tests/golden/options_explain.npz, asserted bit-identical).  Round 3 pinned k_att on two targets of
100 epochs; this pins it on 61 targets at the full horizon and after the first 50 epochs (the reference's Adam state there).

There is no closed-form oracle for the attention encoder, so the conditioning of a target is measured on the reference itself: two more
runs with the initial mask perturbed by +-1 ulp (mask * (1 + 2e-7 (u - 0.5)) inside ExplainModule.__init__, after its own normal_ draw);
sens50 / sens300 = the largest deviation (masked adjacency on the edges, sigmoid(feat_mask)) of those runs from the unperturbed one.

    python tests/golden/make_golden_att_scale.py --procs 6

-> tests/golden/att_scale.npz: targets [T], nb_off / nb_flat, node_idx_new, eoff, vals / feat_sig (300-epoch output: returned mask on the
   edges, sigmoid(feat_mask)), M50 [E][2] / f50 [T][D] (mask entries of both directions and feat_mask after 50 optimiser steps), sens50 /
   sens300 [T], epochs = 300, early = 50
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

EPOCHS, EARLY, TRIALS, EPS = 300, 50, 2, 2e-7


def _sig64(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def _worker(job):
    work, targets = job
    import make_golden as mg
    import make_golden_windows as mgw
    mg.install_shims()
    import torch
    torch.set_num_threads(1)
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    io_utils.log_graph = lambda *a_, **k_: None
    eargs = mg.explain_args("syn1", work, EPOCHS)
    eargs.method = "att"
    os.makedirs(eargs.logdir, exist_ok=True)
    with mg.quiet():
        ckpt = io_utils.load_ckpt(eargs)
    cg = ckpt["cg"]
    model = models.GcnEncoderNode(input_dim=10, hidden_dim=20, embedding_dim=20, label_dim=4, num_layers=3, bn=False, args=eargs)
    model.load_state_dict(ckpt["model_state"])
    Z = np.load(os.path.join(HERE, "options_explain.npz"))
    for k, v in ckpt["model_state"].items():
        assert np.array_equal(Z["route:att:w:" + k], v.detach().numpy().astype(np.float32)), "attention encoder differs from options_explain.npz"
    snaps, rc_box = mgw.install_snapshots(explain)
    # perturbation of the initial mask: applied by a wrapper AROUND the snapshot wrapper's __init__ (after the reference's own normal_ draw)
    cls = explain.ExplainModule
    inner = cls.__init__
    pert = {"gen": None}

    def init(self, *a, **k):
        inner(self, *a, **k)
        if pert["gen"] is not None:
            with torch.no_grad():
                self.mask.mul_(1 + EPS * (torch.rand(self.mask.shape, generator=pert["gen"]) - 0.5))
    cls.__init__ = init
    with mg.quiet():
        ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"], train_idx=cg["train_idx"],
                               args=eargs, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    out = []
    for t in targets:
        t0 = time.time()
        with mg.quiet():
            new, sub_adj, _, _, nb = ex.extract_neighborhood(t)
        r, c = np.nonzero(np.triu(sub_adj, 1))
        rc_box["rc"] = (r, c)
        runs = []
        for trial in range(1 + TRIALS):
            pert["gen"] = None if trial == 0 else torch.Generator().manual_seed(t * 100003 + trial)
            with mg.quiet():
                torch.manual_seed(1000 + t)
                ma = ex.explain(t)
            mod, rec = snaps[-1]
            del snaps[:]
            M50, f50 = rec[EARLY][0], rec[EARLY][3]
            runs.append((ma[r, c].astype(np.float64), _sig64(mod.feat_mask.detach().numpy()), 0.5 * (_sig64(M50[:, 0]) + _sig64(M50[:, 1])), _sig64(f50), M50, f50))
        pert["gen"] = None
        base = runs[0]
        dev = lambda a, b: float(np.abs(a - b).max()) if len(a) else 0.0
        s300 = max(max(dev(x[0], base[0]), dev(x[1], base[1])) for x in runs[1:])
        s50 = max(max(dev(x[2], base[2]), dev(x[3], base[3])) for x in runs[1:])
        out.append(dict(key=int(t), nb=nb.astype(np.int32), new=int(new), nedges=len(r), vals=base[0].astype(np.float32), fsig=base[1].astype(np.float32),
                        M50=base[4], f50=base[5], s50=s50, s300=s300))
        print(f"  att target {t}: n={len(nb)} edges={len(r)} 1-ulp sensitivity after 50 / 300 epochs {s50:.1e} / {s300:.1e} {time.time() - t0:.0f} s", flush=True)
        for f in os.listdir(eargs.logdir):
            p = os.path.join(eargs.logdir, f)
            if os.path.isfile(p):
                os.remove(p)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gw/work")
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args()
    work = os.path.join(a.work, "route_att")
    if not os.path.exists(os.path.join(work, "ckpt", "syn1_att_h20_o20.pth.tar")) and not os.path.isdir(os.path.join(work, "ckpt")):
        import random
        import make_golden as mg
        mg.install_shims()
        import torch
        import train
        targs = mg.train_args("syn1", work)
        targs.method = "att"
        np.random.seed(0)
        random.seed(0)
        torch.manual_seed(0)
        with mg.quiet():
            train.syn_task1(targs)
    targets = ([300] + list(range(400, 700, 5)))[:a.limit]
    jobs = [(work, targets[k::a.procs * 2]) for k in range(a.procs * 2)]
    jobs = [j for j in jobs if j[1]]
    t0 = time.time()
    with mp.get_context("spawn").Pool(a.procs) as pool:
        res = [r for part in pool.map(_worker, jobs) for r in part]
    res.sort(key=lambda r: r["key"])
    out = dict(targets=np.asarray([r["key"] for r in res], np.int64), nb_off=np.cumsum([0] + [len(r["nb"]) for r in res]).astype(np.int64),
               nb_flat=np.concatenate([r["nb"] for r in res]), node_idx_new=np.asarray([r["new"] for r in res], np.int32),
               eoff=np.cumsum([0] + [r["nedges"] for r in res]).astype(np.int64), vals=np.concatenate([r["vals"] for r in res]),
               feat_sig=np.stack([r["fsig"] for r in res]), M50=np.concatenate([r["M50"] for r in res]).astype(np.float32),
               f50=np.stack([r["f50"] for r in res]).astype(np.float32), sens50=np.asarray([r["s50"] for r in res], np.float32),
               sens300=np.asarray([r["s300"] for r in res], np.float32), epochs=np.int64(EPOCHS), early=np.int64(EARLY), trials=np.int64(TRIALS))
    np.savez_compressed(os.path.join(HERE, "att_scale.npz"), **out)
    print(f"att_scale: {len(res)} targets (n = {int(np.diff(out['nb_off']).min())} ... {int(np.diff(out['nb_off']).max())}) in {time.time() - t0:.0f} s; "
          f"1-ulp sensitivity > 2e-6 on {int((out['sens50'] > 2e-6).sum())} targets after 50 epochs, {int((out['sens300'] > 2e-6).sum())} after 300", flush=True)


if __name__ == "__main__":
    main()
