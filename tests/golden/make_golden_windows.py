#!/usr/bin/env python
"""Windowed ("teacher-forced") golden fixtures: the optimiser state of the REAL reference (/root/reference) along the
trajectory of EVERY target, so that an implementation can be started from the reference's own state at epoch k and compared
with the reference's state at epoch k + 50 (or k + 10) - round-off has 50 (10) iterations to act instead of 300, which pins
iterations 50..300 on the targets whose full-horizon outcome is chaotic (VERDICT r2 "next" #1).

Runs only in the build container.  Nothing from the reference is copied: it is imported and executed under the seed protocol
(torch.manual_seed(1000 + target) immediately before Explainer.explain, 300 epochs, explain.py:137-146); a wrapper around the
`step` of the torch.optim.Adam the reference itself builds (utils/train_utils.py:9-10) snapshots, after every 10th step,
    mask / exp_avg / exp_avg_sq on the two directed entries of every sub-graph edge, feat_mask / exp_avg / exp_avg_sq.

    python tests/golden/make_golden_windows.py --what syn1,syn4,syn5,config4 --procs 8

Outcome-blind window classification (decided on the CPU alone, before any GPU run): the closed-form fp32 oracle
(oracle/closed_form.py: same mathematics, other summation order) is started from the reference's state at every 50-epoch
boundary and run for 50 iterations; `cond50[t][w]` is its deviation (masked adjacency and sigmoid(feat_mask)) from the reference's state at
the end of the window.  The two CPU implementations share their BLAS, so their agreement alone understates how sensitive a
window is; `sens50[t][w]` therefore measures the window's own conditioning: the largest deviation of 4 closed-form runs whose
starting mask entries are perturbed by +-1 ulp from the unperturbed closed-form run.  Windows with max(cond50, sens50) > 2e-6
("flagged": a ReLU gate / max-pool tie flips inside them, or Adam amplifies a one-ulp difference beyond 2e-6 within 50 epochs)
additionally get the four 10-epoch snapshots inside the window and `cond10` / `sens10` of their five sub-windows.
Third criterion, `gate50[t][w]` (and `gate10`): the smallest distance of any decision that reaches the loss from its boundary during
the window of the closed-form run - |U| at the ReLU gates (models.py:241, 251) of the rows the prediction reads, and in graph mode
the margin of every max-pool (models.py:283-300).  A margin below 5e-7 is inside the fp32 round-off of the sum that produced
it: which side of the gate an implementation lands on is then decided by its summation order, and Adam's scale-free step turns
that one different gradient into a 1e-4 .. 1e-3 difference within a few iterations (every window in which the GPU kernels left
the reference by more than 1e-5 while both CPU probes saw nothing had such a gate: 7e-8, 8e-8, 0, 1.4e-7, 0, 0, 1e-7, 0).

Fixtures written (tests/golden/<name>_windows.npz):
  targets (or graphs) [T], eoff [T+1] (upper-triangle edges, order of <name>_full_explain.npz), epochs [6] = 50..300,
  M / m / v [6][E][2] float32 (entry (r,c), entry (c,r)), f / mf / vf [6][T][D], cond50 / sens50 / gate50 [T][6],
  fine_tw [F][2] = (target index, window) of the flagged windows, fine_off [F+1] (edge offsets), fine_M / fine_m / fine_v [4][Ef][2]
  (epochs 50 w + 10, 20, 30, 40), fine_f / fine_mf / fine_vf [4][F][D], cond10 / sens10 / gate10 [F][5];
  config4 only: the whole job description (vals / feat_sig of the 300-epoch output, cond_mask / cond_feat, weights) because it
  covers 512 graphs where config4_explain.npz has 64.
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

WIN, SUB, EPOCHS = 50, 10, 300
# (target, window) pairs that get the 10-epoch snapshots although the CPU probes did not flag them (round 6, VERDICT r5 missing #6): windows in
# which the GPU kernels take the other side of a gate the reference decides by 1e-6 ... 2e-5 - with the snapshots the decision suite leaves only
# the 10 epochs around that tie ungated instead of the whole 50-epoch window (tests/golden/syn5_ties.json: sub = -1 before).
FORCE_FINE = {"syn5": {(532, 2), (557, 5), (620, 3)}}
FLAG = 2e-6
MOTIF_START = {"syn1": 300, "syn4": 511, "syn5": 511}


def _setup():
    import torch
    import make_golden as mg
    mg.install_shims()
    torch.set_num_threads(1)
    return mg


def install_snapshots(explain_mod):
    """Wrap ExplainModule.__init__ so that the optimiser the reference builds snapshots its state every SUB steps on the edges in
    rc_box["rc"] (set by the caller before every explanation).  -> (snaps, rc_box), once per process."""
    cls = explain_mod.ExplainModule
    if getattr(cls, "_gnnx_windows", False):
        return cls._gnnx_snaps, cls._gnnx_rc_box
    orig_init = cls.__init__
    snaps, rc_box = [], {}

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self.mask0 = self.mask.detach().clone()
        rec = {}
        snaps.append((self, rec))
        opt = self.optimizer
        orig_step = opt.step
        count = [0]

        def step(*sa, **sk):
            out = orig_step(*sa, **sk)
            count[0] += 1
            if count[0] % SUB == 0:
                r, c = rc_box["rc"]
                st, sf = opt.state[self.mask], opt.state[self.feat_mask]
                g = lambda x: np.stack([x.detach().numpy()[r, c], x.detach().numpy()[c, r]], 1).astype(np.float32)
                rec[count[0]] = (g(self.mask), g(st["exp_avg"]), g(st["exp_avg_sq"]), self.feat_mask.detach().numpy().copy(),
                                 sf["exp_avg"].numpy().copy(), sf["exp_avg_sq"].numpy().copy())
            return out

        opt.step = step

    cls.__init__ = init
    cls._gnnx_windows, cls._gnnx_snaps, cls._gnnx_rc_box = True, snaps, rc_box
    return snaps, rc_box


def _sig64(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def abar_edges(Mrc, w=1.0):
    """masked adjacency on the edges from the two directed mask entries (explain.py:665-678), float64."""
    return w * 0.5 * (_sig64(Mrc[:, 0]) + _sig64(Mrc[:, 1]))


TRIALS, ULP = 4, 2e-7     # sensitivity probe: TRIALS runs with the mask entries on the edges multiplied by 1 + ULP (u - 0.5), u ~ U[0, 1): +-1 ulp


GATE = 5e-7     # a ReLU input / max-pool margin this close to zero is inside fp32 round-off of a sum of <= 20 products of magnitude <= 1


def gate_margin(o):
    """Smallest distance from a decision boundary in the last iteration of closed-form oracle `o`: |U_l| at the ReLU gates that reach
    the loss (node mode: layer 1 on the rows within two hops of the target, layer 2 on the target and its neighbours - the only rows
    whose activations the prediction reads; graph mode: every row) and, in graph mode, the margin between the largest
    and the next smaller value of every max-pooled column (models.py:283-300; bitwise equal rows - the zero-padded ones - tie
    harmlessly: both implementations take the first)."""
    U = o.stages["U"]
    if not o.graph_mode:
        return float(min(np.abs(U[0][o._lvl <= 2]).min(), np.abs(U[1][o._lvl <= 1]).min()))
    g = float(min(np.abs(U[0]).min(), np.abs(U[1]).min()))
    for a in (np.maximum(U[0], 0), np.maximum(U[1], 0), U[2]):
        top = a.max(0)
        below = np.where(a < top[None, :], a, -np.inf).max(0)
        live = (top > 0) & np.isfinite(below)
        if live.any():
            g = min(g, float((top - below)[live].min()))
    return g


def _oracle_dev(o, rc, state, ref_end, k0, steps, seed=None):
    """Closed-form oracle started from the reference's `state` (after k0 steps), `steps` iterations -> deviation from ref_end
    (seed None), or - sensitivity of the window to its own input - (deviation from ref_end, largest deviation of TRIALS runs from
    1-ulp-perturbed starts from the unperturbed run)."""
    if seed is not None:
        base = _oracle_out(o, rc, state, k0, steps)
        gate = o._gate_min
        dev = _dev(base, ref_end)
        rng = np.random.default_rng(seed)
        sens = 0.0
        for _ in range(TRIALS):
            M = (state[0] * (1.0 + ULP * (rng.random(state[0].shape) - 0.5))).astype(np.float32)
            sens = max(sens, _dev(_oracle_out(o, rc, (M,) + tuple(state[1:]), k0, steps), base))
        return dev, sens, gate
    return _dev(_oracle_out(o, rc, state, k0, steps), ref_end)


def _dev(got, want):
    dm = float(np.abs(abar_edges(got[0]) - abar_edges(want[0])).max()) if len(got[0]) else 0.0
    return max(dm, float(np.abs(_sig64(got[3]) - _sig64(want[3])).max()))


def _oracle_out(o, rc, state, k0, steps):
    r, c = rc
    M, m, v, f, mf, vf = state
    o.M[r, c], o.M[c, r] = M[:, 0], M[:, 1]
    o.mM[:] = 0
    o.vM[:] = 0
    o.mM[r, c], o.mM[c, r] = m[:, 0], m[:, 1]
    o.vM[r, c], o.vM[c, r] = v[:, 0], v[:, 1]
    o.f, o.mf, o.vf = f.astype(np.float32).copy(), mf.astype(np.float32).copy(), vf.astype(np.float32).copy()
    o.step = k0
    o._gate_min = np.inf
    for _ in range(steps):
        o.iterate()
        o._gate_min = min(o._gate_min, gate_margin(o))
        o.M[o._off_edges] = o._M0[o._off_edges]      # dead entries (never reach an output): parked, so they cannot saturate the sigmoid
    return (np.stack([o.M[r, c], o.M[c, r]], 1), None, None, o.f.copy())


def classify_windows(sub_adj, sub_feat, sd, gt, pred_label, new_idx, mask0, rec, graph_mode, seed, force=()):
    """(cond50, sens50) [2][6] and, for the flagged windows, (cond10, sens10) [2][5] each (see the module docstring)."""
    from oracle import closed_form
    o = closed_form.ClosedFormOracle(sub_adj.astype(np.float32), sub_feat.astype(np.float32), sd, gt, pred_label, new_idx, mask0,
                                     graph_mode=graph_mode)
    rc = np.nonzero(np.triu(sub_adj, 1))
    o._off_edges, o._M0 = (sub_adj == 0), np.asarray(mask0, np.float32)
    n = sub_adj.shape[0]
    o._lvl = np.zeros(n, np.int64)
    if not graph_mode:       # hop level of every row from the target (off-diagonal pattern of the sub-adjacency)
        pat = (sub_adj != 0) & ~np.eye(n, dtype=bool)
        o._lvl[:] = 9
        o._lvl[new_idx] = 0
        for d in (1, 2):
            o._lvl[(pat[o._lvl == d - 1].sum(0) > 0) & (o._lvl > d)] = d
    E, D = len(rc[0]), sub_feat.shape[1]
    z2 = np.zeros((E, 2), np.float32)
    zd = np.zeros(D, np.float32)
    M0 = np.stack([mask0[rc[0], rc[1]], mask0[rc[1], rc[0]]], 1).astype(np.float32)
    state = lambda k: (M0, z2, z2, zd, zd, zd) if k == 0 else rec[k]
    cond50, cond10 = np.zeros((3, EPOCHS // WIN), np.float32), {}     # [0]: CPU vs CPU, [1]: 1-ulp sensitivity, [2]: smallest gate margin
    for w in range(EPOCHS // WIN):
        cond50[:, w] = _oracle_dev(o, rc, state(WIN * w), rec[WIN * (w + 1)], WIN * w, WIN, seed=(seed, w))
        if cond50[:2, w].max() > FLAG or cond50[2, w] < GATE or w in force:
            cond10[w] = np.asarray([_oracle_dev(o, rc, state(WIN * w + SUB * s), rec[WIN * w + SUB * (s + 1)], WIN * w + SUB * s, SUB,
                                                seed=(seed, w, s)) for s in range(WIN // SUB)], np.float32).T      # [3][5]
    return cond50, cond10


def _pack_target(key, rec, cond50, cond10, extra=None):
    out = dict(key=key, cond50=cond50, coarse=[rec[k] for k in range(WIN, EPOCHS + 1, WIN)], fine=[])
    for w, c10 in sorted(cond10.items()):
        out["fine"].append((w, c10, [rec[WIN * w + SUB * s] for s in range(1, WIN // SUB)]))
    if extra:
        out.update(extra)
    return out


def _node_worker(job):
    dataset, work, targets = job
    mg = _setup()
    import torch
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    args = mg.explain_args(dataset, work, EPOCHS)
    args.logdir = os.path.join(work, f"log_windows_{os.getpid()}")
    os.makedirs(args.logdir, exist_ok=True)
    with mg.quiet():
        ckpt = io_utils.load_ckpt(args)
    cg = ckpt["cg"]
    D, C = cg["feat"].shape[2], cg["pred"].shape[2]
    model = models.GcnEncoderNode(input_dim=D, hidden_dim=20, embedding_dim=20, label_dim=C, num_layers=3, bn=False, args=args)
    model.load_state_dict(ckpt["model_state"])
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in ckpt["model_state"].items()}
    fx = np.load(os.path.join(HERE, dataset + "_ckpt.npz"))     # the minted checkpoint must be the committed fixture's, bit for bit
    assert all(np.array_equal(fx["w:" + k], v) for k, v in sd.items()), "checkpoint differs from tests/golden/%s_ckpt.npz" % dataset
    snaps, rc_box = install_snapshots(explain)
    with mg.quiet():
        ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"],
                               train_idx=cg["train_idx"], args=args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    out = []
    for t in targets:
        with mg.quiet():
            new_idx, sub_adj, sub_feat, sub_label, nb = ex.extract_neighborhood(t)
            rc_box["rc"] = np.nonzero(np.triu(sub_adj, 1))
            torch.manual_seed(1000 + t)
            ma = ex.explain(t)
        mod, rec = snaps[-1]
        del snaps[:]
        r, c = rc_box["rc"]
        # the last snapshot is the state behind the reference's returned mask: masked_adj of the LAST forward = state after 299 steps
        # (not stored), so check the next best thing: every stored M is finite and the edge structure is the fixture's
        assert not np.isnan(ma).any() and all(np.isfinite(x[0]).all() for x in rec.values())
        pl = np.argmax(cg["pred"][0][nb], axis=1)
        force = {w for (tt, w) in FORCE_FINE.get(dataset, ()) if tt == int(t)}
        cond50, cond10 = classify_windows(sub_adj, sub_feat, sd, int(sub_label[new_idx]), pl, int(new_idx), mod.mask0.numpy(), rec, False, int(t), force)
        out.append(_pack_target(int(t), rec, cond50, cond10, dict(nedges=len(r))))
        for f in os.listdir(args.logdir):
            os.remove(os.path.join(args.logdir, f))
    return out


def _graph_worker(job):
    work, gids, wts = job
    mg = _setup()
    import torch
    import models
    from explainer import explain
    from gnn_model_explainer_amd.utils import synthetic
    import make_golden_full as mgf
    args = mg.explain_args("syn1", work, EPOCHS)
    args.bmname = "Mutagenicity"
    args.graph_mode = True
    args.logdir = os.path.join(work, f"log_windows_{os.getpid()}")
    os.makedirs(args.logdir, exist_ok=True)
    model = models.GcnEncoderGraph(input_dim=14, hidden_dim=20, embedding_dim=20, label_dim=2, num_layers=3, bn=False, args=args)
    model.load_state_dict({k: torch.tensor(v) for k, v in wts.items()})
    model.eval()
    A_all, X_all, n_all, y_all = synthetic.molecule_like_graphs(max(gids) + 1, seed=0)
    adj = torch.tensor(A_all[gids])
    feat = torch.tensor(X_all[gids])
    label = torch.tensor(y_all[gids], dtype=torch.long)
    with torch.no_grad():
        pred = model(feat, adj)[0].numpy()[None]
    snaps, rc_box = install_snapshots(explain)
    ex = explain.Explainer(model=model, adj=adj, feat=feat, label=label, pred=pred, train_idx=None, args=args, writer=None,
                           print_training=False, graph_mode=True, graph_idx=0)
    out = []
    for k, g in enumerate(gids):
        rc_box["rc"] = np.nonzero(np.triu(A_all[g], 1))
        with mg.quiet():
            torch.manual_seed(1000 + g)
            ma = ex.explain(node_idx=0, graph_idx=k, graph_mode=True)
        mod, rec = snaps[-1]
        del snaps[:]
        assert not np.isnan(ma).any()
        r, c = rc_box["rc"]
        fsig = torch.sigmoid(mod.feat_mask).detach().numpy()
        cm, cf = mgf._closed_form_dev(A_all[g], X_all[g], wts, int(y_all[g]), None, 0, mod.mask0.numpy(), ma, fsig, EPOCHS, graph_mode=True)
        cond50, cond10 = classify_windows(A_all[g], X_all[g], wts, int(y_all[g]), None, 0, mod.mask0.numpy(), rec, True, int(g))
        out.append(_pack_target(int(g), rec, cond50, cond10,
                                dict(nedges=len(r), vals=ma[r, c].astype(np.float32), fsig=fsig, cm=cm, cf=cf,
                                     maxm=float(mod.mask.detach().abs().max()), nn=int(n_all[g]))))
        for f in os.listdir(args.logdir):
            os.remove(os.path.join(args.logdir, f))
    return out


def assemble(res, id_name):
    res.sort(key=lambda r: r["key"])
    T = len(res)
    eoff = np.cumsum([0] + [r["nedges"] for r in res]).astype(np.int64)
    nck = EPOCHS // WIN
    cat = lambda i, j: np.concatenate([r["coarse"][i][j] for r in res]) if eoff[-1] else np.zeros((0, 2), np.float32)
    out = {id_name: np.asarray([r["key"] for r in res], np.int64), "eoff": eoff,
           "epochs": np.arange(WIN, EPOCHS + 1, WIN).astype(np.int64), "sub": np.int64(SUB), "flag": np.float64(FLAG),
           "cond50": np.stack([r["cond50"][0] for r in res]).astype(np.float32),
           "sens50": np.stack([r["cond50"][1] for r in res]).astype(np.float32), "trials": np.int64(TRIALS), "ulp": np.float64(ULP),
           "gate50": np.stack([r["cond50"][2] for r in res]).astype(np.float32), "gate": np.float64(GATE)}
    for j, nm in enumerate(("M", "m", "v")):
        out[nm] = np.stack([cat(i, j) for i in range(nck)]).astype(np.float32)
    for j, nm in ((3, "f"), (4, "mf"), (5, "vf")):
        out[nm] = np.stack([np.stack([r["coarse"][i][j] for r in res]) for i in range(nck)]).astype(np.float32)
    fine = [(k, w, c10, st) for k, r in enumerate(res) for w, c10, st in r["fine"]]
    nsub = WIN // SUB - 1
    out["fine_tw"] = np.asarray([(k, w) for k, w, _, _ in fine], np.int32).reshape(-1, 2)
    out["fine_off"] = np.cumsum([0] + [res[k]["nedges"] for k, _, _, _ in fine]).astype(np.int64)
    out["cond10"] = np.asarray([c10[0] for _, _, c10, _ in fine], np.float32).reshape(-1, WIN // SUB)
    out["sens10"] = np.asarray([c10[1] for _, _, c10, _ in fine], np.float32).reshape(-1, WIN // SUB)
    out["gate10"] = np.asarray([c10[2] for _, _, c10, _ in fine], np.float32).reshape(-1, WIN // SUB)
    D = res[0]["coarse"][0][3].shape[0]
    for j, nm in enumerate(("fine_M", "fine_m", "fine_v")):
        out[nm] = (np.stack([np.concatenate([st[s][j] for _, _, _, st in fine]) for s in range(nsub)]).astype(np.float32)
                   if fine else np.zeros((nsub, 0, 2), np.float32))
    for j, nm in ((3, "fine_f"), (4, "fine_mf"), (5, "fine_vf")):
        out[nm] = (np.stack([np.stack([st[s][j] for _, _, _, st in fine]) for s in range(nsub)]).astype(np.float32)
                   if fine else np.zeros((nsub, 0, D), np.float32))
    return out


def _report(name, out, t0):
    c50, c10 = np.maximum(out["cond50"], out["sens50"]), np.maximum(out["cond10"], out["sens10"])
    c50 = np.where(out["gate50"] < GATE, np.maximum(c50, 1.0), c50)          # a gate inside round-off of zero flags the window whatever the probes saw
    c10 = np.where(out["gate10"] < GATE, np.maximum(c10, 1.0), c10)
    print(f"{name}: CPU vs CPU alone flags {int((out['cond50'] > FLAG).sum())} windows, the 1-ulp sensitivity probe alone {int((out['sens50'] > FLAG).sum())}, "
          f"a gate / pool margin below {GATE:g} alone {int((out['gate50'] < GATE).sum())}")
    T, W = c50.shape
    fl = c50 > FLAG
    print(f"{name}: {T} targets x {W} windows in {time.time() - t0:.0f} s; flagged 50-epoch windows (CPU vs CPU or 1-ulp sensitivity > 2e-6, or a gate margin < 5e-7): {int(fl.sum())} of {T * W} "
          f"({int((c50 > 1e-5).sum())} > 1e-5, max {c50.max():.2e}; per window {fl.sum(0).tolist()}); targets with a flagged window: "
          f"{int(fl.any(1).sum())}; their 10-epoch sub-windows: {int((c10 > FLAG).sum())} of {c10.size} > 2e-6, {int((c10 > 1e-5).sum())} > 1e-5 "
          f"(max {c10.max() if c10.size else 0:.2e})", flush=True)


def node_windows(dataset, work, procs, limit=None):
    import torch
    ck = torch.load(os.path.join(work, "ckpt", f"{dataset}_base_h20_o20.pth.tar"), weights_only=False)
    N = ck["cg"]["adj"].shape[1]
    targets = list(range(MOTIF_START[dataset], N))[:limit]
    jobs = [(dataset, work, targets[k::procs * 4]) for k in range(procs * 4)]
    jobs = [j for j in jobs if j[2]]
    t0 = time.time()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = [r for part in pool.map(_node_worker, jobs) for r in part]
    out = assemble(res, "targets")
    full = np.load(os.path.join(HERE, dataset + "_full_explain.npz"))
    if limit is None:
        assert np.array_equal(out["targets"], full["targets"]) and np.array_equal(out["eoff"], full["eoff"])
    np.savez_compressed(os.path.join(HERE, dataset + "_windows.npz"), **out)
    _report(dataset, out, t0)


def config4_windows(work, procs, num=512, total=4337, limit=None):
    """512 of the 4337 graphs of the config-4 job, size-stratified (every 8th or 9th graph in order of node count), the same model
    as config4_explain.npz (tests/golden/make_golden_full.py: GcnEncoderGraph, torch.manual_seed(0), N(0, 0.1) biases)."""
    mg = _setup()
    import torch
    import models
    from gnn_model_explainer_amd.utils import synthetic
    args = mg.explain_args("syn1", work, EPOCHS)
    torch.manual_seed(0)
    model = models.GcnEncoderGraph(input_dim=14, hidden_dim=20, embedding_dim=20, label_dim=2, num_layers=3, bn=False, args=args)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    wts = {k: v.detach().numpy().astype(np.float32) for k, v in model.state_dict().items()}
    old = np.load(os.path.join(HERE, "config4_explain.npz"))
    assert all(np.array_equal(old["w:" + k], v) for k, v in wts.items()), "model differs from config4_explain.npz"
    _, _, nn, _ = synthetic.molecule_like_graphs(total, seed=0)
    order = np.argsort(nn, kind="stable")
    gids = sorted(int(g) for g in order[np.linspace(0, total - 1, num).astype(int)])[:limit]
    jobs = [(work, gids[k::procs * 4], wts) for k in range(procs * 4)]
    jobs = [j for j in jobs if j[1]]
    t0 = time.time()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = [r for part in pool.map(_graph_worker, jobs) for r in part]
    out = assemble(res, "graphs")
    out.update(total_graphs=np.int64(total), num_nodes=np.asarray([r["nn"] for r in res], np.int32),
               vals=np.concatenate([r["vals"] for r in res]), feat_sig=np.stack([r["fsig"] for r in res]).astype(np.float32),
               cond_mask=np.asarray([r["cm"] for r in res], np.float32), cond_feat=np.asarray([r["cf"] for r in res], np.float32),
               max_abs_mask=np.asarray([r["maxm"] for r in res], np.float32), full_epochs=np.int64(EPOCHS))
    for k, v in wts.items():
        out["w:" + k] = v
    np.savez_compressed(os.path.join(HERE, "config4_windows.npz"), **out)
    _report("config4", out, t0)
    cm = np.maximum(out["cond_mask"], out["cond_feat"])
    print(f"config4: full horizon, closed form vs reference: {int((cm <= 2e-6).sum())} of {len(cm)} graphs <= 2e-6, {int((cm > 1e-5).sum())} > 1e-5", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gw/work", help="directory holding ckpt/ minted by make_golden.mint_checkpoint")
    ap.add_argument("--what", default="syn1,syn4,syn5,config4")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args()
    what = a.what.split(",")
    if not os.path.exists(os.path.join(a.work, "ckpt", "syn1_base_h20_o20.pth.tar")):
        mg = _setup()
        os.makedirs(a.work, exist_ok=True)
        for ds in ("syn1", "syn4", "syn5"):
            mg.mint_checkpoint(ds, a.work)
    for ds in ("syn1", "syn4", "syn5"):
        if ds in what:
            if not os.path.exists(os.path.join(a.work, "ckpt", f"{ds}_base_h20_o20.pth.tar")):
                _setup().mint_checkpoint(ds, a.work)
            node_windows(ds, a.work, a.procs, a.limit)
    if "config4" in what:
        config4_windows(a.work, a.procs, limit=a.limit)


if __name__ == "__main__":
    main()
