#!/usr/bin/env python
"""Drift per epoch inside a teacher-forced window: engine vs the LIVE reference, next to closed form (CPU) vs the live reference.

VERDICT r4 #4a.  The decision-conditional parity test (tests/test_decision_parity.py) accepts a handful of Tree-Grid (syn5) windows on
a conditioning argument: five windows whose first differing decision the reference takes by 2e-6 .. 2e-5, and a dozen windows in which
every decision agrees and the engine still ends 1e-5 .. 7.5e-5 from the reference's state (all of it in sigmoid(feat_mask)).  This tool
shows WHERE inside such a window the distance opens, epoch by epoch, for two independent implementations started from the same state of
the reference at the window's first epoch:

    engine   - the product's kernels (the HIP emulator build of the same sources by default, `--backend gpu` on an MI355X), resumed one
               epoch at a time through gnnx_run_resume (a run split into resumed segments is bit-identical to the straight run);
    closed   - oracle/closed_form.py, the NumPy fp32 restatement (another summation order than both the reference and the kernels).

and the reference's own margin at its smallest ReLU gate of every epoch (the decisions fixture's NEAR list).  If the CPU pair drifts the
way the engine does - same epochs, same order of magnitude - the window amplifies round-off as such; if only the engine leaves, the
kernel is the difference.  Runs in the build container only (it imports /root/reference, unmodified, through the harness of
tests/golden/make_golden_windows.py, snapshotting the reference's optimiser after EVERY step).

    python tools/drift_per_epoch.py --dataset syn5 --windows 1034:2,896:1,1195:4,533:4,881:3 --out profiles/r05_syn5_drift_per_epoch.txt
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def reference_states(dataset, targets, work):
    """-> {target: (rec {step: (M, m, v [E, 2], f, mf, vf [D])}, rc, sub_adj, sub_feat, gt, pred_label, new_idx, mask0)} from the live reference"""
    import make_golden_windows as mw
    mw.SUB = 1                                   # snapshot after every optimiser step
    mg = mw._setup()
    import torch
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    if not os.path.exists(os.path.join(work, "ckpt")):
        mg.mint_checkpoint(dataset, work)
    args = mg.explain_args(dataset, work, mw.EPOCHS)
    args.logdir = os.path.join(work, "log_drift")
    os.makedirs(args.logdir, exist_ok=True)
    with mg.quiet():
        ckpt = io_utils.load_ckpt(args)
    cg = ckpt["cg"]
    D, C = cg["feat"].shape[2], cg["pred"].shape[2]
    model = models.GcnEncoderNode(input_dim=D, hidden_dim=20, embedding_dim=20, label_dim=C, num_layers=3, bn=False, args=args)
    model.load_state_dict(ckpt["model_state"])
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in ckpt["model_state"].items()}
    fx = np.load(os.path.join(ROOT, "tests", "golden", dataset + "_ckpt.npz"))
    assert all(np.array_equal(fx["w:" + k], v) for k, v in sd.items()), "checkpoint differs from the committed fixture"
    snaps, rc_box = mw.install_snapshots(explain)
    with mg.quiet():
        ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"], train_idx=cg["train_idx"],
                               args=args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    out = {}
    for t in targets:
        with mg.quiet():
            new_idx, sub_adj, sub_feat, sub_label, nb = ex.extract_neighborhood(t)
            rc_box["rc"] = np.nonzero(np.triu(sub_adj, 1))
            torch.manual_seed(1000 + t)
            ex.explain(t)
        mod, rec = snaps[-1]
        del snaps[:]
        pl = np.argmax(cg["pred"][0][nb], axis=1)
        out[t] = (dict(rec), rc_box["rc"], sub_adj.astype(np.float32), sub_feat.astype(np.float32), int(sub_label[new_idx]), pl, int(new_idx),
                  mod.mask0.numpy().copy())
    return out, sd


def sig(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def dist(Mrc, f, ref):
    """(masked-adjacency distance on the edges, sigmoid(feat_mask) distance) to a reference snapshot"""
    a = lambda M: 0.5 * (sig(M[:, 0]) + sig(M[:, 1]))
    dm = float(np.abs(a(Mrc) - a(ref[0])).max()) if len(Mrc) else 0.0
    return dm, float(np.abs(sig(f) - sig(ref[3])).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="syn5")
    ap.add_argument("--windows", required=True, help="comma-separated target:window pairs (50-epoch windows, 0..5)")
    ap.add_argument("--backend", default="emu", choices=["emu", "gpu"])
    ap.add_argument("--work", default=None)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    pairs = [(int(a), int(b)) for a, b in (x.split(":") for x in args.windows.split(","))]
    work = args.work or os.path.join(tempfile.gettempdir(), "gnnx_drift_" + args.dataset)
    os.makedirs(work, exist_ok=True)
    ref, sd = reference_states(args.dataset, sorted({t for t, _ in pairs}), work)

    import helpers
    from gnn_model_explainer_amd import engine
    from gnn_model_explainer_amd.engine import Hyper, Subgraph
    from oracle import closed_form
    import make_golden_windows as mw
    from test_emu_kernels import _Backend
    be = _Backend(args.backend)
    Dn = helpers.Decisions(args.dataset)
    lines = []
    emit = lambda s="": (lines.append(s), print(s))
    emit(f"# tools/drift_per_epoch.py --dataset {args.dataset} --windows {args.windows} --backend {args.backend}")
    emit("# columns: epoch (steps taken) | engine vs reference: masked adjacency, sigmoid(feat_mask) | closed form (CPU) vs reference: the same two | "
         "the reference's smallest |U| at a ReLU gate that reaches the loss in that epoch's forward (its NEAR list: below 1e-4, else '-')")
    for t, w in pairs:
        rec, rc, A, X, gt, pl, new_idx, mask0 = ref[t]
        k0 = 50 * w
        E = len(rc[0])
        z2, zd = np.zeros((E, 2), np.float32), np.zeros(X.shape[1], np.float32)
        M0 = np.stack([mask0[rc[0], rc[1]], mask0[rc[1], rc[0]]], 1).astype(np.float32)
        start = (M0, z2, z2, zd, zd, zd) if k0 == 0 else rec[k0]
        # engine: one resumed epoch at a time
        sg = Subgraph(A, X, gt, new_idx, pl, mask0)
        job = be.job([sg], sd)
        job.set_masks([mask0])
        st = job.set_state_edges(k0, *start) if k0 else None
        eng = []
        for e in range(50):
            job.launch(Hyper(num_iters=1), state=st, keep_state=True)
            st = job.state_out
            Mrc, _, _, fs = job.fetch_state_edges()
            eng.append(dist(Mrc, fs[0, 0, :], rec[k0 + e + 1]))
        # closed form from the same state
        o = closed_form.ClosedFormOracle(A, X, sd, gt, pl, new_idx, mask0)
        o._off_edges, o._M0 = (A == 0), np.asarray(mask0, np.float32)
        o._lvl = np.zeros(A.shape[0], np.int64)
        r, c = rc
        M, m, v, f, mf, vf = start
        o.M[r, c], o.M[c, r] = M[:, 0], M[:, 1]
        o.mM[:] = 0
        o.vM[:] = 0
        o.mM[r, c], o.mM[c, r] = m[:, 0], m[:, 1]
        o.vM[r, c], o.vM[c, r] = v[:, 0], v[:, 1]
        o.f, o.mf, o.vf = f.astype(np.float32).copy(), mf.astype(np.float32).copy(), vf.astype(np.float32).copy()
        o.step = k0
        cf = []
        for e in range(50):
            o.iterate()
            o.M[o._off_edges] = o._M0[o._off_edges]
            cf.append(dist(np.stack([o.M[r, c], o.M[c, r]], 1), o.f, rec[k0 + e + 1]))
        kidx = int(np.nonzero(Dn.ids == t)[0][0])
        near = Dn.near_gates(kidx)
        emit(f"\n## {args.dataset} target {t} window {w} (epochs {k0} .. {k0 + 50}; n = {A.shape[0]}, {E} edges)")
        for e in range(50):
            mg_ = [abs(val) for (ep, l, rr, cc), val in near.items() if ep == k0 + e]
            emit(f"{k0 + e + 1:4d} | {eng[e][0]:.2e} {eng[e][1]:.2e} | {cf[e][0]:.2e} {cf[e][1]:.2e} | {min(mg_):.2e}" if mg_ else
                 f"{k0 + e + 1:4d} | {eng[e][0]:.2e} {eng[e][1]:.2e} | {cf[e][0]:.2e} {cf[e][1]:.2e} | -")
        emit(f"# end of window: engine {max(eng[-1]):.2e}, closed form {max(cf[-1]):.2e}; largest along the window: engine {max(max(x) for x in eng):.2e}, "
             f"closed form {max(max(x) for x in cf):.2e}")
    if args.out:
        with open(args.out, "w") as fh:
            fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
