#!/usr/bin/env python
"""Fourth conditioning probe of the windowed fixtures: how much does a window of the reference's trajectory amplify PER-STEP round-off?

make_golden_windows.py classifies a window with (i) the CPU-vs-CPU deviation and (ii) the deviation of four runs whose START is
perturbed by +-1 ulp.  tests/test_decision_parity.py (round 4) showed what those under-sample: a dozen 50-epoch Tree-Grid windows in
which the engine takes the reference's side of every ReLU gate at every epoch and still ends 1e-5 .. 7e-5 away (probes (i) / (ii):
2e-7 .. 1.5e-6) - smooth stretches of a loss plateau on which Adam's scale-free step keeps amplifying the round-off that every
iteration adds, not only the one difference at the start.  This probe measures exactly that, on the CPU alone and from the
committed fixtures alone (no reference run needed: the window boundaries ARE the reference's states): the closed-form fp32 oracle is
started from the reference's state at a boundary and run over the window TRIALS times with +-1 ulp of relative noise on every entry of
the gradient in every iteration (oracle/closed_form.py `grad_noise`: what a different summation order does); noise50[t][w] is the largest
deviation (masked adjacency on the edges, sigmoid(feat_mask)) of those runs from the noise-free run; noise10 likewise for the 10-epoch
sub-windows of the windows that have fine snapshots.  Outcome-blind: no GPU result enters.
Round 5 adds ssens50 / ssens10, the 1-ulp sensitivity of a window to its WHOLE starting state (see _probe): the completion of probe (ii),
which perturbs the mask entries only.

    python tests/golden/make_golden_noise_probe.py --what syn1,syn4,syn5,config4 --procs 7
-> tests/golden/<name>_noise.npz: targets|graphs [T], noise50 / ssens50 [T][6], noise10 / ssens10 [F][5] (rows of <name>_windows.npz's fine_tw), trials, ulp
"""
import os as _os
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):      # one BLAS thread per worker process: the workers are the parallelism
    _os.environ.setdefault(_k, "1")
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

TRIALS = 4


def _sig64(x):
    return 1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))


def _run(o, rc, state, k0, steps, rng):
    r, c = rc
    M, m, v, f, mf, vf = state
    o.M[:] = o._M0
    o.M[r, c], o.M[c, r] = M[:, 0], M[:, 1]
    o.mM[:] = 0
    o.vM[:] = 0
    o.mM[r, c], o.mM[c, r] = m[:, 0], m[:, 1]
    o.vM[r, c], o.vM[c, r] = v[:, 0], v[:, 1]
    o.f, o.mf, o.vf = f.astype(np.float32).copy(), mf.astype(np.float32).copy(), vf.astype(np.float32).copy()
    o.step = k0
    o.grad_noise = rng
    for _ in range(steps):
        o.iterate()
        o.M[o._off_edges] = o._M0[o._off_edges]      # dead entries (never reach an output): parked, so they cannot saturate the sigmoid
    Mrc = np.stack([o.M[r, c], o.M[c, r]], 1)
    return 0.5 * (_sig64(Mrc[:, 0]) + _sig64(Mrc[:, 1])), _sig64(o.f)


STATE_TRIALS = 8


def _probe(o, rc, state, k0, steps, seed):
    """-> (deviation under per-step gradient noise, 1-ulp sensitivity of the window to its WHOLE starting state)"""
    base = _run(o, rc, state, k0, steps, None)
    d = lambda got: max(float(np.abs(got[0] - base[0]).max()) if len(base[0]) else 0.0, float(np.abs(got[1] - base[1]).max()))
    dev = 0.0
    for trial in range(TRIALS):
        dev = max(dev, d(_run(o, rc, state, k0, steps, np.random.default_rng((seed, k0, trial)))))
    # Round 5 (ssens50 / ssens10).  Probe (ii) of make_golden_windows.py perturbs the MASK entries of the window's start by +-1 ulp; the
    # state an implementation is handed has five more parts - the two Adam moments, the feature mask and its moments.  tools/drift_per_epoch.py
    # showed the engine's distance in the expansive Tree-Grid windows opening with a one-ulp difference of a FEATURE-MASK parameter in the
    # window's first steps and growing at the window's own rate (the closed form's distance grows at the same rate from a later start), which
    # a perturbation of the mask entries alone does not see: half the trials perturb every part of the state, half the feature mask alone.
    sens = 0.0
    for trial in range(STATE_TRIALS):
        rng = np.random.default_rng((seed, k0, 7919 + trial))
        parts = range(6) if trial % 2 == 0 else (3,)
        st = list(state)
        for i in parts:
            st[i] = (np.asarray(st[i], np.float32) * (np.float32(1) + np.float32(2.0 ** -23) * (2 * rng.random(np.shape(st[i]), dtype=np.float32) - 1))).astype(np.float32)
        sens = max(sens, d(_run(o, rc, tuple(st), k0, steps, None)))
    return dev, sens


def _worker(job):
    name, ks = job
    import helpers
    from oracle import closed_form
    W = helpers.Windows(name)
    graph_mode = name == "config4"
    if graph_mode:
        from gnn_model_explainer_amd.utils import synthetic
        sd = {k[2:]: W.z[k] for k in W.z.files if k.startswith("w:")}
        A_all, X_all, nn, y_all = synthetic.molecule_like_graphs(int(W.ids.max()) + 1, seed=0)
    else:
        ck = helpers.load_ckpt(name)
        sd = ck["sd"]
        full = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    out = []
    for k in ks:
        ident = int(W.ids[k])
        if graph_mode:
            A, X, gt, pl, new = A_all[ident], X_all[ident], int(y_all[ident]), None, 0
        else:
            nb = full["nb_flat"][full["nb_off"][k]:full["nb_off"][k + 1]].astype(np.int64)
            A, X, lab, pl = helpers.subgraph(ck, nb)
            new = int(full["node_idx_new"][k])
            gt = int(lab[new])
        m0 = helpers.seeded_mask0(ident, A.shape[0]).numpy()
        o = closed_form.ClosedFormOracle(A.astype(np.float32), X.astype(np.float32), sd, gt, pl, new, m0, graph_mode=graph_mode)
        o._off_edges, o._M0 = (A == 0), np.asarray(m0, np.float32)
        rc = np.nonzero(np.triu(A, 1))
        E, D = len(rc[0]), X.shape[1]
        z2, zd = np.zeros((E, 2), np.float32), np.zeros(D, np.float32)
        M0 = np.stack([m0[rc[0], rc[1]], m0[rc[1], rc[0]]], 1).astype(np.float32)
        kk = np.asarray([k])

        def state(b, fine=None):
            st = W.boundary(b, kk) if fine is None else W.fine(fine[0], fine[1], kk)
            if st is None:
                return (M0, z2, z2, zd, zd, zd)
            return tuple(np.asarray(a[0] if i >= 4 else a) for i, a in enumerate(st[1:]))
        n50 = [_probe(o, rc, state(w), W.win * w, W.win, ident) for w in range(W.W)]
        n10 = {}
        for w in range(W.W):
            if (int(k), w) in W.fine_row:
                n10[w] = [_probe(o, rc, state(w) if s == 0 else state(None, (w, s)), W.win * w + W.sub * s, W.sub, ident) for s in range(W.nsub)]
        out.append((int(k), n50, n10))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="syn1,syn4,syn5,config4")
    ap.add_argument("--procs", type=int, default=7)
    ap.add_argument("--limit", type=int, default=None)
    a = ap.parse_args()
    import helpers
    for name in a.what.split(","):
        W = helpers.Windows(name)
        ks = list(range(W.T))[:a.limit]
        jobs = [(name, ks[i::a.procs * 8]) for i in range(a.procs * 8)]
        jobs = [j for j in jobs if j[1]]
        t0 = time.time()
        with mp.get_context("spawn").Pool(a.procs) as pool:
            res = [r for part in pool.map(_worker, jobs) for r in part]
        noise50 = np.zeros((W.T, W.W), np.float32)
        noise10 = np.zeros((len(W.z["fine_tw"]), W.nsub), np.float32)
        ssens50, ssens10 = np.zeros_like(noise50), np.zeros_like(noise10)      # round 5: 1-ulp sensitivity to the whole starting state
        for k, n50, n10 in res:
            noise50[k] = [x[0] for x in n50]
            ssens50[k] = [x[1] for x in n50]
            for w, v in n10.items():
                noise10[W.fine_row[(k, w)]] = [x[0] for x in v]
                ssens10[W.fine_row[(k, w)]] = [x[1] for x in v]
        id_name = "graphs" if name == "config4" else "targets"
        np.savez_compressed(os.path.join(HERE, name + "_noise.npz"), **{id_name: W.ids, "noise50": noise50, "noise10": noise10,
                                                                          "ssens50": ssens50, "ssens10": ssens10, "state_trials": np.int64(STATE_TRIALS),
                                                                          "trials": np.int64(TRIALS), "ulp": np.float64(2.0 ** -23)})
        print(f"{name}: whole-state sensitivity > 2e-6 in {int((ssens50 > 2e-6).sum())} windows, > 1e-5 in {int((ssens50 > 1e-5).sum())}, largest {float(ssens50.max()):.2e}", flush=True)
        old = np.maximum(W.z["cond50"], W.z["sens50"])
        done = np.zeros(W.T, bool)
        done[[r[0] for r in res]] = True
        print(f"{name}: {int(done.sum())} targets x {W.W} windows in {time.time() - t0:.0f} s; per-step-noise deviation > 2e-6 in {int((noise50[done] > 2e-6).sum())} windows "
              f"({int(((noise50 > 2e-6) & (old <= 2e-6))[done].sum())} of them not flagged by probes (i) / (ii); probes (i) / (ii) flag {int((old[done] > 2e-6).sum())}), "
              f"> 1e-5 in {int((noise50[done] > 1e-5).sum())}; 10-epoch sub-windows: {int((noise10 > 2e-6).sum())} of {noise10.size} > 2e-6", flush=True)


if __name__ == "__main__":
    main()
