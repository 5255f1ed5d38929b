#!/usr/bin/env python
"""Windowed fixtures for the targets of BASELINE config 5 (BA-House x100k) BEYOND 16 383 sub-graph nodes - the class no edge-sparse kernel took
before round 6 (k_sparse_xl, gnnx_sparse_large.hpp) - from the LIVE reference.

Same construction as make_golden_ba100k_windows.py: the reference's own ExplainModule (explain.py:582-820, unmodified) on a sub-graph extracted
by sparse BFS, its optimiser state snapshotted every 10 steps and its ReLU gates recorded at every epoch.  A dense 16 400 x 16 400 fp32 tensor is
1.08 GB and the reference's autograd keeps ~25 of them alive: one target at a time, all cores, ~13 s per epoch here - hence 100 epochs (two
50-epoch windows) for the first target and one window for the others, not 300.  The targets: the three smallest sub-graphs beyond 16 383 nodes in a
seed-fixed sample of 3000 of ALL 99 997 nodes (they are BA nodes two hops from the graph's largest hub).

    python tests/golden/make_golden_ba100k_xl.py [--epochs 100,50,50] [--threads 6]            # about 45 minutes, ~30 GB of memory

Written: tests/golden/ba100k_xl_windows.npz (the layout of ba100k_windows.npz; `nepochs` [T] = epochs the reference ran on each target: windows
beyond it are absent - M / m / v rows are zero there and `have` [W][T] says so) and tests/golden/ba100k_xl_decisions.npz.
"""
import argparse
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

WIN = 50
N_MIN = 16383


def pick_targets(k=3):
    from gnn_model_explainer_amd.utils import synthetic
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    N, edges, _ = synthetic.ba_house(42857, 11428, seed=0)
    idx = KHopIndex(synthetic.csr_from_edges(N, edges), 3)
    rng = np.random.default_rng(5)
    samp = np.sort(rng.choice(N, 3000, replace=False))
    size = np.concatenate([idx.sizes(samp[b:b + 250]) for b in range(0, len(samp), 250)])
    big = np.nonzero(size > N_MIN)[0]
    big = big[np.argsort(size[big], kind="stable")][:k]
    return [int(samp[i]) for i in big], [int(size[i]) for i in big]


def run_target(t, epochs, work, threads):
    mg = mgw._setup()
    import torch
    torch.set_num_threads(threads)
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    import make_golden_full as mgf
    from gnn_model_explainer_amd.utils import synthetic
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    args = mg.explain_args("syn1", work, epochs)
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
    t0 = time.time()
    nb = idx.neighbors(t)
    new = int(np.searchsorted(nb, t))
    sub = idx.sub_adjacency(nb)
    n = len(nb)
    pl = np.argmax(pred[nb], axis=1)
    r, c = np.nonzero(np.triu(sub, 1))
    rc_box["rc"] = (r, c)
    pat = (sub != 0) & ~np.eye(n, dtype=bool)
    lvl = np.full(n, 9)
    lvl[new] = 0
    for d in (1, 2):
        lvl[(pat[lvl == d - 1].sum(0) > 0) & (lvl > d)] = d
    del pat
    recd.take()
    print(f"  target {t}: n={n} edges={len(r)} rows within two hops {int((lvl <= 2).sum())}; reference running {epochs} epochs ...", flush=True)
    ma, fsig, mask0, loss, maxm = mgf.reference_explain_subgraph(mg, model, sub, feat[nb], label[nb], pl, new, epochs, work, 1000 + t)
    mod, rec = snaps[-1]
    del snaps[:]
    pre, last = recd.take()
    mgd.EPOCHS = epochs              # (encode() walks that many epochs)
    piece = mgd.encode(pre, last, (lvl <= 2, lvl <= 1), False)
    piece["key"] = int(t)
    out = dict(key=int(t), nb=nb.astype(np.int32), new=new, nedges=len(r), epochs=epochs, rec={k: rec[k] for k in sorted(rec)},
               vals=ma[r, c].astype(np.float32), fsig=fsig, piece=piece, label=int(label[t]),
               mask0_rc=np.stack([mask0[r, c], mask0[c, r]], 1).astype(np.float32))
    print(f"  target {t}: loss={loss:.4f} max|M|={maxm:.2f} {time.time() - t0:.0f} s", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gw/work")
    ap.add_argument("--epochs", default="100,50,50")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--part-dir", default="/tmp/gw/xl_parts")
    a = ap.parse_args()
    eps = [int(x) for x in a.epochs.split(",")]
    if not os.path.exists(os.path.join(a.work, "ckpt", "syn1_base_h20_o20.pth.tar")):
        mg = mgw._setup()
        os.makedirs(a.work, exist_ok=True)
        mg.mint_checkpoint("syn1", a.work)      # the reference's own train.py, seeds fixed: the weights of tests/golden/syn1_ckpt.npz
    targets, sizes = pick_targets(len(eps))
    print("targets", targets, "sizes", sizes, flush=True)
    os.makedirs(a.part_dir, exist_ok=True)
    import pickle
    res = []
    for t, ep in zip(targets, eps):
        part = os.path.join(a.part_dir, f"{t}_{ep}.pkl")
        if os.path.exists(part):
            res.append(pickle.load(open(part, "rb")))
            continue
        # one target per PROCESS: the reference's dense tensors (tens of GB) are returned to the system before the next one starts
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(1) as pool:
            r = pool.apply(run_target, (t, ep, a.work, a.threads))
        pickle.dump(r, open(part, "wb"))
        res.append(r)
    res.sort(key=lambda r: r["key"])
    T = len(res)
    W = max(r["epochs"] for r in res) // WIN
    D = len(res[0]["fsig"])
    eoff = np.cumsum([0] + [r["nedges"] for r in res]).astype(np.int64)
    E = int(eoff[-1])
    out = dict(targets=np.asarray([r["key"] for r in res], np.int64), size=np.asarray([len(r["nb"]) for r in res], np.int32),
               nb_off=np.cumsum([0] + [len(r["nb"]) for r in res]).astype(np.int64), nb_flat=np.concatenate([r["nb"] for r in res]),
               node_idx_new=np.asarray([r["new"] for r in res], np.int32), eoff=eoff, epochs=np.arange(WIN, WIN * W + 1, WIN).astype(np.int64),
               sub=np.int64(10), nepochs=np.asarray([r["epochs"] for r in res], np.int32),
               gt_label=np.asarray([r["label"] for r in res], np.int32),
               vals=np.concatenate([r["vals"] for r in res]), feat_sig=np.stack([r["fsig"] for r in res]).astype(np.float32),
               mask0_rc=np.concatenate([r["mask0_rc"] for r in res]),
               probed=np.zeros(T, bool), cond50=np.zeros((T, W), np.float32), sens50=np.zeros((T, W), np.float32), noise50=np.zeros((T, W), np.float32))
    have = np.zeros((W, T), bool)
    for nm in ("M", "m", "v"):
        out[nm] = np.zeros((W, E, 2), np.float32)
    for nm in ("f", "mf", "vf"):
        out[nm] = np.zeros((W, T, D), np.float32)
    # the 10-epoch snapshots of every window the reference ran (fine_tw = (target index, window))
    fine_tw, fine = [], {nm: [[] for _ in range(4)] for nm in ("M", "m", "v", "f", "mf", "vf")}
    for k, r in enumerate(res):
        for w in range(r["epochs"] // WIN):
            have[w, k] = True
            st = r["rec"][WIN * (w + 1)]
            for j, nm in enumerate(("M", "m", "v")):
                out[nm][w, eoff[k]:eoff[k + 1]] = st[j]
            for j, nm in ((3, "f"), (4, "mf"), (5, "vf")):
                out[nm][w, k] = st[j]
            fine_tw.append((k, w))
            for s in range(1, 5):
                st = r["rec"][WIN * w + 10 * s]
                for j, nm in enumerate(("M", "m", "v", "f", "mf", "vf")):
                    fine[nm][s - 1].append(st[j])
    out["have"] = have
    out["fine_tw"] = np.asarray(fine_tw, np.int32).reshape(-1, 2)
    out["fine_off"] = np.cumsum([0] + [res[k]["nedges"] for k, _ in fine_tw]).astype(np.int64)
    for nm in ("M", "m", "v"):
        out["fine_" + nm] = np.stack([np.concatenate(fine[nm][s]) for s in range(4)]).astype(np.float32)
    for nm in ("f", "mf", "vf"):
        out["fine_" + nm] = np.stack([np.stack(fine[nm][s]) for s in range(4)]).astype(np.float32)
    nf = len(fine_tw)
    out.update(gate50=np.full((T, W), np.inf, np.float32), gate=np.float64(5e-7), cond10=np.zeros((nf, 5), np.float32),
               sens10=np.zeros((nf, 5), np.float32), gate10=np.full((nf, 5), np.inf, np.float32))
    np.savez_compressed(os.path.join(HERE, "ba100k_xl_windows.npz"), **out)
    mgd.EPOCHS = max(r["epochs"] for r in res)
    dec = mgd.assemble([r["piece"] for r in res], "targets", False)
    np.savez_compressed(os.path.join(HERE, "ba100k_xl_decisions.npz"), **dec)
    print(f"ba100k_xl: {T} targets, n = {out['size'].tolist()}, edges {np.diff(eoff).tolist()}, epochs {out['nepochs'].tolist()}", flush=True)


if __name__ == "__main__":
    main()
