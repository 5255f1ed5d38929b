"""The XL route against the LIVE reference on sub-graphs BEYOND 16 383 nodes (VERDICT r5 "next" 1: ">= 3 targets with n > 16 383 against the live
reference's ExplainModule state").

tests/golden/ba100k_xl_windows.npz / ba100k_xl_decisions.npz (make_golden_ba100k_xl.py): the reference's own ExplainModule (explain.py:582-820) run on the
three smallest sub-graphs beyond 16 383 nodes of a seed-fixed sample of ALL BA-House x100k nodes (n = 16 388, 16 440, 16 589: dense 1.08 GB tensors,
~25 s per epoch on this container's CPUs), its optimiser state snapshotted every 10 steps and every ReLU gate recorded at every epoch - 100 epochs on the
first target, 50 on the others.

  * CPU (`-m "not gpu"`): the fixture is self-consistent with the seed protocol - its initial mask on the edges is, bit for bit, what the engine's
    host walk draws for `torch.manual_seed(1000 + target)` (so the trajectories start from the state the engine starts from);
  * GPU: every window teacher-forced from the reference's state (the seeded start / its state after 50 steps): the engine's decisions are the
    reference's at every epoch and the window ends within 1e-5 of the reference's masked adjacency and sigmoid(feat_mask); the 10-epoch snapshots too."""
import os

import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper

FIX = os.path.join(helpers.GOLDEN, "ba100k_xl_windows.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(FIX), reason="tests/golden/ba100k_xl_windows.npz not generated")


def _graph():
    from gnn_model_explainer_amd.utils import synthetic
    ck = helpers.load_ckpt("syn1")
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, ck["sd"])
    return ck, csr, feat, pred, label


def test_fixture_starts_from_the_seed_protocols_masks():
    z = np.load(FIX)
    assert (z["size"] > 16383).all() and len(z["targets"]) >= 3
    ck, csr, feat, pred, label = _graph()
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    idx = KHopIndex(csr, 3)
    rcs = []
    for k, t in enumerate(z["targets"]):
        nb = idx.neighbors(int(t))
        assert np.array_equal(nb, z["nb_flat"][z["nb_off"][k]:z["nb_off"][k + 1]])
        sub = csr[nb][:, nb].tocoo()
        up = sub.row < sub.col
        order = np.lexsort((sub.col[up], sub.row[up]))
        rc = np.stack([sub.row[up][order], sub.col[up][order]], 1).astype(np.int32)
        assert len(rc) == int(z["eoff"][k + 1] - z["eoff"][k])
        rcs.append(rc)
    rc = np.concatenate(rcs)
    vals = engine.init_edge_masks_on_edges(z["size"], 1000 + z["targets"], z["eoff"], rc, threads=4)
    assert np.array_equal(vals.numpy(), z["mask0_rc"])


@pytest.mark.gpu
def test_xl_windows_and_decisions_against_the_live_reference():
    W = helpers.Windows("ba100k_xl")
    D = helpers.Decisions("ba100k_xl")
    z = W.z
    ck, csr, feat, pred, label = _graph()
    g = engine.device_graph(csr, feat, pred)
    targets = z["targets"].astype(np.int64)
    dn = engine.khop_device(g, targets, 3)
    assert np.array_equal(dn.sizes, z["size"]) and np.array_equal(dn.rows, z["node_idx_new"])
    xj = engine.XLJob(g, dn, None, label[targets], ck["sd"])
    eoff, rc = xj.edge_ids()
    assert np.array_equal(eoff, z["eoff"])
    assert np.array_equal(label[targets], z["gt_label"])
    xj.set_masks_seeded_device(1000 + targets)
    assert np.array_equal(xj.M_e[:xj.E].cpu().numpy(), z["mask0_rc"])          # the device engine walk == the reference's seeded draw, on 187 k edges
    have = z["have"]
    report = []
    for w in range(W.W):
        ks = [k for k in range(W.T) if have[w, k]]
        if not ks:
            continue
        # every target runs (one launch); targets without this window restart from their seeded masks and are not judged
        xj.reset_masks()
        start = None
        if w > 0:
            st = W.boundary(w, list(range(W.T)))
            M = xj.M_e[:xj.E].cpu().numpy()
            m = np.zeros_like(M)
            v = np.zeros_like(M)
            f = np.zeros((W.T, 10), np.float32)
            mf, vf = f.copy(), f.copy()
            for k in ks:
                a, b = int(eoff[k]), int(eoff[k + 1])
                M[a:b], m[a:b], v[a:b] = st[1][a:b], st[2][a:b], st[3][a:b]
                f[k], mf[k], vf[k] = st[4][k], st[5][k], st[6][k]
            # (one first_iter per launch: the judged targets are at 50 w; the others, restarted, are ignored)
            start = (W.win * w, M, m, v, f, mf, vf)
        for steps, what in ((10, "10-epoch snapshot"), (W.win, "window")):
            if start is not None:
                sx = xj.set_state_edges(*start)
            else:
                xj.reset_masks()
                sx = None
            xj.launch(Hyper(num_iters=steps), state=sx, keep_state=True, trace=(steps == W.win))
            mask_rc, _, _, fs = xj.fetch_state_edges()
            want = W.boundary(w + 1, list(range(W.T))) if steps == W.win else None
            for k in ks:
                a, b = int(eoff[k]), int(eoff[k + 1])
                if steps == W.win:
                    wm, wf = want[1][a:b], want[4][k]
                else:
                    i = W.fine_row[(k, w)]
                    fa, fb = int(z["fine_off"][i]), int(z["fine_off"][i + 1])
                    wm, wf = z["fine_M"][0][fa:fb], z["fine_f"][0][i]
                em = float(np.abs(helpers.abar_from_mask_rc(mask_rc[a:b]) - helpers.abar_from_mask_rc(wm)).max())
                ef = float(np.abs(helpers._sig64(fs[k, 0]) - helpers._sig64(wf)).max())
                report.append((int(targets[k]), w, what, em, ef))
                assert em <= helpers.WIN_TOL and ef <= helpers.WIN_TOL, (int(targets[k]), w, what, em, ef)
            if steps == W.win:
                gates, _ = xj.fetch_trace()
                for k in ks:
                    dis = D.first_disagreement(k, W.win * w, gates[k])
                    assert dis is None or dis[2] < D.near_tol_strict, (int(targets[k]), w, dis[:1], dis[2])
    print("\n".join("target %d window %d %s: masked adjacency %.2e, sigmoid(feat_mask) %.2e" % r for r in report))
