"""The logging form of the edge-sparse resident kernel (loss scalars of explain.py:808-819 without leaving the kernel the headline is
measured on, the decision trace of gnnx_set_trace) and the constant-feature form (gnnx_plan_analyze_features), emulator + GPU."""
import os

import numpy as np
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, Subgraph
from oracle import closed_form
from test_emu_kernels import _Backend, _node_case


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def _largest_syn1_case():
    ck = helpers.load_ckpt("syn1")
    full = np.load(os.path.join(helpers.GOLDEN, "syn1_full_explain.npz"))
    k = int(np.argmax(np.diff(full["nb_off"])))
    t = int(full["targets"][k])
    nb = full["nb_flat"][full["nb_off"][k]:full["nb_off"][k + 1]].astype(np.int64)
    A, X, lab, yhat = helpers.subgraph(ck, nb)
    new = int(full["node_idx_new"][k])
    return Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy())


def _levels(sg):
    n = sg.adj.shape[0]
    pat = (sg.adj != 0) & ~np.eye(n, dtype=bool)
    lvl = np.full(n, 9)
    lvl[sg.target_row] = 0
    for d in (1, 2):
        lvl[(pat[lvl == d - 1].sum(0) > 0) & (lvl > d)] = d
    return lvl


def _check_density_and_pred(loss, o):
    """What the reference prints next to the loss every epoch (explain.py:148-159): ExplainModule.mask_density AFTER the epoch's step (slot 5 of the
    kernels' log) and the class probabilities of the epoch's forward (slots 8 ...) against the closed form (pinned to the live reference's
    values by test_closed_form_density_and_pred_vs_live_reference)."""
    dens = np.asarray([x[0] for x in o.trace_log])
    pred = np.stack([x[1] for x in o.trace_log])
    C = min(pred.shape[1], 8)
    assert np.allclose(loss[:, 5], dens, rtol=2e-5, atol=1e-7), (loss[:, 5], dens)
    assert np.abs(loss[:, 8:8 + C] - pred[:, :C]).max() < 2e-6, (loss[:, 8:8 + C], pred[:, :C])


def test_closed_form_density_and_pred_vs_live_reference():
    """tests/golden/logging_explain.npz (make_golden_logging.py: the LIVE reference's per-epoch mask density and class probabilities, n = 6 / 48 / 310):
    the closed form reproduces them - the density is the one of the UPDATED mask (explain.py:142-148)."""
    z = np.load(os.path.join(helpers.GOLDEN, "logging_explain.npz"))
    ck = helpers.load_ckpt("syn1")
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    for t in (int(v) for v in z["targets"]):
        nb = idx.neighbors_batch(np.asarray([t]))[0]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(np.searchsorted(nb, t))
        o = closed_form.ClosedFormOracle(A, X, ck["sd"], int(lab[new]), yhat, new, helpers.seeded_mask0(t, len(nb)).numpy())
        for _ in range(int(z["epochs"])):
            o.iterate()
        dens = np.asarray([x[0] for x in o.trace_log])
        pred = np.stack([x[1] for x in o.trace_log])
        assert np.abs(dens - z[f"{t}:density"]).max() < 1e-6 and np.abs(pred - z[f"{t}:pred"]).max() < 2e-6
        assert np.allclose(np.asarray(o.trace)[:, 0], z[f"{t}:loss"], rtol=2e-6)


def _gate_words(U):
    return ((U > 0) * (1 << np.arange(U.shape[1]))).sum(1).astype(np.uint32)


def test_logging_form_mixed_launch_loss_and_gates_vs_closed_form(be, monkeypatch):
    """One mixed launch (64-, 256- and 512-thread classes, n = 6 / 48 / 310) with loss logging and the decision trace switched on: the
    five loss terms of every iteration agree with the closed form (the size and entropy sums include the n^2 - 2E entries OFF the edges,
    advanced by k_dead_entries), the gate words are the closed form's signs of U1 (rows within two hops) and U2 (the target and its
    neighbours), and the masks are those of the plain run of the general form (which the logging form is) bit for bit."""
    ck = helpers.load_ckpt("syn1")
    subs = [_node_case("syn1", 302)[2], _node_case("syn1", 309)[2], _largest_syn1_case()]
    iters = 5
    monkeypatch.setenv("GNNX_XCONST", "1")       # the plain run below: the general form's arithmetic (0 and 1 are bit-identical; the default, 2, is not)
    job = be.job(subs, ck["sd"])
    assert list(job.route()) == [6, 5, 8]      # (n = 48: the 256-thread class, a pair workgroup with one body - its logging form)
    plain = job.run([s.mask0 for s in subs], Hyper(num_iters=iters))
    job.set_masks([s.mask0 for s in subs])
    hy = Hyper(num_iters=iters, record_loss=True)
    job.launch(hy, trace=True)
    res = job.fetch(hy)
    gates, pool = job.fetch_trace()
    assert pool is None
    for i, sg in enumerate(subs):
        assert np.array_equal(res.masked_adj[i], plain.masked_adj[i]) and np.array_equal(res.feat_mask[i], plain.feat_mask[i])
        o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
        lvl = _levels(sg)
        for it in range(iters):
            o.iterate()
            for l, lim in ((0, 2), (1, 1)):
                want = _gate_words(o.stages["U"][l])
                want[lvl > lim] = 0
                assert np.array_equal(want, gates[i][it, :, l]), (i, it, l)
        tr = np.asarray(o.trace)            # total, pred, size, lap, ent, feat_size
        assert np.allclose(res.loss[i][:, :5], tr[:, 1:6], rtol=2e-5, atol=1e-7)
        _check_density_and_pred(res.loss[i], o)
        assert np.abs(res.mask[i] - o.M).max() < 2e-5          # every entry of M, on and off the edges


def test_logging_form_of_the_large_target_kernel_loss_and_gates_vs_closed_form(be):
    """Route 7 (k_sparse_large: n = 900, far edges run their closed recursions outside the loop) with loss logging and the decision trace:
    the five loss terms of every iteration agree with the closed form - the size / entropy / Laplacian sums are completed by the far edges
    (float atomics from their recursions) and by the entries off the edges (k_dead_entries) -, the gate words are the closed form's signs on
    the rows within two hops (layer 1) and on the target and its neighbours (layer 2), the masks are the plain run's bit for bit, and every
    entry of M (on and off the edges) is the dense optimisation's."""
    from test_emu_kernels import _hub_case
    sd, subs = _hub_case()
    iters = 4
    job = be.job(subs, sd)
    assert list(job.route()) == [7]
    plain = job.run([s.mask0 for s in subs], Hyper(num_iters=iters))
    job.set_masks([s.mask0 for s in subs])
    hy = Hyper(num_iters=iters, record_loss=True)
    job.launch(hy, trace=True)
    res = job.fetch(hy)
    gates, pool = job.fetch_trace()
    sg = subs[0]
    assert np.array_equal(res.masked_adj[0], plain.masked_adj[0]) and np.array_equal(res.feat_mask[0], plain.feat_mask[0])
    o = closed_form.ClosedFormOracle(sg.adj, sg.feat, sd, sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
    lvl = _levels(sg)
    for it in range(iters):
        o.iterate()
        for l, lim in ((0, 2), (1, 1)):
            want = _gate_words(o.stages["U"][l])
            want[lvl > lim] = 0
            assert np.array_equal(want, gates[0][it, :, l]), (it, l)
    tr = np.asarray(o.trace)            # total, pred, size, lap, ent, feat_size
    assert np.allclose(res.loss[0][:, :5], tr[:, 1:6], rtol=5e-5, atol=1e-7), (res.loss[0][:, :5], tr[:, 1:6])
    _check_density_and_pred(res.loss[0], o)
    assert np.abs(res.mask[0] - o.M).max() < 2e-5


def test_logging_form_graph_mode_loss_gates_and_pool_rows(be):
    z = np.load(helpers.GOLDEN + "/graphmode_explain.npz")
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    subs = [Subgraph(z["adj"][g], z["feat"][g], int(z["label"][g]), 0, None, helpers.seeded_mask0(g, 100).numpy()) for g in (1, 4)]
    iters = 5
    job = be.job(subs, sd, graph_mode=True)
    assert set(job.route()) <= {4, 5, 6}
    job.set_masks([s.mask0 for s in subs])
    hy = Hyper(num_iters=iters, record_loss=True)
    job.launch(hy, trace=True)
    res = job.fetch(hy)
    gates, pool = job.fetch_trace()
    for i, sg in enumerate(subs):
        o = closed_form.ClosedFormOracle(sg.adj, sg.feat, sd, sg.gt_label, None, 0, sg.mask0, graph_mode=True)
        for it in range(iters):
            o.iterate()
            U = o.stages["U"]
            for l in (0, 1):
                assert np.array_equal(_gate_words(U[l]), gates[i][it, :, l])
            for l, a in enumerate((np.maximum(U[0], 0), np.maximum(U[1], 0), U[2])):
                assert np.array_equal(a.argmax(0), pool[i, it, l, :a.shape[1]])       # first maximal row, like torch.max
                assert (pool[i, it, l, a.shape[1]:] == -1).all()
        tr = np.asarray(o.trace)
        assert np.allclose(res.loss[i][:, [0, 1, 3, 4]], tr[:, [1, 2, 4, 5]], rtol=2e-5, atol=1e-7) and (res.loss[i][:, 2] == 0).all()
        _check_density_and_pred(res.loss[i], o)


def test_logged_loss_is_the_streaming_kernels_loss(be):
    """Same targets, loss logging on the resident route and on the dense streaming kernels (which update every entry of M)."""
    ck = helpers.load_ckpt("syn1")
    subs = [_node_case("syn1", 302)[2], _node_case("syn1", 309)[2]]
    out = []
    for use_resident in (True, False):
        job = be.job(subs, ck["sd"])
        out.append(job.run([s.mask0 for s in subs], Hyper(num_iters=8, record_loss=True, use_resident=use_resident)))
    for i in range(len(subs)):
        keep = [0, 1, 2, 3, 4, 5] + list(range(8, 16))       # (6, 7: the streaming kernels' numerator / denominator of the density)
        assert np.allclose(out[0].loss[i][:, keep], out[1].loss[i][:, keep], rtol=2e-5, atol=1e-7)
        assert (out[0].loss[i][:, 5] > 0).all() and (out[0].loss[i][:, 8:12].sum(1) > 0.99).all()
        assert np.abs(out[0].mask[i] - out[1].mask[i]).max() < 2e-5


def test_trace_needs_the_sparse_resident_kernel(be):
    ck, gx, sg = _node_case("syn1", 309)
    job = be.job([sg], ck["sd"], analyze=False)          # dense resident / streaming plan
    job.set_masks([sg.mask0])
    with pytest.raises(RuntimeError, match="decision trace"):
        job.launch(Hyper(num_iters=2), trace=True)
    job.lib.gnnx_set_trace(job.handle, None, None)


def test_constant_feature_form_is_bit_identical_to_the_general_form(be, monkeypatch):
    """syn1 / syn4 features are constant rows (gengraph.py: ConstFeatureGen): the plan picks a constant-feature form
    (gnnx_plan_analyze_features).  GNNX_XCONST=1: the general form's products in the general form's order without the gathers - every
    output bit-equal to GNNX_XCONST=0 (the general form); the default, 2, is the algebraic form (layer 1 as a function of the row sums of
    the masked adjacency): same mathematics, other rounding - within round-off of the others and of the closed form."""
    ck = helpers.load_ckpt("syn1")
    subs = [_node_case("syn1", 302)[2], _node_case("syn1", 309)[2], _largest_syn1_case()]
    outs = {}
    for flag in ("1", "0", "2"):
        monkeypatch.setenv("GNNX_XCONST", flag)
        job = be.job(subs, ck["sd"])
        job.set_masks([s.mask0 for s in subs])
        job.launch(Hyper(num_iters=6), keep_state=True)
        em = job.fetch_edges(with_mask=True)
        outs[flag] = (em.masked_adj, em.mask_rc, em.feat_mask) + job.fetch_state_edges()[1:]
    for a, b in zip(outs["1"], outs["0"]):
        assert np.array_equal(a, b)
    assert not np.array_equal(outs["2"][1], outs["0"][1])          # (another rounding: the form really ran)
    for a, b, tol in zip(outs["2"], outs["0"], (1e-6, 1e-5, 1e-5, 1e-6, 1e-8, 1e-5)):
        assert np.abs(a.astype(np.float64) - b).max() < tol
    eoff = np.concatenate([[0], np.cumsum([int((np.triu(s.adj, 1) != 0).sum()) for s in subs])])
    for i, sg in enumerate(subs):
        o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
        want = o.run(6)
        r, c = np.nonzero(np.triu(sg.adj, 1))
        assert np.abs(outs["2"][0][eoff[i]:eoff[i + 1]] - want[r, c]).max() < 2e-6


def test_constant_feature_form_refuses_other_features_loudly(be):
    """The plan's promise is about the X it looked at; handed other features at run time the form must not silently use row 0."""
    ck, gx, sg = _node_case("syn1", 309)
    job = be.job([sg], ck["sd"])
    job.set_masks([sg.mask0])
    job.X[int(job.offR[0]) + 3, 2] = 0.5          # one feature of one node differs now
    job.launch(Hyper(num_iters=2))
    res = job.fetch(Hyper(num_iters=2))
    assert np.isnan(res.masked_adj[0]).all()
    job.analyze()                                  # the plan looks again: general form
    job.set_masks([sg.mask0])
    res = job.run([sg.mask0], Hyper(num_iters=2))
    X = sg.feat.copy()
    X[3, 2] = 0.5
    o = closed_form.ClosedFormOracle(sg.adj, X, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
    assert np.abs(res.masked_adj[0] - o.run(2)).max() < 2e-6
