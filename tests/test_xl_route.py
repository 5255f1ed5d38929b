"""The XL route (include/gnnx.h: gnnx_xl_*; engine.XLJob): node-mode targets of any size, CSR-native - k_xl_rowdeg / k_xl_rowptr / k_xl_emit build the
sub-graph CSRs from the resident graph, k_sparse_large<.., XL = true> runs the loop with its state in a global scratch block, results are edge lists.

One source, one arithmetic: on every target BOTH forms take, the XL form must reproduce route 7 (k_sparse_large, LDS form - pinned to the live
reference window by window, tests/test_windowed_parity.py / test_decision_parity.py) BIT FOR BIT - edge values, mask parameters, Adam moments, feature
mask, decision trace.  Every case on the emulator (`-m "not gpu"`) and on the GPU (`-m gpu`); the targets beyond route 7's range (n > 16 383) are judged
against the live reference itself in tests/test_xl_reference_windows.py."""
import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, Subgraph
from oracle import closed_form
from test_emu_kernels import _Backend


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def _edge_vals(m0, rc):
    return torch.from_numpy(np.stack([m0[rc[:, 0], rc[:, 1]], m0[rc[:, 1], rc[:, 0]]], 1).astype(np.float32).copy())


def _both(be, sgs, sd, iters, keep_state=False, trace=False):
    """-> (EdgeMasks of route 7, EdgeMasks of the XL form, the two jobs)"""
    job = be.job(sgs, sd)
    assert set(job.route()) == {7}, job.route()
    job.set_masks([s.mask0 for s in sgs])
    job.launch(Hyper(num_iters=iters, edge_results_only=True), keep_state=keep_state, trace=trace)
    em = job.fetch_edges(with_mask=True)
    xj = engine.xl_job_from_subgraphs(sgs, sd, device=be.device, lib=be.lib)
    eoff, rc = xj.edge_ids()
    rc = rc.cpu().numpy()
    assert np.array_equal(eoff, em.eoff) and np.array_equal(rc, em.rc)          # the edge order of gnnx_gather_edges
    vals = torch.cat([_edge_vals(s.mask0, rc[eoff[k]:eoff[k + 1]]) for k, s in enumerate(sgs)])
    xj.set_masks_on_edges(vals)
    xj.launch(Hyper(num_iters=iters), keep_state=keep_state, trace=trace)
    xe = xj.fetch_edges(with_mask=True)
    return em, xe, job, xj


def _same(em, xe):
    assert np.array_equal(xe.masked_adj, em.masked_adj), np.abs(xe.masked_adj - em.masked_adj).max()
    assert np.array_equal(xe.mask_rc, em.mask_rc)
    assert np.array_equal(xe.feat_mask, em.feat_mask)


def _hub_graph(rng, n, density, hub_deg, weighted=False):
    A, X = helpers.random_graph(rng, n, 10, density=density)
    hub = int(rng.integers(0, n))
    idx = rng.choice(np.arange(n), hub_deg, replace=False)
    idx = idx[idx != hub]
    A[hub, idx] = 1
    A[idx, hub] = 1
    if weighted:
        W = rng.uniform(0.25, 2.0, (n, n)).astype(np.float32)
        A = A * np.triu(W, 1)
        A = A + A.T + np.diag(rng.uniform(0.5, 1.5, n).astype(np.float32))
    return A, X, hub, idx


def test_xl_equals_route7_far_edges_and_closed_form(be):
    """n = 900, average degree 2.4: most edges are beyond two hops (their closed recursions), 12 iterations: XL == route 7 bit for bit, and the
    closed form on every edge."""
    rng = np.random.default_rng(17)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 900
    A, X = helpers.random_graph(rng, n, 10, density=2.4 / n)
    t = int(np.argmax(A.sum(1)))
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)
    em, xe, _, _ = _both(be, [sg], sd, 12)
    _same(em, xe)
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, t, m0)
    want = o.run(12)
    assert np.abs(xe.dense(0) - want).max() < 5e-6
    rc = xe.rc
    assert np.abs(xe.mask_rc[:, 0] - o.M[rc[:, 0], rc[:, 1]]).max() < 5e-5 and np.abs(xe.feat_mask[0] - o.f).max() < 5e-5


@pytest.mark.parametrize("iters", [1, 3])
def test_xl_weighted_adjacency_self_loops_split_hub_rows(be, iters):
    """Non-binary symmetric weights, a non-zero diagonal (masked out), a 150-neighbour hub next to the target (its row split over three 64-entry
    slots), 1 and 3 iterations (the returned mask is the one of the LAST forward - the initial one after a single iteration)."""
    rng = np.random.default_rng(41)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 700
    A, X, hub, idx = _hub_graph(rng, n, 2.2 / n, 150, weighted=True)
    t = int(idx[0])
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 3, t, rng.integers(0, 4, n), m0)
    em, xe, _, _ = _both(be, [sg], sd, iters)
    _same(em, xe)
    o = closed_form.ClosedFormOracle(A, X, sd, 3, sg.pred_label, t, m0)
    want = o.run(iters)
    got = xe.dense(0)
    assert np.abs(got * A - want).max() < 5e-6 and np.all(np.diag(got) == 0)


@pytest.mark.parametrize("kinds", [1, 7, 40])
def test_xl_feature_dictionary_and_l2_rows(be, kinds):
    """constant / categorical features (the LDS dictionary) and 40 distinct rows (feature rows from L2): XL == route 7"""
    rng = np.random.default_rng(100 + kinds)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 800
    A, _, hub, idx = _hub_graph(rng, n, 2.5 / n, 100)
    table = rng.standard_normal((kinds, 10)).astype(np.float32)
    X = table[rng.integers(0, kinds, n)]
    t = int(idx[0])
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 2, t, rng.integers(0, 4, n), m0)
    em, xe, _, _ = _both(be, [sg], sd, 4)
    _same(em, xe)


def test_xl_ragged_batch_more_than_512_row_slots_resume_and_trace(be):
    """Three targets of different sizes in one XL launch - one of them next to a 700-neighbour hub (more than 512 row slots: several rounds) - with
    the optimiser state handed back and the decision trace on: everything equals route 7 bit for bit; and 5 iterations == 2 + 3 resumed."""
    rng = np.random.default_rng(5)
    sd = helpers.load_ckpt("syn1")["sd"]             # the reference's widths (the trace needs them)
    sgs = []
    for n, dens, hubdeg in ((1400, 1.0, 700), (600, 2.4, 60), (1000, 2.0, 200)):
        A, _, hub, idx = _hub_graph(rng, n, dens / n, hubdeg)
        X = np.ones((n, 10), np.float32)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        sgs.append(Subgraph(A, X, int(rng.integers(0, 4)), int(idx[0]), rng.integers(0, 4, n), m0))
    em, xe, job, xj = _both(be, sgs, sd, 5, keep_state=True)
    _same(em, xe)
    s7, sx = job.fetch_state_edges(), xj.fetch_state_edges()
    for a, b in zip(s7, sx):
        assert np.array_equal(a, b)
    # the decision trace (the LOG instantiations - separate kernels: on the GPU the compiler contracts a product of the two differently, 1 ulp after five
    # iterations; on the emulator they are bit-identical too): every decision equal, values within 2 ulp
    emt, xet, jobt, xjt = _both(be, sgs, sd, 5, trace=True)
    assert np.abs(xet.masked_adj - emt.masked_adj).max() <= (0.0 if be.name == "emu" else 2.4e-7)
    assert np.abs(xet.masked_adj - xe.masked_adj).max() <= (0.0 if be.name == "emu" else 2.4e-7)
    g7, gx = jobt.fetch_trace()[0], xjt.fetch_trace()[0]
    for a, b in zip(g7, gx):
        assert np.array_equal(a, b) and a.any()
    # resume: 2 iterations, then 3 from the state handed back
    xj.reset_masks()
    xj.launch(Hyper(num_iters=2), keep_state=True)
    M2, m2, v2, f2 = xj.fetch_state_edges()
    st = xj.set_state_edges(2, M2, m2, v2, f2[:, 0], f2[:, 1], f2[:, 2])
    xj.launch(Hyper(num_iters=3), state=st, keep_state=True)
    x23 = xj.fetch_edges(with_mask=True)
    _same(xe, x23)
    for a, b in zip(sx, xj.fetch_state_edges()):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("case", ["isolated target", "edgeless"])
def test_xl_degenerate_graphs(be, case):
    """a target without neighbours / a sub-graph without any edge: nothing to optimise on the prediction path, the run must not fault"""
    rng = np.random.default_rng(9)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 700
    if case == "edgeless":
        A = np.zeros((n, n), np.float32)
        X = rng.standard_normal((n, 10)).astype(np.float32)
        t = 5
    else:
        A, X = helpers.random_graph(rng, n, 10, density=2.0 / n)
        t = 0
        A[t, :] = 0
        A[:, t] = 0
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)
    xj = engine.xl_job_from_subgraphs([sg], sd, device=be.device, lib=be.lib)
    eoff, rc = xj.edge_ids()
    rc = rc.cpu().numpy()
    xj.set_masks_on_edges(_edge_vals(m0, rc))
    xj.launch(Hyper(num_iters=3))
    xe = xj.fetch_edges(with_mask=True)
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, t, m0)
    want = o.run(3)
    assert not np.isnan(xe.masked_adj).any() and not np.isnan(xe.feat_mask).any()
    assert np.abs(xe.dense(0) - want).max() < 5e-6 and np.abs(xe.feat_mask[0] - o.f).max() < 5e-5


def test_xl_other_encoder_widths(be):
    """an encoder that is not the reference's (D = 7, H = 16, O = 12, C = 3): the 32-wide instantiation"""
    rng = np.random.default_rng(77)
    sd = helpers.random_model(rng, 7, 16, 12, 3)
    n = 650
    A, _, hub, idx = _hub_graph(rng, n, 2.4 / n, 90)
    X = rng.standard_normal((n, 7)).astype(np.float32)
    t = int(idx[0])
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 2, t, rng.integers(0, 3, n), m0)
    xj = engine.xl_job_from_subgraphs([sg], sd, device=be.device, lib=be.lib)
    eoff, rc = xj.edge_ids()
    rc = rc.cpu().numpy()
    xj.set_masks_on_edges(_edge_vals(m0, rc))
    xj.launch(Hyper(num_iters=3))
    xe = xj.fetch_edges(with_mask=True)
    o = closed_form.ClosedFormOracle(A, X, sd, 2, sg.pred_label, t, m0)
    want = o.run(3)
    assert np.abs(xe.dense(0) - want).max() < 5e-6 and np.abs(xe.feat_mask[0] - o.f).max() < 5e-5


def test_xl_sub_csr_from_a_full_graph_and_seeded_masks(be):
    """The route as the explainer uses it: ONE resident graph (syn1), k-hop lists from gnnx_khop, targets whose lists are strict subsets of the
    graph, initial masks from the seed protocol on the edges - against the live reference's 50-epoch golden outputs of those targets."""
    ck = helpers.load_ckpt("syn1")
    z = np.load(helpers.GOLDEN + "/syn1_full_explain.npz")
    import scipy.sparse as sp
    g = engine.device_graph(sp.csr_matrix(ck["adj"]), ck["feat"], ck["pred"], device=be.device)
    targets = np.asarray([300, 333, 401, 650], np.int64)
    dn = engine.khop_device(g, targets, 3, lib=be.lib)
    xj = engine.XLJob(g, dn, None, ck["label"][targets], ck["sd"], lib=be.lib)
    xj.set_masks_seeded(1000 + targets, threads=2)
    iters = int(z["early_epochs"])
    xj.launch(Hyper(num_iters=iters))
    xe = xj.fetch_edges()
    ids = [int(t) for t in z["targets"]]
    for k, t in enumerate(targets):
        j = ids.index(int(t))
        a, b = int(z["eoff"][j]), int(z["eoff"][j + 1])
        assert int(xe.eoff[k + 1] - xe.eoff[k]) == b - a
        want = z["vals_early"][a:b]
        got = xe.masked_adj[int(xe.eoff[k]):int(xe.eoff[k + 1])]
        assert np.abs(got - want).max() < 1e-5, (t, np.abs(got - want).max())


def test_xl_device_engine_walk_without_n2_scratch(be):
    """gnnx_xl_mt_edge_words: the raw engine words of every directed entry's Box-Muller pair, picked as the stream positions pass - no n^2 scratch -
    then ATen's own transform on the host: torch.equal with `manual_seed; normal_` on every edge entry, for streams below 16 values, of exactly 16,
    ragged ones (the redrawn tail), several blocks, entries of one engine block spread over two staged chunks."""
    if not engine.pair_staging_ok():
        pytest.skip("the host's normal_ lacks the pair-staging property")
    rng = np.random.default_rng(3)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    sgs, seeds = [], []
    for n, dens in ((3, 1.0), (4, 1.0), (5, 0.9), (37, 0.3), (64, 0.9), (129, 0.05), (700, 0.004), (25, 1.0)):
        A, X = helpers.random_graph(rng, n, 10, density=dens)
        sgs.append(Subgraph(A, X, 0, 0, rng.integers(0, 4, n), None))
        seeds.append(int(rng.integers(0, 2 ** 31)))
    xj = engine.xl_job_from_subgraphs(sgs, sd, device=be.device, lib=be.lib)
    seeds = np.asarray(seeds, np.int64)
    xj.set_masks_seeded_device(seeds, threads=2)
    got = xj.M_e[:xj.E].cpu().numpy()
    eoff, rc = xj.edge_ids()
    rc = rc.cpu().numpy()
    for k, sgr in enumerate(sgs):
        n = sgr.adj.shape[0]
        torch.manual_seed(int(seeds[k]))
        m0 = torch.empty(n, n).normal_(1.0, np.sqrt(2.0) * np.sqrt(2.0 / (n + n))).numpy()
        e = rc[eoff[k]:eoff[k + 1]]
        want = np.stack([m0[e[:, 0], e[:, 1]], m0[e[:, 1], e[:, 0]]], 1)
        assert np.array_equal(got[eoff[k]:eoff[k + 1]], want), (k, n)
    # and the host walk gives the same
    xj.set_masks_seeded(seeds, threads=2)
    assert np.array_equal(xj.M_e[:xj.E].cpu().numpy(), got)
