"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against
(a) the committed golden outputs of the REAL reference and (b) the oracle on seeded inputs.
Tolerance: 1e-5 absolute on masked_adj and sigma(feat_mask) (fp32, exact-f32 MFMA)."""
import os

import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
from oracle import closed_form

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _node_subgraph(ck, gx, t):
    nb = gx[f"{t}:neighbors"]
    A, X, lab, yhat = helpers.subgraph(ck, nb)
    new = int(gx[f"{t}:node_idx_new"])
    return Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy())


def test_library_is_the_hip_build():
    lib = engine.get_library()
    assert b"gfx950" in lib.gnnx_version()
    assert os.path.basename(engine.library_path()) == "libgnnx_hip.so"


@pytest.mark.parametrize("name", ["syn1", "syn4", "syn5"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_golden_reference_outputs_node_mode(name, use_graph):
    """All golden targets of a dataset as ONE batched job, 300 iterations, vs the reference's own outputs."""
    ck, gx = helpers.load_ckpt(name), helpers.load_explain(name)
    targets = [int(t) for t in gx["targets"]]
    subs = [_node_subgraph(ck, gx, t) for t in targets]
    job = MaskOptimJob(subs, ck["sd"])
    hy = Hyper(num_iters=int(gx["epochs"]), record_loss=True, use_graph=use_graph)
    res = job.run([s.mask0 for s in subs], hy)
    for i, t in enumerate(targets):
        rc = gx[f"{t}:edge_rc"]
        got = res.masked_adj[i][rc[:, 0], rc[:, 1]]
        err = np.abs(got - gx[f"{t}:masked_adj_edges"]).max()
        if t in helpers.ILL_CONDITIONED.get(name, ()):
            # plateau-crossing targets: round-off is amplified even between two CPU implementations (tests/test_oracle_golden.py), so the 300-epoch outcome
            # bounds nothing - reported here, gated window by window against the reference's own state by tests/test_decision_parity.py (no tolerance of their own)
            ferr = np.abs(_sig(res.feat_mask[i]) - gx[f"{t}:feat_mask_sigmoid"]).max()
            print(f"{name}/{t} (ill-conditioned over the horizon, reported): masked_adj err {err:.2e}, feat {ferr:.2e}")
            assert np.isfinite(ferr) and err <= helpers.ILL_TOL_MASK, f"{name}/{t}: masked_adj err {err}"
            continue
        assert err <= TOL, f"{name}/{t} n={len(subs[i].adj)}: masked_adj err {err}"
        assert np.all(res.masked_adj[i][subs[i].adj == 0] == 0)
        assert np.array_equal(res.masked_adj[i], res.masked_adj[i].T)
        ferr = np.abs(_sig(res.feat_mask[i]) - gx[f"{t}:feat_mask_sigmoid"]).max()
        assert ferr <= TOL, f"{name}/{t}: feat mask err {ferr}"
        merr = np.abs(res.mask[i][rc[:, 0], rc[:, 1]] - gx[f"{t}:final_mask_edges"]).max()
        assert merr <= 1e-3, f"{name}/{t}: final mask parameter err {merr}"
        loss = res.loss[i][:, :5].sum(1)
        assert np.allclose(loss, gx[f"{t}:loss"], rtol=1e-4, atol=1e-4), f"{name}/{t}: loss trace"


def test_golden_reference_outputs_graph_mode():
    z = np.load(os.path.join(helpers.GOLDEN, "graphmode_explain.npz"))
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    G = z["adj"].shape[0]
    subs = [Subgraph(z["adj"][g], z["feat"][g], int(z["label"][g]), 0, None,
                     helpers.seeded_mask0(g, z["adj"].shape[1]).numpy()) for g in range(G)]
    job = MaskOptimJob(subs, sd, graph_mode=True)
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=int(z["epochs"]), record_loss=True))
    for g in range(G):
        assert np.abs(res.masked_adj[g] - z[f"{g}:masked_adj"]).max() <= TOL
        assert np.abs(_sig(res.feat_mask[g]) - z[f"{g}:feat_mask_sigmoid"]).max() <= TOL
        assert np.allclose(res.loss[g][:, :5].sum(1), z[f"{g}:loss"], rtol=1e-4, atol=1e-4)


def test_single_iteration_stages_vs_oracle():
    """One forward (probabilities + masked adjacency) and one full step on a mid-size target."""
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    s = _node_subgraph(ck, gx, 555)
    job = MaskOptimJob([s], ck["sd"])
    probs, ma = job.forward([s.mask0])
    o = closed_form.ClosedFormOracle(s.adj, s.feat, ck["sd"], s.gt_label, s.pred_label, s.target_row, s.mask0)
    o.iterate()
    assert np.abs(ma[0] - o.stages["Abar"]).max() < 1e-6
    assert np.abs(probs[0] - o.stages["p"]).max() < 1e-5
    edges = s.adj != 0
    res = job.run([s.mask0], Hyper(num_iters=1))                 # n = 104: sparse on-chip-resident kernel
    assert np.abs(res.mask[0] - o.M)[edges].max() < 1e-5
    assert np.array_equal(res.mask[0][~edges], s.mask0[~edges])  # non-edge entries are dead state there
    assert np.abs(res.feat_mask[0] - o.f).max() < 1e-5
    res = MaskOptimJob([s], ck["sd"], analyze=False).run([s.mask0], Hyper(num_iters=1))   # dense streaming kernels
    assert np.abs(res.mask[0] - o.M).max() < 1e-5
    assert np.abs(res.feat_mask[0] - o.f).max() < 1e-5


def test_ragged_batch_equals_individual_jobs_and_is_deterministic():
    """Batching must not change any bit: every target takes the kernel its own size and edge count select (streaming,
    dense single-tile resident, sparse resident), whatever else is in the batch."""
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    subs = [_node_subgraph(ck, gx, t) for t in (302, 555, 309, 302)]
    for use_resident in (False, True):
        hy = Hyper(num_iters=30, use_resident=use_resident)
        res = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], hy)
        again = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], hy)
        for i, s in enumerate(subs):
            solo = MaskOptimJob([s], ck["sd"]).run([s.mask0], hy)
            assert np.array_equal(solo.masked_adj[0], res.masked_adj[i])
            assert np.array_equal(again.masked_adj[i], res.masked_adj[i])
        assert np.array_equal(res.masked_adj[0], res.masked_adj[3])


def test_edge_cases_single_node_and_unsupported():
    ck, _ = helpers.load_ckpt("syn1"), None
    one = Subgraph(np.zeros((1, 1), np.float32), np.ones((1, 10), np.float32), 0, 0, np.zeros(1), np.ones((1, 1), np.float32))
    res = MaskOptimJob([one], ck["sd"]).run([one.mask0], Hyper(num_iters=5))
    assert res.masked_adj[0].shape == (1, 1) and res.masked_adj[0][0, 0] == 0 and np.isfinite(res.mask[0]).all()
    asym = Subgraph(np.triu(np.ones((4, 4), np.float32), 1), np.ones((4, 10), np.float32), 0, 0, np.zeros(4), None)
    with pytest.raises(NotImplementedError):
        MaskOptimJob([asym], ck["sd"])


def test_full_size_properties_syn1_all_motif_nodes():
    """BASELINE config 2 at full size (400 targets): size-independent properties of the output."""
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    subs = []
    for t in range(300, 700):
        new, A, nb = idx.extract(t)
        subs.append(Subgraph(A, ck["feat"][nb], int(ck["label"][t]), new, np.argmax(ck["pred"][nb], 1),
                             helpers.seeded_mask0(t, len(nb)).numpy()))
    res = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], Hyper(num_iters=300, use_graph=True))
    for s, ma, fm in zip(subs, res.masked_adj, res.feat_mask):
        assert np.isfinite(ma).all() and np.isfinite(fm).all()
        assert ma.min() >= 0 and ma.max() <= 1
        assert np.array_equal(ma, ma.T) and np.all(ma[s.adj == 0] == 0) and np.all(np.diag(ma) == 0)


@pytest.mark.parametrize("name", ["syn1", "syn4", "syn5"])
def test_hybrid_resident_plus_streaming_vs_reference(name):
    """Without loss logging, small targets (n <= 96: 1, 2 or 3 row blocks) run in the on-chip-resident kernels on side
    streams while the other targets stream: same golden outputs, and bitwise-equal to an all-streaming run would be too strict
    (different summation order), so both are held to the reference tolerance."""
    ck, gx = helpers.load_ckpt(name), helpers.load_explain(name)
    targets = [int(t) for t in gx["targets"]]
    subs = [_node_subgraph(ck, gx, t) for t in targets]
    assert any(len(s.adj) <= 96 for s in subs)
    for use_graph in (False, True):
        res = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], Hyper(num_iters=int(gx["epochs"]), use_graph=use_graph))
        for i, t in enumerate(targets):
            rc = gx[f"{t}:edge_rc"]
            err = np.abs(res.masked_adj[i][rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max()
            ferr = np.abs(_sig(res.feat_mask[i]) - gx[f"{t}:feat_mask_sigmoid"]).max()
            ill = t in helpers.ILL_CONDITIONED.get(name, ())
            if ill:      # (reported, not gated at the horizon: see test_golden_reference_outputs_node_mode)
                print(f"{name}/{t} (ill-conditioned over the horizon, reported): masked_adj err {err:.2e}, feat {ferr:.2e}")
                assert err <= helpers.ILL_TOL_MASK
                continue
            assert err <= TOL, f"{name}/{t}: {err}"
            assert ferr <= TOL, f"{name}/{t}: feat {ferr}"
            assert np.array_equal(res.masked_adj[i], res.masked_adj[i].T)


def test_multi_block_resident_kernel_vs_golden_and_streaming():
    """Without gnnx_plan_analyze, a batch whose targets all have n <= 96 runs entirely in the DENSE resident kernels:
    syn1 target 309 (n = 48, two row blocks) + 302 (one block), 300 iterations against the reference's golden masks
    and against the streaming path."""
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    subs = [_node_subgraph(ck, gx, t) for t in (309, 302)]
    m0 = [s.mask0 for s in subs]
    res = MaskOptimJob(subs, ck["sd"], analyze=False).run(m0, Hyper(num_iters=300, use_graph=True, use_resident=True))
    stream = MaskOptimJob(subs, ck["sd"]).run(m0, Hyper(num_iters=300, use_graph=True, use_resident=False))
    for i, t in enumerate((309, 302)):
        rc = gx[f"{t}:edge_rc"]
        assert np.abs(res.masked_adj[i][rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max() <= TOL
        assert np.abs(_sig(res.feat_mask[i]) - gx[f"{t}:feat_mask_sigmoid"]).max() <= TOL
        assert np.abs(res.masked_adj[i] - stream.masked_adj[i]).max() <= TOL
        assert np.array_equal(res.masked_adj[i], res.masked_adj[i].T)


def test_resident_only_batch_is_deterministic():
    ck, gx = helpers.load_ckpt("syn4"), helpers.load_explain("syn4")
    subs = [_node_subgraph(ck, gx, int(t)) for t in gx["targets"]] * 8
    hy = Hyper(num_iters=100, use_graph=True)
    a = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], hy)
    b = MaskOptimJob(subs, ck["sd"]).run([s.mask0 for s in subs], hy)
    for x, y in zip(a.masked_adj, b.masked_adj):
        assert np.array_equal(x, y)
    assert np.array_equal(a.masked_adj[0], a.masked_adj[4])


@pytest.mark.parametrize("D,H,O,C,n,graph_mode,path", [
    (7, 13, 9, 3, 21, False, "resident"), (7, 13, 9, 3, 21, False, "stream"), (5, 32, 32, 6, 45, False, "stream"),
    (5, 32, 32, 6, 45, False, "resident"), (5, 32, 32, 6, 45, False, "sparse"), (31, 8, 3, 2, 70, False, "stream"),
    (31, 8, 3, 2, 70, False, "resident"), (31, 8, 3, 2, 70, False, "sparse"), (10, 20, 20, 4, 96, False, "resident"),
    (7, 13, 9, 3, 130, False, "sparse"), (14, 20, 20, 2, 40, True, "stream"), (3, 9, 17, 9, 33, True, "stream"),
    (14, 20, 20, 2, 40, True, "sparse"), (3, 9, 17, 5, 70, True, "sparse"),
    (10, 20, 20, 4, 700, False, "sparse"), (7, 13, 9, 3, 1500, False, "sparse"),
    (10, 20, 20, 4, 300, False, "stream"), (10, 20, 20, 4, 300, False, "sparse"), (10, 20, 20, 4, 400, False, "sparse"),
])
def test_generic_shapes_match_closed_form(D, H, O, C, n, graph_mode, path):
    """Encoder shapes other than the fixtures': odd widths, full 32-wide layers, wide input, many classes; through the
    streaming kernels, the dense resident kernels (no analysis) and the sparse resident kernel."""
    rng = np.random.default_rng(D * 1000 + H * 10 + n)
    sd = helpers.random_model(rng, D, H, O, C)
    A, X = helpers.random_graph(rng, n, D, density=0.15 if n < 100 else 0.03 if n < 400 else 0.01 if n < 600 else 2.0 / n)
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    t, gt = int(rng.integers(0, n)), int(rng.integers(0, C))
    yhat = None if graph_mode else rng.integers(0, C, n)
    sg = Subgraph(A, X, gt, 0 if graph_mode else t, yhat, m0)
    iters = 6
    job = MaskOptimJob([sg], sd, graph_mode=graph_mode, analyze=(path == "sparse"))
    res = job.run([m0], Hyper(num_iters=iters, use_resident=(path != "stream")))
    o = closed_form.ClosedFormOracle(A, X, sd, gt, yhat, 0 if graph_mode else t, m0, graph_mode=graph_mode)
    want = o.run(iters)
    live = (A != 0) if path == "sparse" else np.ones_like(A, bool)
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5
    assert np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    if path == "sparse":
        assert np.array_equal(res.mask[0][~live], m0[~live])      # proves the sparse kernel ran (dead state untouched)
