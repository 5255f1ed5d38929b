"""The product's kernel + host source against the oracle, stage by stage and over short optimisation runs - every case
on TWO backends through one body:
  * "emu": the SAME sources compiled against tests/emu (a HIP emulator), because the build container has no GPU
    (`-m "not gpu"`);
  * "gpu": libgnnx_hip.so on a real MI355X through the C ABI (`-m gpu`) - the emulator runs waves as sequential
    fibers and cannot see a missing wave_sync, a DPP shift that differs on hardware, a spill miscompile or an LDS race,
    so every edge case (hub rows split over slots, > 512 row slots, edgeless / isolated targets, weighted adjacency with
    self-loops, the mixed launch, ...) has its hardware twin here."""
import numpy as np
import torch
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, Subgraph
from oracle import closed_form


class _Backend:
    def __init__(self, name):
        self.name = name
        if name == "emu":
            from emu.emu_engine import emu_library
            self.lib, self.device = emu_library(), "cpu"
        else:
            import torch
            assert torch.cuda.is_available(), "the gpu twins need an MI355X"
            self.lib, self.device = engine.get_library(), "cuda:0"

    def job(self, subgraphs, state_dict, graph_mode=False, analyze=True):
        return engine.MaskOptimJob(subgraphs, state_dict, graph_mode=graph_mode, device=self.device, lib=self.lib, analyze=analyze)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def _node_case(name, t):
    ck, gx = helpers.load_ckpt(name), helpers.load_explain(name)
    nb = gx[f"{t}:neighbors"]
    A, X, lab, yhat = helpers.subgraph(ck, nb)
    new = int(gx[f"{t}:node_idx_new"])
    m0 = helpers.seeded_mask0(t, len(nb)).numpy()
    return ck, gx, Subgraph(A, X, int(lab[new]), new, yhat, m0)


@pytest.mark.parametrize("name,t", [("syn1", 302), ("syn4", 511), ("syn1", 309)])
def test_forward_probs_and_masked_adj(be, name, t):
    ck, gx, sg = _node_case(name, t)
    job = be.job([sg], ck["sd"])
    probs, ma = job.forward([sg.mask0])
    o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
    o.iterate()
    assert np.abs(ma[0] - o.stages["Abar"]).max() < 1e-6
    assert np.abs(probs[0] - o.stages["p"]).max() < 1e-5


@pytest.mark.parametrize("use_resident", [True, False])
@pytest.mark.parametrize("name,t,iters", [("syn1", 302, 12), ("syn4", 511, 12), ("syn1", 309, 6)])
def test_short_run_matches_closed_form(be, name, t, iters, use_resident):
    """Masks, ALL n x n mask parameters and the logged loss terms (explain.py:808-819) after a short run - on the edge-sparse resident
    kernel in its logging form (+ k_dead_entries for the entries off the edges) and on the dense streaming kernels."""
    ck, gx, sg = _node_case(name, t)
    job = be.job([sg], ck["sd"])
    assert set(job.route()) <= {4, 5, 6, 8}
    hy = Hyper(num_iters=iters, record_loss=True, use_resident=use_resident)
    res = job.run([sg.mask0], hy)
    o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
    want = o.run(iters)
    assert np.abs(res.masked_adj[0] - want).max() < 2e-6
    assert np.abs(res.mask[0] - o.M).max() < 2e-5
    assert np.abs(res.feat_mask[0] - o.f).max() < 2e-5
    tr = np.asarray(o.trace)          # loss, pred, size, lap, ent, feat_size
    got = res.loss[0]
    assert np.allclose(got[:, 0], tr[:, 1], atol=1e-5)
    assert np.allclose(got[:, 1], tr[:, 2], rtol=1e-5)
    assert np.allclose(got[:, 2], tr[:, 3], rtol=1e-4, atol=1e-7)
    assert np.allclose(got[:, 3], tr[:, 4], rtol=1e-5)
    assert np.allclose(got[:, 4], tr[:, 5], rtol=1e-6)


def test_batch_of_ragged_targets_matches_individual_runs(be):
    ck, gx, a = _node_case("syn1", 302)
    _, _, b = _node_case("syn1", 309)
    ck4, _, c = _node_case("syn1", 302)
    hy = Hyper(num_iters=4)
    job = be.job([a, b, c], ck["sd"])
    res = job.run([a.mask0, b.mask0, c.mask0], hy)
    for i, s in enumerate((a, b, c)):
        o = closed_form.ClosedFormOracle(s.adj, s.feat, ck["sd"], s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(4)).max() < 2e-6
    assert np.array_equal(res.masked_adj[0], res.masked_adj[2])


def test_graph_mode_short_run_matches_closed_form(be):
    z = np.load(helpers.GOLDEN + "/graphmode_explain.npz")
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    g = 4
    A, X, lab = z["adj"][g], z["feat"][g], int(z["label"][g])
    m0 = helpers.seeded_mask0(g, A.shape[0]).numpy()
    job = be.job([Subgraph(A, X, lab, 0, None, m0)], sd, graph_mode=True)
    res = job.run([m0], Hyper(num_iters=5, record_loss=True))
    o = closed_form.ClosedFormOracle(A, X, sd, lab, None, 0, m0, graph_mode=True)
    want = o.run(5)
    assert np.abs(res.masked_adj[0] - want).max() < 2e-6
    assert np.abs(res.feat_mask[0] - o.f).max() < 2e-5
    assert np.allclose(res.loss[0][:, :5].sum(1), np.asarray(o.trace)[:, 0], rtol=1e-5)


def test_outputs_are_bitwise_symmetric_and_zero_off_edges(be):
    ck, gx, sg = _node_case("syn1", 309)       # n = 48 -> 2x2 tiles: diagonal and off-diagonal tile pairs
    res = be.job([sg], ck["sd"]).run([sg.mask0], Hyper(num_iters=5))
    ma = res.masked_adj[0]
    assert np.array_equal(ma, ma.T) and np.all(ma[sg.adj == 0] == 0) and np.all(np.diag(ma) == 0)


@pytest.mark.parametrize("sparse", [False, True])
def test_large_target_many_row_blocks(be, sparse):
    """n = 310 -> ld = 320: 10 row blocks.  Dense streaming kernels (55 tile pairs, per-block partials summed over 10
    slots) and the sparse on-chip-resident kernel (1432 undirected edges, 3 row blocks per wave)."""
    ck, gx, sg = _node_case("syn1", 300)
    assert sg.adj.shape[0] == 310
    res = be.job([sg], ck["sd"], analyze=sparse).run([sg.mask0], Hyper(num_iters=2))
    o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
    want = o.run(2)
    edges = sg.adj != 0
    assert np.abs(res.masked_adj[0] - want).max() < 2e-6
    assert np.abs(res.feat_mask[0] - o.f).max() < 2e-5
    if sparse:   # only the entries on edges are live state; the others keep their initial values
        assert np.abs(res.mask[0] - o.M)[edges].max() < 2e-5
        assert np.array_equal(res.mask[0][~edges], sg.mask0[~edges])
    else:
        assert np.abs(res.mask[0] - o.M).max() < 2e-5


@pytest.mark.parametrize("graph_mode,wide,ku", [(False, "1", "64"), (True, "1", "96"), (False, "4", "0"), (True, "4", "64"), (False, "2", "128")])
def test_contraction_units_slices_and_row_block_groups(be, graph_mode, wide, ku, monkeypatch):
    """k_conv cuts the K range of long rows into slices (GNNX_CONV_KU; partial tiles in slabs, k_conv_reduce joins them in slice
    order in the next launch) and can run groups of 4 / 2 / 1 adjacent row blocks per workgroup (GNNX_CONV_WIDE).  n = 310
    (ld = 320, 10 row blocks: groups 4 + 4 + 2, 5 / 3 / 2 slices) and n = 275 (ld = 288, 9 row blocks: 4 + 4 + 1).  Every
    combination has to agree with whole rows / one row block per workgroup to summation-order noise and with the closed form."""
    ck, gx, sg = _node_case("syn1", 300)
    keep = np.unique(np.r_[np.arange(274), sg.target_row])[:275]
    if sg.target_row not in keep:
        keep[-1] = sg.target_row
        keep = np.sort(keep)
    row = int(np.searchsorted(keep, sg.target_row))
    cases = [sg, Subgraph(sg.adj[np.ix_(keep, keep)], sg.feat[keep], sg.gt_label, row, sg.pred_label[keep], sg.mask0[np.ix_(keep, keep)])]
    if graph_mode:
        cases = [Subgraph(c.adj, c.feat, c.gt_label, 0, None, c.mask0) for c in cases]
    hy = Hyper(num_iters=3, record_loss=True)
    runs = {}
    for key, (w, k) in {"plain": ("1", "0"), "cut": (wide, ku)}.items():
        monkeypatch.setenv("GNNX_CONV_WIDE", w)
        monkeypatch.setenv("GNNX_CONV_KU", k)
        job = be.job(cases, ck["sd"], graph_mode=graph_mode, analyze=False)
        runs[key] = job.run([c.mask0 for c in cases], hy)
    for k, c in enumerate(cases):
        o = closed_form.ClosedFormOracle(c.adj, c.feat, ck["sd"], c.gt_label, c.pred_label, c.target_row, c.mask0, graph_mode=graph_mode)
        want = o.run(3)
        assert np.abs(runs["cut"].masked_adj[k] - want).max() < 2e-6
        assert np.abs(runs["cut"].feat_mask[k] - o.f).max() < 2e-5
        assert np.abs(runs["cut"].masked_adj[k] - runs["plain"].masked_adj[k]).max() < 1e-6
        assert np.allclose(runs["cut"].loss[k], runs["plain"].loss[k], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("analyze", [False, True])
def test_resident_kernel_matches_streaming_and_reference(be, analyze):
    """Single-tile targets (n <= 32) run on chip: in the dense resident kernel k_resident<1> (plan not analysed) or in the
    64-thread class of the sparse resident kernel; an all-resident batch launches no streaming kernel at all.  Both must
    agree with the streaming path and with the closed form."""
    ck, gx = helpers.load_ckpt("syn4"), helpers.load_explain("syn4")
    subs = [_node_case("syn4", t)[2] for t in (511, 870)]
    iters = 40
    job = be.job(subs, ck["sd"], analyze=analyze)
    assert list(job.route()) == ([6, 6] if analyze else [1, 1])
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=iters, use_resident=True))
    stream = be.job(subs, ck["sd"]).run([s.mask0 for s in subs], Hyper(num_iters=iters, use_resident=False))
    for a, b, fa, fb in zip(res.masked_adj, stream.masked_adj, res.feat_mask, stream.feat_mask):
        assert np.abs(a - b).max() < 1e-6 and np.abs(fa - fb).max() < 1e-5
        assert np.array_equal(a, a.T)
    s = subs[0]
    o = closed_form.ClosedFormOracle(s.adj, s.feat, ck["sd"], s.gt_label, s.pred_label, s.target_row, s.mask0)
    assert np.abs(res.masked_adj[0] - o.run(iters)).max() < 2e-6
    live = (s.adj != 0) if analyze else np.ones_like(s.adj, bool)
    assert np.abs(res.mask[0] - o.M)[live].max() < 2e-5 and np.abs(res.feat_mask[0] - o.f).max() < 2e-5


@pytest.mark.parametrize("analyze", [False, True])
def test_resident_kernel_full_run_vs_golden(be, analyze):
    ck, gx, sg = _node_case("syn1", 302)
    res = be.job([sg], ck["sd"], analyze=analyze).run([sg.mask0], Hyper(num_iters=300))
    rc = gx["302:edge_rc"]
    assert np.abs(res.masked_adj[0][rc[:, 0], rc[:, 1]] - gx["302:masked_adj_edges"]).max() <= 1e-5
    assert np.abs(1 / (1 + np.exp(-res.feat_mask[0])) - gx["302:feat_mask_sigmoid"]).max() <= 1e-5


def test_two_block_resident_kernel_vs_streaming_and_golden(be):
    """syn1 target 309 (n = 48 -> 2 row blocks: diagonal and off-diagonal tile pairs, mirror entries in registers)
    through the resident kernel: 300 iterations against the reference's golden mask, and against the streaming path."""
    ck, gx, sg = _node_case("syn1", 309)
    res = be.job([sg], ck["sd"], analyze=False).run([sg.mask0], Hyper(num_iters=300, use_resident=True))
    rc = gx["309:edge_rc"]
    assert np.abs(res.masked_adj[0][rc[:, 0], rc[:, 1]] - gx["309:masked_adj_edges"]).max() <= 1e-5
    assert np.abs(1 / (1 + np.exp(-res.feat_mask[0])) - gx["309:feat_mask_sigmoid"]).max() <= 1e-5
    short = be.job([sg], ck["sd"], analyze=False).run([sg.mask0], Hyper(num_iters=20, use_resident=True))
    stream = be.job([sg], ck["sd"]).run([sg.mask0], Hyper(num_iters=20, use_resident=False))
    assert np.abs(short.masked_adj[0] - stream.masked_adj[0]).max() < 1e-6
    assert np.abs(short.mask[0] - stream.mask[0]).max() < 1e-5
    assert np.array_equal(short.masked_adj[0], short.masked_adj[0].T)


def test_device_side_packing_equals_host_packing(be):
    """gnnx_pack_csr (sub-graphs sliced from CSR on the device) must build exactly the buffers the host packer builds."""
    import torch
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    targets = [302, 309, 555, 300]
    nbs = idx.neighbors_batch(targets)
    rows = [int(np.searchsorted(nb, v)) for v, nb in zip(targets, nbs)]
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"], device=be.device)
    assert graph.binary
    dev = engine.MaskOptimJob.from_csr(graph, nbs, rows, ck["label"][targets], ck["sd"], lib=be.lib)
    subs = [Subgraph(ck["adj"][np.ix_(nb, nb)], ck["feat"][nb], int(ck["label"][t]), r, np.argmax(ck["pred"][nb], 1), None)
            for t, nb, r in zip(targets, nbs, rows)]
    host = be.job(subs, ck["sd"])
    assert torch.equal(dev.A.cpu(), host.A.cpu()) and torch.equal(dev.X.cpu(), host.X.cpu()) and torch.equal(dev.yhat.cpu(), host.yhat.cpu())
    for a, s in zip(dev.adjacency(), subs):
        assert np.array_equal(a, s.adj)
    # the pack kernel zeroes its own row blocks (no memset ahead of it): poisoned buffers come out identical
    nb_flat, nb_off_d, g = dev._keepalive
    dev.A.fill_(float("nan")); dev.X.fill_(7.0); dev.yhat.fill_(-3.0)
    dev._enter()
    engine._check(dev.lib, dev.lib.gnnx_pack_csr(dev.handle, g.indptr.data_ptr(), g.indices.data_ptr(), None, g.feat.data_ptr(), g.feat.shape[1],
                                                 g.pred_label.data_ptr(), nb_flat.data_ptr(), nb_off_d.data_ptr(), dev.A.data_ptr(), dev.X.data_ptr(),
                                                 dev.yhat.data_ptr(), dev._stream()))
    dev._leave()
    assert torch.equal(dev.A.cpu(), host.A.cpu()) and torch.equal(dev.X.cpu(), host.X.cpu()) and torch.equal(dev.yhat.cpu(), host.yhat.cpu())


def test_pack_with_analysis_equals_the_separate_calls(be):
    """gnnx_pack_csr_analyze (row degrees / upper-triangle counts as by-products of the packing kernel, edge counts from the analysis copy,
    gnnx_edge_layout from the row starts left on the device) against gnnx_pack_csr + gnnx_plan_analyze_features + gnnx_edge_counts +
    gnnx_edge_positions: same packed arrays, same route, same edge layout - also on a weighted graph with a self-loop."""
    import torch
    import scipy.sparse as sp
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    targets = np.asarray([302, 309, 555, 300, 420, 699])
    for weighted in (False, True):
        csr = sp.csr_matrix(idx.csr, dtype=np.float32)
        if weighted:
            rng = np.random.default_rng(3)
            w = sp.triu(csr, 1).tocoo()
            vals = rng.uniform(0.5, 1.5, w.nnz).astype(np.float32)
            up = sp.coo_matrix((vals, (w.row, w.col)), shape=csr.shape)
            csr = (up + up.T + sp.diags(np.where(np.arange(csr.shape[0]) % 7 == 0, 2.0, 0.0).astype(np.float32))).tocsr()
        graph = engine.device_graph(csr, ck["feat"], ck["pred"], device=be.device)
        assert graph.binary == (not weighted)
        dn = engine.khop_device(graph, targets, 3, lib=be.lib)
        fused = engine.MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"], lib=be.lib)
        plain = engine.MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"], lib=be.lib, analyze=False)
        engine._check(plain.lib, plain.lib.gnnx_plan_analyze_features(plain.handle, plain.A.data_ptr(), plain.X.data_ptr(), plain._stream()))
        assert getattr(plain, "_edge_counts", None) is None and fused._edge_counts is not None
        assert torch.equal(fused.A.cpu(), plain.A.cpu()) and torch.equal(fused.X.cpu(), plain.X.cpu())
        assert np.array_equal(fused.route(), plain.route())
        fused._edge_layout()
        plain._edge_layout()          # gnnx_edge_counts + gnnx_edge_positions
        assert np.array_equal(fused._eoff, plain._eoff) and int(fused._eoff[-1]) > 0
        E = int(fused._eoff[-1])
        assert torch.equal(fused._rc[:E].cpu(), plain._rc[:E].cpu()) and torch.equal(fused._epos[:E].cpu(), plain._epos[:E].cpu())


@pytest.mark.parametrize("D,H,O,C,n,graph_mode,path", [
    (7, 13, 9, 3, 21, False, "resident"),     # odd widths, single-tile resident kernel
    (7, 13, 9, 3, 21, False, "stream"),       # same through the streaming kernels
    (5, 32, 32, 6, 45, False, "stream"),      # full-width hidden layers, two row blocks
    (5, 32, 32, 6, 45, False, "resident"),    # ... in the two-block dense resident kernel
    (5, 32, 32, 6, 45, False, "sparse"),      # ... in the sparse resident kernel
    (31, 8, 3, 2, 70, False, "stream"),       # wide input (beyond the 16 columns of the common case), three row blocks
    (31, 8, 3, 2, 70, False, "resident"),     # ... in the three-block dense resident kernel
    (31, 8, 3, 2, 70, False, "sparse"),
    (10, 20, 20, 4, 96, False, "resident"),   # largest dense-resident target (no padding rows)
    (7, 13, 9, 3, 130, False, "sparse"),      # odd widths, 5 row blocks (two per wave for wave 0)
    (14, 20, 20, 2, 40, True, "stream"),      # graph mode
    (3, 9, 17, 9, 33, True, "stream"),        # graph mode, odd widths, more classes than the resident paths take
    (14, 20, 20, 2, 40, True, "sparse"),      # graph mode in the sparse resident kernel (three full layers, max-pool head)
    (3, 9, 17, 5, 70, True, "sparse"),        # ... odd widths, O > H
])
def test_generic_shapes_match_closed_form(be, D, H, O, C, n, graph_mode, path):
    rng = np.random.default_rng(D * 1000 + H * 10 + n)
    sd = helpers.random_model(rng, D, H, O, C)
    A, X = helpers.random_graph(rng, n, D)
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    t, gt = int(rng.integers(0, n)), int(rng.integers(0, C))
    yhat = None if graph_mode else rng.integers(0, C, n)
    sg = Subgraph(A, X, gt, 0 if graph_mode else t, yhat, m0)
    iters = 4
    job = be.job([sg], sd, graph_mode=graph_mode, analyze=(path == "sparse"))
    res = job.run([m0], Hyper(num_iters=iters, use_resident=(path != "stream")))
    o = closed_form.ClosedFormOracle(A, X, sd, gt, yhat, 0 if graph_mode else t, m0, graph_mode=graph_mode)
    want = o.run(iters)
    live = (A != 0) if path == "sparse" else np.ones_like(A, bool)
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5
    assert np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    if path == "sparse":
        assert np.array_equal(res.mask[0][~live], m0[~live])


def test_sparse_resident_kernel_full_run_vs_golden(be):
    """syn1 targets 555 (n = 104) and 309 (n = 48) as one batch with 302 (n = 6, dense single-tile kernel): 300
    iterations in the sparse on-chip-resident kernel against the reference's golden masks."""
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    targets = (555, 309, 302)
    subs = [_node_case("syn1", t)[2] for t in targets]
    res = be.job(subs, ck["sd"]).run([s.mask0 for s in subs], Hyper(num_iters=300))
    for i, t in enumerate(targets):
        rc = gx[f"{t}:edge_rc"]
        assert np.abs(res.masked_adj[i][rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max() <= 1e-5
        assert np.abs(1 / (1 + np.exp(-res.feat_mask[i])) - gx[f"{t}:feat_mask_sigmoid"]).max() <= 1e-5
        assert np.array_equal(res.masked_adj[i], res.masked_adj[i].T)
        assert np.all(res.masked_adj[i][subs[i].adj == 0] == 0)


def test_plan_routing_by_size_and_edge_count(be):
    """gnnx_plan_analyze routes every target by what it finds in the packed adjacency (gnnx_get_route): targets whose edge
    state fits a CU -> sparse resident kernel in the smallest size class that holds them (6: 64 threads, n <= 32;
    5: 256 threads, n <= 128; 4: 1024 threads, n <= 512; a batch that needs class 4 uses only it and the dense single-tile
    kernel), dense graphs with more than 2048 undirected edges -> streaming (0).  Without the analysis nothing is routed to the sparse kernel."""
    rng = np.random.default_rng(5)
    sd = helpers.random_model(rng, 10, 20, 20, 4)

    def sub(n, density):
        A, X = helpers.random_graph(rng, n, 10, density=density)
        return Subgraph(A, X, 1, 0, rng.integers(0, 4, n), np.ones((n, n), np.float32))

    subs = [sub(20, 0.2), sub(60, 0.1), sub(200, 0.02), sub(120, 0.5)]
    assert (subs[3].adj != 0).sum() // 2 > 2048
    route = list(be.job(subs, sd).route())
    # a 512-thread target in the batch: the mid-size target keeps its 256-thread class and the single-tile one its 64-thread class, and
    # all three share ONE launch (k_sparse_resident_mixed: the 256-thread targets two to a workgroup - "pair" workgroups, round 5)
    assert route == [6, 5, 8, 0]
    # a small batch of 256-thread and single-tile targets becomes one mixed launch too; a 256-thread target alone has its own launch
    assert list(be.job(subs[:2], sd).route()) == [6, 5]
    assert list(be.job(subs[1:2], sd).route()) == [5]
    assert list(be.job(subs, sd, analyze=False).route()) == [1, 0, 0, 0]
    small = [sub(20, 0.2), sub(60, 0.1)]
    assert list(be.job(small, sd, analyze=False).route()) == [1, 2]      # all-small batch: dense resident kernels
    res = be.job(subs, sd).run([s.mask0 for s in subs], Hyper(num_iters=2))
    for s_, ma in zip(subs, res.masked_adj):
        o = closed_form.ClosedFormOracle(s_.adj, s_.feat, sd, s_.gt_label, s_.pred_label, s_.target_row, s_.mask0)
        assert np.abs(ma - o.run(2)).max() < 5e-6


@pytest.mark.parametrize("n", [24, 70, 200])
def test_sparse_kernel_weighted_adjacency_and_self_loops(be, n):
    """Non-binary symmetric edge weights and a non-zero diagonal (masked out by the reference, explain.py:618, 678)
    through the three size classes of the sparse resident kernel."""
    rng = np.random.default_rng(n)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    A, X = helpers.random_graph(rng, n, 10, density=0.08)
    W = rng.uniform(0.25, 2.0, (n, n)).astype(np.float32)
    A = A * np.triu(W, 1)
    A = A + A.T + np.diag(rng.uniform(0.5, 1.5, n).astype(np.float32))
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 2, 5, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert job.route()[0] in {24: (6,), 70: (5,), 200: (4, 8)}[n]
    res = job.run([m0], Hyper(num_iters=5))
    o = closed_form.ClosedFormOracle(A, X, sd, 2, sg.pred_label, 5, m0)
    want = o.run(5)                      # explain.py:209-211: masked adjacency of the last forward TIMES the adjacency
    live = (A != 0) & ~np.eye(n, dtype=bool)
    assert np.abs(res.masked_adj[0].astype(np.float64) * A - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5
    assert np.all(np.diag(res.masked_adj[0]) == 0)


def test_large_target_sparse_kernel_with_split_hub_rows(be):
    """n = 600 (beyond the LDS-resident classes): k_sparse_large keeps the edge state in LDS and the row arrays in the
    workspace, updates M / m / v in place on the edges and splits a 150-neighbour hub row over three 64-entry slots."""
    rng = np.random.default_rng(3)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 600
    A, X = helpers.random_graph(rng, n, 10, density=0.004)
    hub = 7
    idx = rng.choice(np.arange(n), 150, replace=False)
    idx = idx[idx != hub]
    A[hub, idx] = 1
    A[idx, hub] = 1
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    t = int(idx[0])                       # a target adjacent to the hub: the hub row is in both row sets
    sg = Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=4))
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, t, m0)
    want = o.run(4)
    live = A != 0
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    assert np.array_equal(res.mask[0][~live], m0[~live])
    assert np.array_equal(res.masked_adj[0], res.masked_adj[0].T)
    dense = be.job([sg], sd, analyze=False).run([m0], Hyper(num_iters=4))      # dense streaming kernels
    assert np.abs(res.masked_adj[0] - dense.masked_adj[0]).max() < 2e-6


def test_large_target_more_than_512_row_slots(be):
    """A target next to a 700-neighbour hub (n = 1400): more than 512 row slots within two hops, so k_sparse_large walks
    its slot records in two rounds; the hub row takes 11 slots of 64 entries."""
    rng = np.random.default_rng(5)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 1400
    A, X = helpers.random_graph(rng, n, 10, density=1.0 / n)
    hub = 11
    idx = rng.choice(np.arange(n), 700, replace=False)
    idx = idx[idx != hub]
    A[hub, idx] = 1
    A[idx, hub] = 1
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    t = int(idx[0])
    sg = Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=3))
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, t, m0)
    want = o.run(3)
    live = A != 0
    assert np.abs(res.masked_adj[0] - want).max() < 1e-5          # a 700-term row sum in another order
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    assert np.array_equal(res.mask[0][~live], m0[~live])


def _hub_case():
    """A k_sparse_large target with far edges (used by tests/test_windowed_parity.py): (state_dict, [Subgraph])."""
    rng = np.random.default_rng(17)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 900
    A, X = helpers.random_graph(rng, n, 10, density=2.4 / n)
    t = int(np.argmax(A.sum(1)))
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    return sd, [Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)]


def test_large_target_far_edges_run_their_own_recursion(be):
    """A sparse graph of large diameter (n = 900, average degree 2.4): most edges have both endpoints more than two hops
    from the target, never see a prediction gradient and are optimised by k_sparse_large outside its iteration loop (a
    closed scalar recursion per mask entry: size + entropy + Laplacian terms through Adam); 12 iterations against the
    closed form on every entry."""
    rng = np.random.default_rng(17)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 900
    A, X = helpers.random_graph(rng, n, 10, density=2.4 / n)
    t = int(np.argmax(A.sum(1)))
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)
    lvl = np.full(n, 9)
    lvl[t] = 0
    for d in (1, 2):
        lvl[(A[lvl == d - 1].sum(0) > 0) & (lvl > d)] = d
    far = (A != 0) & (lvl[:, None] > 2) & (lvl[None, :] > 2)
    assert far.sum() > 0.5 * (A != 0).sum()
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=12))
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, t, m0)
    want = o.run(12)
    live = A != 0
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.masked_adj[0] - want)[far].max() < 1e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    assert np.array_equal(res.mask[0][~live], m0[~live])
    assert np.array_equal(res.masked_adj[0], res.masked_adj[0].T)


@pytest.mark.parametrize("iters", [1, 2])
def test_large_target_weighted_self_loops_and_short_runs(be, iters):
    """k_sparse_large with non-binary symmetric weights and a non-zero diagonal, for 1 / 2 iterations: the returned mask is the
    one of the LAST forward (explain.py:209-211) - the initial one after a single iteration - for the near edges (kept in LDS)
    and for the far edges (their own recursion) alike, while the mask parameters have taken `iters` Adam steps."""
    rng = np.random.default_rng(41)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 700
    A, X = helpers.random_graph(rng, n, 10, density=2.2 / n)
    W = rng.uniform(0.25, 2.0, (n, n)).astype(np.float32)
    A = A * np.triu(W, 1)
    A = A + A.T + np.diag(rng.uniform(0.5, 1.5, n).astype(np.float32))
    t = int(np.argmax((A != 0).sum(1)))
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 3, t, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=iters))
    o = closed_form.ClosedFormOracle(A, X, sd, 3, sg.pred_label, t, m0)
    want = o.run(iters)
    live = (A != 0) & ~np.eye(n, dtype=bool)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    first = 0.5 * (sig(m0) + sig(m0.T)) * A * live                  # A . sym(sigma(M0)), off the diagonal (explain.py:665-678)
    got = res.masked_adj[0].astype(np.float64)
    if iters == 1:
        assert np.abs(got - first).max() < 1e-6
    assert np.abs(got * A - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5
    assert np.all(np.diag(res.masked_adj[0]) == 0) and np.all(got[~live] == 0)


@pytest.mark.parametrize("kinds", [1, 7, 40])
def test_large_target_feature_dictionary(be, kinds):
    """Feature matrices with few distinct rows (constant features: every synthetic configuration of the reference; one-hot /
    categorical ones) are kept by k_sparse_large as a dictionary in LDS - at most 32 rows, found by bit-exact comparison;
    with 40 kinds the kernel must notice and gather the rows from the workspace as for dense features.  Same results."""
    rng = np.random.default_rng(100 + kinds)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 800
    A, _ = helpers.random_graph(rng, n, 10, density=2.5 / n)
    hub = 17
    idx = rng.choice(np.arange(n), 100, replace=False)
    idx = idx[idx != hub]
    A[hub, idx] = 1
    A[idx, hub] = 1
    table = rng.standard_normal((kinds, 10)).astype(np.float32)
    table[0, 3] = -0.0                                  # the comparison is on bit patterns
    X = table[rng.integers(0, kinds, n)]
    t = int(idx[0])
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 2, t, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=4))
    o = closed_form.ClosedFormOracle(A, X, sd, 2, sg.pred_label, t, m0)
    want = o.run(4)
    live = A != 0
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5


def test_large_target_beyond_4095_rows(be):
    """n = 4300 with a 300-neighbour hub next to the target: nothing k_sparse_large keeps in LDS scales with n (only the
    entries of the rows within two hops do), so the target takes the sparse kernel instead of streaming 4320^2 dense
    blocks through every iteration."""
    rng = np.random.default_rng(23)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    n = 4300
    A, X = helpers.random_graph(rng, n, 10, density=2.0 / n)
    hub = 4200
    idx = rng.choice(np.arange(n), 300, replace=False)
    idx = idx[idx != hub]
    A[hub, idx] = 1
    A[idx, hub] = 1
    t = int(idx.max())
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 2, t, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert list(job.route()) == [7]
    res = job.run([m0], Hyper(num_iters=3))
    o = closed_form.ClosedFormOracle(A, X, sd, 2, sg.pred_label, t, m0)
    want = o.run(3)
    live = A != 0
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    assert np.array_equal(res.mask[0][~live], m0[~live])
    assert np.all(res.masked_adj[0][~live] == 0)


@pytest.mark.parametrize("case,n", [("edgeless", 40), ("isolated target", 60), ("isolated target", 700)])
def test_sparse_kernels_degenerate_graphs(be, case, n):
    """No edge at all / a target without neighbours (its row of every layer is the bias direction, the edge masks only
    get their regulariser gradients): the sparse kernels must stay finite and agree with the closed form."""
    rng = np.random.default_rng(n)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    A, X = helpers.random_graph(rng, n, 10, density=0.05 if n < 100 else 0.004)
    if case == "edgeless":
        A = A * 0
    else:
        A[5, :] = 0
        A[:, 5] = 0
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 1, 5, rng.integers(0, 4, n), m0)
    job = be.job([sg], sd)
    assert job.route()[0] >= 4
    res = job.run([m0], Hyper(num_iters=3))
    o = closed_form.ClosedFormOracle(A, X, sd, 1, sg.pred_label, 5, m0)
    want = o.run(3)
    assert np.isfinite(res.masked_adj[0]).all() and np.isfinite(res.feat_mask[0]).all()
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    live = A != 0
    if live.any():
        assert np.abs(res.mask[0] - o.M)[live].max() < 5e-5


def test_mixed_launch_of_large_and_single_tile_targets(be):
    """One launch for a 512-thread target and nine single-tile targets (k_sparse_resident_mixed: eight single-tile targets
    per workgroup, one per wave; the second such workgroup holds a single one): every target must match its solo run bit
    for bit (same code path, other workgroup shape) and the closed form."""
    rng = np.random.default_rng(9)
    sd = helpers.random_model(rng, 10, 20, 20, 4)

    def sub(n, density, t):
        A, X = helpers.random_graph(rng, n, 10, density=density)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        return Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)

    subs = [sub(200, 0.02, 3)] + [sub(int(rng.integers(6, 33)), 0.2, 2) for _ in range(9)]
    job = be.job(subs, sd)
    assert list(job.route()) == [8] + [6] * 9
    hy = Hyper(num_iters=4)
    res = job.run([s.mask0 for s in subs], hy)
    for i, s in enumerate(subs):
        o = closed_form.ClosedFormOracle(s.adj, s.feat, sd, s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(4)).max() < 5e-6
        solo = be.job([s], sd).run([s.mask0], hy)
        assert np.array_equal(solo.masked_adj[0], res.masked_adj[i])
        assert np.array_equal(solo.feat_mask[0], res.feat_mask[i])


@pytest.mark.parametrize("D,H,O", [(10, 16, 16), (8, 20, 12), (10, 32, 32)])
def test_mixed_launch_with_other_encoder_widths(be, D, H, O):
    """Encoders whose widths are not the reference's (hidden = output = 20, D = 10) through the mixed launch - a 512-thread target, a pair
    workgroup, single-tile targets - and k_sparse_large: widths up to the reference's take the <5, 10> instantiations with run-time widths
    (round 5; the 32-wide <16, 16> ones carry 700-1000 B of scratch per lane), wider ones <16, 16>.  Closed form, 3 iterations."""
    rng = np.random.default_rng(D * 100 + H)
    sd = helpers.random_model(rng, D, H, O, 4)

    def sub(n, density, t):
        A, X = helpers.random_graph(rng, n, D, density=density)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        return Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)

    subs = [sub(200, 0.02, 3), sub(70, 0.06, 5), sub(50, 0.1, 0), sub(20, 0.2, 2), sub(600, 0.004, 7)]
    job = be.job(subs, sd)
    assert list(job.route()) == [8, 5, 5, 6, 7]
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=3))
    for i, s in enumerate(subs):
        o = closed_form.ClosedFormOracle(s.adj, s.feat, sd, s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(3)).max() < 5e-6, i


@pytest.mark.parametrize("H", [16, 32])
@pytest.mark.parametrize("const", [False, True])
def test_compile_time_widths_hidden_16_and_32(be, H, const):
    """--hidden-dim = --output-dim = 16 / 32 on 10 input features (round 6): the node-mode resident kernels' compile-time-width instantiations
    <5, 8> / <5, 16> - the mixed launch (a 512-thread target, a pair workgroup, single-wave targets) - in the general form (random feature rows)
    and, with constant feature rows, the algebraic form (until round 6 only the reference's own widths had one).  Closed form, 3 iterations."""
    rng = np.random.default_rng(1000 + H + const)
    sd = helpers.random_model(rng, 10, H, H, 4)
    row = rng.uniform(0.2, 1.5, 10).astype(np.float32)

    def sub(n, density, t):
        A, X = helpers.random_graph(rng, n, 10, density=density)
        if const:
            X = np.tile(row, (n, 1)).astype(np.float32)
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        return Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)

    subs = [sub(180, 0.02, 3), sub(70, 0.06, 5), sub(50, 0.1, 0), sub(20, 0.2, 2), sub(12, 0.2, 1)]
    job = be.job(subs, sd)
    route = list(job.route())
    assert route[0] == 8 and route[-1] == 6 and set(route) <= {5, 6, 8}, route      # (with 33-float rows a target can move up a class)
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=3))
    for i, s in enumerate(subs):
        o = closed_form.ClosedFormOracle(s.adj, s.feat, sd, s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(3)).max() < (2e-5 if const else 5e-6), i
        solo = be.job([s], sd).run([s.mask0], Hyper(num_iters=3))      # the class's own launch: the same body
        assert np.array_equal(solo.masked_adj[0], res.masked_adj[i]), i


def test_pair_workgroups_two_256_thread_targets_per_workgroup(be, monkeypatch):
    """Targets of the 256-thread class (n <= 128) run TWO to a 512-thread workgroup of the mixed launch, each body in its half of the threads
    and of the LDS pool, every __syncthreads() a barrier for both (k_sparse_resident_mixed).  Three of them (an odd count: the last pair
    workgroup holds one body, its other half leaves) beside a 512-thread target and single-tile targets; one of the three has so many
    neighbours that its layer-2 rows do not fit one wave, so its partner must drop the single-wave fusion too (the joint `fuseB`).  Every
    target must match its solo run - the stand-alone 256-thread launch - bit for bit, and the closed form; a batch of 256-thread and
    single-tile targets alone (no larger one) pairs as well; GNNX_PAIR_256=0 restores the round-4 routing."""
    rng = np.random.default_rng(11)
    sd = helpers.random_model(rng, 10, 20, 20, 4)

    def sub(n, density, t, hub=0):
        A, X = helpers.random_graph(rng, n, 10, density=density)
        if hub:      # target t with `hub` neighbours: more rows in the layer-2 row set than one wave has slots
            A[t, :] = 0
            A[:, t] = 0
            nb = rng.choice(np.setdiff1d(np.arange(n), [t]), hub, replace=False)
            A[t, nb] = 1
            A[nb, t] = 1
        m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
        return Subgraph(A, X, 1, t, rng.integers(0, 4, n), m0)

    subs = [sub(200, 0.02, 3), sub(70, 0.06, 5), sub(100, 0.05, 7, hub=40), sub(50, 0.1, 0)] + [sub(int(rng.integers(6, 33)), 0.2, 2) for _ in range(3)]
    job = be.job(subs, sd)
    assert list(job.route()) == [8, 5, 5, 5, 6, 6, 6]
    hy = Hyper(num_iters=4)
    res = job.run([s.mask0 for s in subs], hy)
    for i, s in enumerate(subs):
        o = closed_form.ClosedFormOracle(s.adj, s.feat, sd, s.gt_label, s.pred_label, s.target_row, s.mask0)
        assert np.abs(res.masked_adj[i] - o.run(4)).max() < 5e-6, i
        solo = be.job([s], sd).run([s.mask0], hy)
        assert np.array_equal(solo.masked_adj[0], res.masked_adj[i]), i
        assert np.array_equal(solo.feat_mask[0], res.feat_mask[i]), i
    few = subs[1:3] + subs[4:6]          # no 512-thread target: the two 256-thread targets still share a workgroup of the mixed launch
    job2 = be.job(few, sd)
    assert list(job2.route()) == [5, 5, 6, 6]
    res2 = job2.run([s.mask0 for s in few], hy)
    for i, k in enumerate((1, 2, 4, 5)):
        assert np.array_equal(res2.masked_adj[i], res.masked_adj[k]) and np.array_equal(res2.feat_mask[i], res.feat_mask[k])
    monkeypatch.setenv("GNNX_PAIR_256", "0")
    assert list(be.job(subs, sd).route()) == [8, 8, 8, 8, 6, 6, 6]


def test_device_side_engine_walk_draws_the_reference_masks_on_the_edges(be):
    """The seeded initial masks with the mt19937 walk on the DEVICE (gnnx_mt_edge_words: k_mt_stream + k_mt_gather_pairs) and only the
    transform of the picked word pairs on the host (gnnx_host_transform_edge_words, through ATen's own normal_): on every edge entry the
    value torch.manual_seed(seed); torch.FloatTensor(n, n).normal_(1.0, std) puts there, bit for bit - for streams below 16 values (the
    scalar path), of exactly 16, ragged ones (the redrawn tail: entries among the last 16 positions), multiples of 16, and streams of
    several engine blocks whose 16-value groups straddle nothing (624 = 39 x 16)."""
    if not engine.pair_staging_ok():
        pytest.skip("the host's normal_ lacks the pair-staging property (the pipeline then keeps the host walk)")
    rng = np.random.default_rng(21)
    sd = helpers.random_model(rng, 10, 20, 20, 4)
    sizes = [3, 4, 5, 7, 16, 31, 32, 33, 50, 64, 70, 97, 130]
    subs = []
    for n in sizes:
        A, X = helpers.random_graph(rng, n, 10, density=min(0.9, 6.0 / n))
        A[n - 1, n - 2] = A[n - 2, n - 1] = 1.0      # an entry among the last positions of the stream: the tail rule of ragged streams
        A[0, 1] = A[1, 0] = 1.0
        subs.append(Subgraph(A, X, 1, 0, rng.integers(0, 4, n), None))
    job = be.job(subs, sd)
    seeds = 1000 + np.arange(len(sizes)) * 7
    words = job.draw_edge_words_device(seeds)
    em_rc = job._rc.cpu().numpy()
    eoff = job._eoff
    got = engine.transform_edge_words(sizes, seeds, eoff, em_rc, words.cpu().contiguous(), threads=3)
    raw = engine.init_edge_masks_raw(sizes, seeds=seeds)
    off = 0
    for k, n in enumerate(sizes):
        full = raw[off:off + n * n].view(n, n)
        off += n * n
        a, b = int(eoff[k]), int(eoff[k + 1])
        assert b > a
        r, c = em_rc[a:b, 0], em_rc[a:b, 1]
        assert torch.equal(got[a:b, 0], full[r, c]) and torch.equal(got[a:b, 1], full[c, r]), (k, n)
    # and it is what the host walk gives (the path it replaces in the pipeline)
    assert torch.equal(got, engine.init_edge_masks_on_edges(sizes, seeds, eoff, em_rc, threads=2))


def test_device_khop_equals_reference_neighbor_lists(be):
    """gnnx_khop vs the neighbour lists of the reference's own neighborhoods / extract_neighborhood (stored by
    tests/golden/make_golden_full.py for every syn1 motif node) - bit for bit, including node_idx_new - and vs the host
    walk sets on nodes of the BA part (hubs, larger sets)."""
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    ck = helpers.load_ckpt("syn1")
    z = np.load(helpers.GOLDEN + "/syn1_full_explain.npz")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"], device=be.device)
    sel = np.arange(0, len(z["targets"]), 1 if be.name == "gpu" else 16)
    targets = z["targets"][sel]
    dn = engine.khop_device(graph, targets, 3, lib=be.lib)
    for k, (s, nb) in enumerate(zip(sel, dn.lists())):
        assert np.array_equal(nb, z["nb_flat"][z["nb_off"][s]:z["nb_off"][s + 1]])
        assert dn.rows[k] == z["node_idx_new"][s] and dn.sizes[k] == len(nb)
    more = np.asarray([0, 1, 17, 150, 299])
    dn = engine.khop_device(graph, more, 3, lib=be.lib)
    for v, nb, row in zip(more, dn.lists(), dn.rows):
        want = idx.neighbors(int(v))
        assert np.array_equal(nb, want) and row == np.searchsorted(want, v)
    for hops in (1, 2):          # other depths; with one hop a node is not in its own set
        dn = engine.khop_device(graph, more, hops, lib=be.lib)
        for v, nb, row in zip(more, dn.lists(), dn.rows):
            want = KHopIndex(idx.csr, hops).neighbors(int(v))
            assert np.array_equal(nb, want) and row == (np.searchsorted(want, v) if v in want else -1)


def test_one_pass_khop_equals_two_passes(be):
    """khop_device(one_pass=True): ONE launch, lists at fixed-stride offsets, sizes and rows from the same pass - same lists, same packed batch."""
    import torch
    ck = helpers.load_ckpt("syn1")
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"], device=be.device)
    targets = np.asarray([302, 309, 555, 300, 699, 420, 301])
    two = engine.khop_device(graph, targets, 3, lib=be.lib)
    one = engine.khop_device(graph, targets, 3, lib=be.lib, one_pass=True)
    assert np.array_equal(one.sizes, two.sizes) and np.array_equal(one.rows, two.rows)
    assert np.array_equal(one.nb_off.cpu().numpy(), np.arange(len(targets) + 1) * graph.num_nodes)
    for a, b in zip(one.lists(), two.lists()):
        assert np.array_equal(a, b)
    ja = engine.MaskOptimJob.from_csr(graph, one, None, ck["label"][targets], ck["sd"], lib=be.lib)
    jb = engine.MaskOptimJob.from_csr(graph, two, None, ck["label"][targets], ck["sd"], lib=be.lib)
    assert torch.equal(ja.A.cpu(), jb.A.cpu()) and torch.equal(ja.X.cpu(), jb.X.cpu()) and torch.equal(ja.yhat.cpu(), jb.yhat.cpu())
    assert np.array_equal(ja.route(), jb.route())
    assert engine.khop_device(graph, targets, 3, lib=be.lib, one_pass=True).nb_off is one.nb_off      # the offsets are cached per graph
    cap, engine._ONE_PASS_MAX_INTS = engine._ONE_PASS_MAX_INTS, len(targets) * graph.num_nodes - 1   # padded lists beyond the cap: two passes, compact lists
    try:
        back = engine.khop_device(graph, targets, 3, lib=be.lib, one_pass=True)
    finally:
        engine._ONE_PASS_MAX_INTS = cap
    assert np.array_equal(back.nb_off.cpu().numpy(), two.nb_off.cpu().numpy()) and np.array_equal(back.rows, two.rows)


def test_device_khop_isolated_node_has_empty_set(be):
    """Reference semantics (utils/graph_utils.py:152-157): no explicit self term, so an isolated node has an EMPTY set."""
    import scipy.sparse as sp
    n = 70
    rng = np.random.default_rng(1)
    A, X = helpers.random_graph(rng, n, 4, density=0.05)
    A[5, :] = 0
    A[:, 5] = 0
    graph = engine.device_graph(sp.csr_matrix(A), X, None, device=be.device)
    dn = engine.khop_device(graph, np.asarray([5, 6]), 3, lib=be.lib)
    assert dn.sizes[0] == 0 and dn.rows[0] == -1 and dn.sizes[1] > 0


def test_raw_mask_stream_and_edge_lists(be):
    """gnnx_scatter_masks spreads the host's RNG stream (one contiguous n x n draw per target) exactly like the host
    packer; gnnx_gather_edges returns exactly the non-zero entries of the dense result, in row-major order."""
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    ck = helpers.load_ckpt("syn1")
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    targets = np.asarray([302, 309, 555, 300, 699])
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"], device=be.device)
    dn = engine.khop_device(graph, targets, 3, lib=be.lib)
    job = engine.MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"], lib=be.lib)
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
    M = job.M.cpu().numpy()
    for v, n, t in zip(job._square_views(M), dn.sizes, targets):
        assert np.array_equal(v[:n, :n], helpers.seeded_mask0(int(t), int(n)).numpy()) and not v[n:].any() and not v[:, n:].any()
    hy = Hyper(num_iters=3)
    job.launch(hy)
    dense, em = job.fetch(hy), job.fetch_edges(with_mask=True)
    for k in range(len(targets)):
        assert np.array_equal(em.dense(k, np.float32), dense.masked_adj[k])
        a, b = em.eoff[k], em.eoff[k + 1]
        r, c = em.rc[a:b, 0], em.rc[a:b, 1]
        rr, cc = np.nonzero(np.triu(job.adjacency()[k], 1))
        assert np.array_equal(r, rr) and np.array_equal(c, cc)
        assert np.array_equal(em.mask_rc[a:b, 0], dense.mask[k][r, c]) and np.array_equal(em.mask_rc[a:b, 1], dense.mask[k][c, r])
    job.set_masks_raw_resident()            # a second run from the resident stream reproduces the first bit for bit
    job.launch(hy)
    again = job.fetch_edges()
    assert np.array_equal(again.masked_adj, em.masked_adj) and np.array_equal(again.feat_mask, em.feat_mask)


def test_gradient_baseline_vs_reference(be):
    """model="grad" (explain.py:125-133, adj_feat_grad :717-738): sigmoid(|dL/dA| + |dL/dA|^T) * A of the unmasked
    sub-graph against the REAL reference's outputs (tests/golden/flags_explain.npz), n = 6 ... 310, as one batched job."""
    z = np.load(helpers.GOLDEN + "/flags_explain.npz")
    ck = helpers.load_ckpt("syn1")
    targets = [302, 309, 555] if be.name == "emu" else [302, 309, 555, 330, 400, 300]
    subs = []
    for t in targets:
        nb = z[f"grad:{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(np.searchsorted(nb, t))
        subs.append(Subgraph(A, X, int(yhat[new]), new, yhat, None))        # label = the PREDICTED label of the target (explain.py:130)
    job = be.job(subs, ck["sd"], analyze=False)
    for s, t, got in zip(subs, targets, job.grad_baseline()):
        r, c = np.nonzero(np.triu(s.adj, 1))
        assert np.array_equal(got, got.T) and np.all(got[s.adj == 0] == 0)
        want = z[f"grad:{t}:masked_adj_edges"]
        assert np.abs(got[r, c] - want).max() <= 1e-6, (t, np.abs(got[r, c] - want).max())
        # the information is in the logit |G_ij| + |G_ji| (the sigmoid squeezes it into [0.5, 0.54]): compare it too
        logit = lambda v: np.log(v.astype(np.float64) / (1.0 - v.astype(np.float64)))
        assert np.abs(logit(got[r, c]) - logit(want)).max() <= 5e-6


@pytest.mark.parametrize("graph_mode", [False, True])
def test_mask_act_relu_matches_restatement_and_reproduces_nan(be, graph_mode):
    """mask_act="ReLU" (explain.py:669-670, 757-760) on the dense streaming kernels: relu(M) in the masked adjacency, the
    size term and the entropy term.  With every mask entry inside (0, 1) the run is finite and must match the torch
    restatement of the reference; with the reference's own N(1, .) initialisation the entropy's log(1 - relu(M)) is NaN for
    every entry > 1 and - exactly like the reference (tests/golden/flags_explain.npz: 100 % NaN) - the result is all NaN."""
    import torch
    from oracle import reference_restatement as rr
    rng = np.random.default_rng(11)
    D, H, O, C, n = (14, 20, 20, 2, 40) if graph_mode else (10, 20, 20, 4, 37)
    sd = helpers.random_model(rng, D, H, O, C)
    A, X = helpers.random_graph(rng, n, D, density=0.1)
    yhat = None if graph_mode else rng.integers(0, C, n)
    m0 = rng.uniform(0.3, 0.7, (n, n)).astype(np.float32)
    sg = Subgraph(A, X, 1, 0 if graph_mode else 3, yhat, m0)
    iters = 3
    job = engine.MaskOptimJob([sg], sd, graph_mode=graph_mode, device=be.device, lib=be.lib, mask_relu=True)
    assert job.route()[0] == 0                                   # dense streaming kernels
    res = job.run([m0], Hyper(num_iters=iters))
    o = rr.MaskOptimOracle(torch.tensor(A), torch.tensor(X), {k: torch.tensor(v) for k, v in sd.items()}, 1, yhat, sg.target_row,
                           graph_mode=graph_mode, mask0=torch.tensor(m0), mask_act="ReLU")
    want = o.run(iters)
    assert np.isfinite(want).all()
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.mask.detach().numpy()).max() < 5e-5
    assert np.abs(res.feat_mask[0] - o.feat_mask.detach().numpy()).max() < 5e-5
    bad = helpers.seeded_mask0(5, n).numpy()                     # N(1, sqrt(2 / n)): about half of the entries exceed 1
    res = job.run([bad], Hyper(num_iters=4))
    assert np.isnan(res.masked_adj[0]).all()


def test_device_denoise_and_auc_equal_reference_postprocessing(be):
    """gnnx_denoise_edges vs the node / edge sets the REAL reference's io_utils.denoise_graph(threshold_num=20) produced
    on its own masks (stored in tests/golden/syn1_explain.npz), and gnnx_auc_counts vs sklearn's roc_auc_score on the same
    scores - the post-processing of explain_nodes_gnn_stats (explain.py:306-351) without leaving the device."""
    import torch
    from sklearn.metrics import roc_auc_score
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    targets = [int(t) for t in gx["targets"]]
    subs = []
    for t in targets:
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        subs.append(Subgraph(A, X, int(lab[new]), new, yhat, None))
    job = be.job(subs, ck["sd"], analyze=False)
    job._edge_layout()
    # feed the REFERENCE's masks (edge values in the layout of gather_edges_device: upper triangle, row-major)
    vals = []
    for t, s in zip(targets, subs):
        rc = gx[f"{t}:edge_rc"]
        d = np.zeros_like(s.adj)
        d[rc[:, 0], rc[:, 1]] = gx[f"{t}:masked_adj_edges"]
        r, c = np.nonzero(np.triu(s.adj, 1))
        vals.append(d[r, c])
    vals = np.concatenate(vals).astype(np.float32)
    vals_d = torch.from_numpy(vals).to(job.device)
    keep, thr, stats = job.denoise(20, vals_d)
    rc_all = job._rc.cpu().numpy()
    for k, t in enumerate(targets):
        a, b = job._eoff[k], job._eoff[k + 1]
        kept = rc_all[a:b][keep[a:b]]
        want_e = gx[f"{t}:denoised_edges"]
        assert np.array_equal(kept[np.lexsort((kept[:, 1], kept[:, 0]))], want_e), t
        assert np.array_equal(np.unique(kept), gx[f"{t}:denoised_nodes"]) and stats[k][0] == len(gx[f"{t}:denoised_nodes"])
        assert stats[k][1] == len(want_e)
        pos = np.sort(vals[a:b][vals[a:b] > 0])
        assert thr[k] == pos[-min(20, len(pos))]
    # AUC: motif ground truth of make_pred_real (explain.py:535-579) as 0/1 per edge
    real = np.zeros(len(vals), np.uint8)
    motif = {(0, 1), (1, 2), (2, 3), (0, 3), (0, 4), (1, 4)}
    for k, (t, s) in enumerate(zip(targets, subs)):
        a, b = job._eoff[k], job._eoff[k + 1]
        st = s.target_row
        for e in range(a, b):
            r, c = rc_all[e]
            real[e] = (int(r) - st, int(c) - st) in motif
    auc, P, N = job.auc(real, vals_d)
    assert P == int(real.sum()) and N == len(real) - P
    assert abs(auc - roc_auc_score(real, vals)) < 1e-12


@pytest.mark.parametrize("graph_mode", [False, True])
def test_batch_norm_matches_closed_form(be, graph_mode):
    """--bn (apply_bn, models.py:222-228, 241-253): every node's hidden activation standardised over its features after the
    ReLU of the two hidden layers - forward, backward through the standardisation, G product and heads on the dense
    streaming kernels vs the closed form (itself checked against torch autograd and, through the restatement, bit-pinned
    to the reference by tests/golden/make_golden_flags.py)."""
    rng = np.random.default_rng(23)
    D, H, O, C, n = (14, 20, 20, 2, 40) if graph_mode else (10, 20, 20, 4, 70)
    sd = helpers.random_model(rng, D, H, O, C)
    A, X = helpers.random_graph(rng, n, D, density=0.08)
    yhat = None if graph_mode else rng.integers(0, C, n)
    m0 = (1.0 + rng.standard_normal((n, n)) * np.sqrt(2.0 / n)).astype(np.float32)
    sg = Subgraph(A, X, 1, 0 if graph_mode else 9, yhat, m0)
    iters = 5
    job = engine.MaskOptimJob([sg], sd, graph_mode=graph_mode, device=be.device, lib=be.lib, bn=True)
    assert job.route()[0] == 0
    res = job.run([m0], Hyper(num_iters=iters))
    o = closed_form.ClosedFormOracle(A, X, sd, 1, yhat, sg.target_row, m0, graph_mode=graph_mode, bn=True)
    want = o.run(iters)
    assert np.abs(res.masked_adj[0] - want).max() < 5e-6
    assert np.abs(res.mask[0] - o.M).max() < 5e-5 and np.abs(res.feat_mask[0] - o.f).max() < 5e-5
    plain = engine.MaskOptimJob([sg], sd, graph_mode=graph_mode, device=be.device, lib=be.lib, analyze=False).run([m0], Hyper(num_iters=iters, use_resident=False))
    assert np.abs(plain.masked_adj[0] - res.masked_adj[0]).max() > 1e-4          # the flag does change the answer


def test_batch_norm_vs_reference_golden(be):
    """The REAL reference run with --bn (tests/golden/flags_explain.npz): 300 epochs on syn1 targets (n = 6, 48, 104)."""
    if be.name == "emu":
        pytest.skip("300 epochs of the dense streaming kernels: hardware only (the emulator covers short runs above)")
    z = np.load(helpers.GOLDEN + "/flags_explain.npz")
    ck = helpers.load_ckpt("syn1")
    targets = [302, 309, 555]
    subs = []
    for t in targets:
        nb = z[f"bn:{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(np.searchsorted(nb, t))
        subs.append(Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy()))
    job = engine.MaskOptimJob(subs, ck["sd"], device=be.device, lib=be.lib, bn=True)
    res = job.run([s.mask0 for s in subs], Hyper(num_iters=300, use_graph=True))
    for s, t, ma, fm in zip(subs, targets, res.masked_adj, res.feat_mask):
        assert float(z[f"bn:{t}:cond"].max()) < 2e-6                             # well conditioned (CPU vs CPU)
        r, c = np.nonzero(np.triu(s.adj, 1))
        assert np.abs(ma[r, c] - z[f"bn:{t}:masked_adj_edges"]).max() <= 1e-5
        assert np.abs(1 / (1 + np.exp(-fm.astype(np.float64))) - z[f"bn:{t}:feat_sig"]).max() <= 1e-5


def test_edge_only_results_equal_dense_results(be):
    """Hyper.edge_results_only (gnnx_hyper.edge_results_only): the edge-sparse kernels skip the ld^2 zero-fill of every target's Abar block
    and write it on the edges only; the edge lists must be bit-identical to those of a dense-result run (and fetch() refuses)."""
    ck = helpers.load_ckpt("syn1")
    subs = [_node_case("syn1", t)[2] for t in (302, 309, 330)]
    outs = []
    for eo in (False, True):
        job = be.job(subs, ck["sd"])
        job.set_masks([s.mask0 for s in subs])
        job.Abar.fill_(float("nan"))                  # whatever the kernels do not write stays NaN
        hy = Hyper(num_iters=7, edge_results_only=eo)
        job.launch(hy)
        em = job.fetch_edges(with_mask=True)
        outs.append(em)
        if eo:
            with pytest.raises(ValueError, match="edges only"):
                job.fetch(hy)
        else:
            assert np.isfinite(job.fetch(hy).masked_adj[0]).all()
    assert np.array_equal(outs[0].masked_adj, outs[1].masked_adj) and np.array_equal(outs[0].mask_rc, outs[1].mask_rc)
    assert np.array_equal(outs[0].feat_mask, outs[1].feat_mask) and np.isfinite(outs[1].masked_adj).all()
