"""Windowed ("teacher-forced") parity against the optimiser state of the LIVE reference, and the resume entry point itself.

tests/golden/<name>_windows.npz (tests/golden/make_golden_windows.py) holds, for EVERY target of syn1 / syn4 / syn5 and for 512
graphs of the config-4 job, the state of the torch.optim.Adam the reference builds (mask, exp_avg, exp_avg_sq on the edges,
feature mask and its moments) after 50, 100, ..., 300 epochs of Explainer.explain (explain.py:137-146).  The engine is started
from the reference's state at a boundary (gnnx_run_resume) and must reproduce the reference's state 50 epochs later within
1e-5 (masked adjacency from the mask entries of both directions, and sigmoid(feat_mask)).  Round-off then has 50 iterations
to act instead of 300, so this covers iterations 50..300 of the targets whose full-horizon outcome is chaotic - 561 of the 720
syn5 targets that the full-horizon rule (test_gpu_full_configs.py) can only report.

The classification is outcome-blind: make_golden_windows.py runs the closed-form fp32 oracle over the same windows on the CPU
and flags the windows where two CPU implementations already disagree by more than 2e-6 (a ReLU gate or a max-pool tie flips
inside them); those windows are tested at 10-epoch granularity from the reference's 10-epoch snapshots, and the handful of
10-epoch sub-windows the two CPU implementations still disagree on (syn5: 25 of 5835) are reported, bounded, not gated at 1e-5.

  * every route, emulator + GPU: a run split into resumed segments is BIT-identical to the straight run (state in == state out);
  * emulator (CPU suite): a few targets per config through all six windows;
  * GPU: every target, every window, every config.
"""
import os

import numpy as np
import pytest

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
from test_emu_kernels import _Backend, _node_case

TOL = helpers.WIN_TOL
SUB_FLAG_BOUND = 5e-3      # a 10-epoch sub-window two CPU implementations disagree on: bounded by the largest branch jump (helpers.BRANCH_JUMP_MAX)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


# ------------------------------------------------------------------ the resume entry point ------------------------------------------------------------------
def _split_equals_straight(be, subs, sd, analyze, use_resident, graph_mode=False, total=12, splits=(5, 4, 3), routes=None):
    outs = []
    for parts in ((total,), splits):
        job = engine.MaskOptimJob(subs, sd, graph_mode=graph_mode, device=be.device, lib=be.lib, analyze=analyze)
        if routes is not None:
            assert sorted(set(job.route())) == sorted(set(routes)), job.route()
        job.set_masks([s.mask0 for s in subs])
        st, done = None, 0
        for k in parts:
            job.launch(Hyper(num_iters=k, use_resident=use_resident), state=st, keep_state=True)
            st = job.state_out
            done += k
            assert st.first_iter == done
        em = job.fetch_edges(with_mask=True)
        outs.append((em.masked_adj, em.mask_rc, em.feat_mask) + job.fetch_state_edges()[1:])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert np.abs(outs[0][3]).max() > 0 and np.abs(outs[0][4]).max() > 0      # the moments really travelled


@pytest.mark.parametrize("analyze,use_resident,routes", [(True, True, (5, 6, 8)), (False, True, (1, 2)), (False, False, None)])
def test_resumed_segments_equal_straight_run_node(be, analyze, use_resident, routes):
    """Sparse resident (64-, 256- and 512-thread classes in one mixed launch: single-tile waves, a pair workgroup, a whole workgroup), dense
    resident (1 and 2 row blocks), streaming."""
    from test_logging_and_trace import _largest_syn1_case
    ck = helpers.load_ckpt("syn1")
    subs = [_node_case("syn1", 302)[2], _node_case("syn1", 309)[2], _node_case("syn4", 511)[2]] + ([_largest_syn1_case()] if analyze else [])
    _split_equals_straight(be, subs, ck["sd"], analyze, use_resident, routes=routes)


def test_resumed_segments_equal_straight_run_large_class(be):
    """k_sparse_large (hub rows, far edges as closed recursions in registers): near and far edges both resume."""
    from test_emu_kernels import _hub_case
    sd, subs = _hub_case()
    _split_equals_straight(be, subs, sd, True, True, total=6, splits=(2, 4), routes=(7,))


@pytest.mark.parametrize("analyze", [True, False])
def test_resumed_segments_equal_straight_run_graph_mode(be, analyze):
    z = np.load(helpers.GOLDEN + "/graphmode_explain.npz")
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    subs = [Subgraph(z["adj"][g], z["feat"][g], int(z["label"][g]), 0, None, helpers.seeded_mask0(g, 100).numpy()) for g in (1, 4)]
    _split_equals_straight(be, subs, sd, analyze, True, graph_mode=True, total=6, splits=(4, 2))


# ------------------------------------------------------------------ windows vs the reference's own state ------------------------------------------------------------------
def _verdict(what, rows, bound=SUB_FLAG_BOUND, list_name=None):
    """rows: (key, window, sub or -1, err, conditioning) of every tested (target, window[, sub-window]).
    Windows the CPU probes call well conditioned (conditioning <= 2e-6: helpers.Windows) must lie within 1e-5 of the reference's state; one that does
    not passes only if the DECISION suite has explained that very window (tests/golden/<list_name>_ties.json, part "windows": the engine's first
    differing decision there is a tie of the reference, or every decision is identical and the drift is within 4 x the window's conditioning - round 6:
    the 99.5 % share of rounds 3-5 is gone) and it stays within SUB_FLAG_BOUND.  Sub-windows the probes flag are reported and bounded.
    -> summary string (asserts on failure)."""
    rows = np.asarray(rows, np.float64).reshape(-1, 5)
    agreed = rows[:, 4] <= helpers.WIN_FLAG
    n_ok = int((agreed & (rows[:, 3] <= TOL)).sum())
    bad = rows[agreed & (rows[:, 3] > TOL)]
    loose = rows[~agreed]
    coarse = rows[:, 2] < 0
    msg = (f"{what}: {int(agreed.sum())} well-conditioned windows ({int((agreed & coarse).sum())} of 50 epochs, {int((agreed & ~coarse).sum())} of 10), "
           f"{n_ok} within 1e-5 ({100.0 * n_ok / max(1, int(agreed.sum())):.2f} %, worst {rows[agreed, 3].max() if agreed.any() else 0:.2e}); "
           f"{len(loose)} flagged 10-epoch sub-windows: {int((loose[:, 3] <= TOL).sum())} within 1e-5, worst {loose[:, 3].max() if len(loose) else 0:.2e}")
    print(msg)
    if len(bad):
        print(f"{what}: beyond 1e-5 (id, window, sub-window, error, measured conditioning): "
              f"{[(int(r[0]), int(r[1]), int(r[2]), float('%.2e' % r[3]), float('%.1e' % r[4])) for r in bad[:40]]}")
    dump = os.environ.get("GNNX_DUMP_WINDOWS")
    if dump:       # measurement aid: every (id, window, sub-window, error, conditioning) row of this config
        os.makedirs(dump, exist_ok=True)
        np.save(os.path.join(dump, what.split(" ")[0] + "_windows_rows.npy"), rows)
    if len(bad):
        ties = helpers.load_ties(list_name) if list_name else None
        assert list_name is None or ties is not None or os.environ.get("GNNX_WRITE_TIES") == "1", f"{helpers.ties_path(list_name)} missing"
        listed = set() if ties is None else ({(i, w) for (i, w, _) in ties["windows"]} | set(ties["resolved"]))
        unexplained = [(int(r[0]), int(r[1]), int(r[2]), float(r[3])) for r in bad if (int(r[0]), int(r[1])) not in listed]
        # (a partial run - the emulator's handful of targets - has no list: every well-conditioned window must then be inside)
        assert not unexplained or os.environ.get("GNNX_WRITE_TIES") == "1", msg + f"; well-conditioned windows beyond 1e-5 that the decision suite's list does not explain: {unexplained[:10]}"
    assert not len(bad) or bad[:, 3].max() <= bound, msg
    assert not len(loose) or loose[:, 3].max() <= bound, msg
    return msg


def _windows_of_job(W, make_job, ks_all, what, coarse_windows=None, bound=SUB_FLAG_BOUND, list_name=None):
    """All 50-epoch windows of the targets ks_all (fixture indices) + the 10-epoch sub-windows of their flagged windows.
    make_job(ks) -> a MaskOptimJob over those targets with the seeded initial masks in M."""
    rows = []
    job = make_job(ks_all)
    eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[ks_all])])
    for w in (range(W.W) if coarse_windows is None else coarse_windows):
        mask_rc, feat = helpers.run_window(job, W.boundary(w, ks_all), W.win)
        em, ef = helpers.window_errors(eoff, mask_rc, feat, W.boundary(w + 1, ks_all))
        for i, k in enumerate(ks_all):
            if not W.flagged[k, w]:
                rows.append((W.ids[k], w, -1, max(em[i], ef[i]), W.cond50[k, w]))
        ks = np.asarray([k for k in ks_all if W.flagged[k, w]], np.int64)
        if not len(ks):
            continue
        sub_job = make_job(ks)     # fresh: sub-window 0 of window 0 starts from the seeded initial masks
        sub_eoff = np.concatenate([[0], np.cumsum(np.diff(W.eoff)[ks])])
        c10 = W.cond10(w, ks)
        for s in range(W.nsub):
            mask_rc, feat = helpers.run_window(sub_job, W.sub_state(w, s, ks), W.sub)
            em, ef = helpers.window_errors(sub_eoff, mask_rc, feat, W.sub_state(w, s + 1, ks))
            rows += [(W.ids[k], w, s, max(em[i], ef[i]), c10[i, s]) for i, k in enumerate(ks)]
    return _verdict(what, rows, bound, list_name)


def _node_subgraph_job(be, name, W, full):
    ck = helpers.load_ckpt(name)
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)

    def make(ks):
        subs = []
        for k in ks:
            t = int(W.ids[k])
            nb = full["nb_flat"][full["nb_off"][k]:full["nb_off"][k + 1]].astype(np.int64)
            A, X, lab, yhat = helpers.subgraph(ck, nb)
            new = int(full["node_idx_new"][k])
            subs.append(Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy()))
        job = be.job(subs, ck["sd"])
        job.set_masks([s.mask0 for s in subs])
        return job
    return make


@pytest.mark.parametrize("name,picks", [("syn4", 3), ("syn5", 4), ("syn1", 2)])
def test_windows_on_the_emulator_few_targets(name, picks):
    """CPU suite: the product's kernel sources on the emulator through all six windows of a few targets per dataset - one of
    them with a flagged window whenever the dataset has one small enough (10-epoch sub-windows from the fine snapshots)."""
    be = _Backend("emu")
    W = helpers.Windows(name)
    full = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    size = np.diff(full["nb_off"])
    small = np.nonzero(size <= (60 if name == "syn1" else 40))[0]
    fl = [k for k in small if W.flagged[k].any()]
    ks = list(small[np.linspace(0, len(small) - 1, picks).astype(int)])
    if fl:
        ks[-1] = fl[0]
    ks = np.asarray(sorted(set(int(k) for k in ks)), np.int64)
    _windows_of_job(W, _node_subgraph_job(be, name, W, full), ks, f"{name} (emulator, targets {[int(W.ids[k]) for k in ks]})")


def test_windows_graph_mode_on_the_emulator():
    be = _Backend("emu")
    W = helpers.Windows("config4")
    from gnn_model_explainer_amd.utils import synthetic
    sd = {k[2:]: W.z[k] for k in W.z.files if k.startswith("w:")}
    A, X, nn, y = synthetic.molecule_like_graphs(int(W.ids.max()) + 1, seed=0)
    ks = np.asarray([0, int(np.nonzero(W.flagged.any(1))[0][0])], np.int64)

    def make(kk):
        subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(g, A.shape[1]).numpy()) for g in W.ids[kk]]
        job = be.job(subs, sd, graph_mode=True)
        job.set_masks([s.mask0 for s in subs])
        return job
    _windows_of_job(W, make, ks, "config4 (emulator)", coarse_windows=(0, 3), bound=helpers.CONFIG4_WINDOW_JUMP)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["syn1", "syn4", "syn5"])
def test_windows_every_target_every_window_gpu(name):
    """BASELINE configs 2 and 3: ALL 400 / 360 / 720 targets x 6 windows as batched jobs through the device-side pipeline."""
    from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
    W = helpers.Windows(name)
    ck = helpers.load_ckpt(name)
    full = np.load(os.path.join(helpers.GOLDEN, name + "_full_explain.npz"))
    assert np.array_equal(W.ids, full["targets"]) and np.array_equal(W.eoff, full["eoff"])
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])

    def make(ks):
        targets = W.ids[ks]
        nbs = [full["nb_flat"][full["nb_off"][k]:full["nb_off"][k + 1]].astype(np.int64) for k in ks]
        job = MaskOptimJob.from_csr(graph, nbs, full["node_idx_new"][ks], ck["label"][targets], ck["sd"])
        job.set_masks_raw(engine.init_edge_masks_raw([len(nb) for nb in nbs], seeds=1000 + targets))
        assert np.array_equal(np.diff(job.fetch_edges().eoff), np.diff(W.eoff)[ks])
        return job
    _windows_of_job(W, make, np.arange(W.T), name, list_name=name)


@pytest.mark.gpu
def test_windows_config4_512_graphs_gpu():
    """BASELINE config 4 (graph mode): 512 size-stratified graphs of the 4337-graph job x 6 windows."""
    from gnn_model_explainer_amd.utils import synthetic
    W = helpers.Windows("config4")
    sd = {k[2:]: W.z[k] for k in W.z.files if k.startswith("w:")}
    A, X, nn, y = synthetic.molecule_like_graphs(int(W.ids.max()) + 1, seed=0)

    def make(ks):
        subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(g, A.shape[1]).numpy()) for g in W.ids[ks]]
        job = MaskOptimJob(subs, sd, graph_mode=True)
        job.set_masks([s.mask0 for s in subs])
        return job
    # graph mode: a max-pool tie that flips moves the mask by up to 6e-2 (helpers.CONFIG4_WINDOW_JUMP, measured on the CPU in round 2)
    _windows_of_job(W, make, np.arange(W.T), "config4", bound=helpers.CONFIG4_WINDOW_JUMP, list_name="config4")


# ------------------------------------------------------------------ k_sparse_large window by window against the dense streaming kernels ------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["spread of the 2048-target sample", "six largest of the 16384-target set"])
def test_windows_sparse_large_against_the_streaming_kernels_gpu(which):
    """The reference cannot be run on the BA-House x100k graph (its dense 100k x 100k neighbourhood matrix), so k_sparse_large - the
    kernel of the scaling workload's largest targets - is pinned over the WHOLE horizon to the dense streaming kernels (k_conv / k_mask:
    every entry of the dense mask, no sparsity shortcut, themselves pinned to the reference window by window on configs 2-4): the
    streaming route is the teacher, its optimiser state after 0, 50, ..., 250 iterations starts k_sparse_large (gnnx_run_resume), and
    after 50 iterations each the masked adjacency on the edges and the mask parameters have to agree.  Twelve targets of the
    2048-target sample routed to k_sparse_large, from the smallest to the largest (n = 320 ... 2460).  Measured: 72 / 72 windows within
    1e-5 (masked adjacency 8e-7, mask parameters 3.6e-6 at worst)."""
    import torch
    import bench
    from gnn_model_explainer_amd.engine import AdamState, MaskOptimJob
    if which.startswith("six largest"):     # n = 3000 ... 5600: the rows-beyond-4095 variants of the plan kernels, hub rows split over many slots
        wl = bench.Workload("ba100k", 16384)
        graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
        sizes = engine.khop_device(graph, wl.targets, 3).sizes
        targets = wl.targets[np.sort(np.argsort(sizes)[-6:])]
    else:
        wl = bench.Workload("ba100k", 2048)
        graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
        dn_all = engine.khop_device(graph, wl.targets, 3)
        probe = MaskOptimJob.from_csr(graph, dn_all, None, wl.label[wl.targets], wl.ck["sd"])
        large = np.nonzero(probe.route() == 7)[0]
        probe.close()
        assert len(large) >= 12
        order = large[np.argsort(dn_all.sizes[large])]
        pick = np.sort(order[np.linspace(0, len(order) - 1, 12).astype(int)])
        targets = wl.targets[pick]
    dn = engine.khop_device(graph, targets, 3)
    teacher = MaskOptimJob.from_csr(graph, dn, None, wl.label[targets], wl.ck["sd"], analyze=False)
    student = MaskOptimJob.from_csr(graph, dn, None, wl.label[targets], wl.ck["sd"])
    assert set(teacher.route()) == {0} and set(student.route()) == {7}
    raw = engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets)
    teacher.set_masks_raw(raw)
    student.set_masks_raw(raw)
    st, rows = None, []
    for w in range(6):
        M0 = teacher.M.clone()
        teacher.launch(Hyper(num_iters=50, use_resident=False), state=st, keep_state=True)
        want = teacher.fetch_edges(with_mask=True)
        student.M.copy_(M0)
        start = None if st is None else AdamState(st.first_iter, st.m, st.v, st.feat)
        student.launch(Hyper(num_iters=50), state=start, keep_state=True)
        got = student.fetch_edges(with_mask=True)
        st = teacher.state_out
        for k in range(len(targets)):
            e = slice(int(want.eoff[k]), int(want.eoff[k + 1]))
            rows.append((int(targets[k]), w, float(np.abs(got.masked_adj[e] - want.masked_adj[e]).max()), float(np.abs(got.mask_rc[e] - want.mask_rc[e]).max()),
                         float(np.abs(got.feat_mask[k] - want.feat_mask[k]).max())))
    err = np.asarray([[r[2], r[3], r[4]] for r in rows])
    ok = (err[:, 0] <= 1e-5) & (err[:, 2] <= 1e-5)
    worst = rows[int(np.argmax(err[:, 0]))]
    print(f"k_sparse_large vs streaming, sizes {sorted(int(n) for n in dn.sizes)}: {int(ok.sum())} / {len(rows)} windows within 1e-5 "
          f"(masked_adj worst {err[:, 0].max():.2e} at target {worst[0]} window {worst[1]}, mask parameter worst {err[:, 1].max():.2e}, feat {err[:, 2].max():.2e})")
    assert ok.all() and err[:, 0].max() <= SUB_FLAG_BOUND, [r for r, o in zip(rows, ok) if not o]      # (measured: every window, since round 4)
