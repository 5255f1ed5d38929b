"""GPU parity on BASELINE.json's configs AT FULL SIZE, against outputs of the REAL reference.

Fixtures: tests/golden/*_full_explain.npz, config4_explain.npz, ba100k_explain.npz - produced by
tests/golden/make_golden_full.py, which imports /root/reference and runs it under the seed protocol
(torch.manual_seed(1000 + target) before every explanation) for 300 epochs, and for the first 50 epochs of the
same trajectory.

What "parity" can mean here.  The reference's trajectory is not a continuous function of its input: a ReLU gate of the
encoder that crosses zero within fp32 round-off of an iteration boundary switches one iteration earlier or later, and
Adam's scale-free step turns that into a 1e-5 .. 1e-3 shift of the final mask (a chaotic O(1) divergence on Tree-Grid).
Measured on the CPU alone (tests/golden/make_golden_branches.py, make_golden_full.py): a 1-ulp perturbation of the initial
mask moves 41 / 400 syn1, 39 / 360 syn4, 598 / 720 syn5 and 42 / 64 config-4 targets by more than 2e-6 after 300 epochs,
and the closed-form fp32 oracle (same mathematics, other summation order) differs from the reference on the chaotic ones
(`cond_mask`, `cond_feat` in the fixtures).  So "the reference's output" is a small SET per target, and any
other implementation - on CPU or GPU - lands on one member of it.  The rule (helpers.explained_outcome; round 6: no percentage):
  * every CALM target (conditioning over the horizon <= 2e-6, measured on the CPU alone: CPU-vs-CPU deviation + the window probes) lies
    within 1e-5 (masked_adj AND sigmoid(feat_mask)) of the reference's ONE output, or it is on the decision suite's committed list
    (tests/golden/<name>_ties.json: the calm targets tests/test_decision_parity.py found beyond 1e-5 on the GPU, each with the tie of the reference
    at which the engine first leaves its decisions, or with identical decisions and a drift inside the accumulated round-off bound) and within the
    largest branch jump seen on the CPU (5e-3); anything else fails;
  * the same after the first 50 epochs of the same trajectory (the list's windows of the first 50 epochs).
  The distance to the pre-declared alternate outcomes (`*_branches.npz`) is still printed (the "three numbers"), it gates nothing.
Every target of every config - every size, every kernel route - is thereby compared with the reference at 1e-5.
Whole configs run as ONE batched job through the device-side pipeline: k-hop sets, packing, raw-RNG mask upload,
optimisation, edge-list results (gnnx_khop, gnnx_pack_csr, gnnx_scatter_masks, gnnx_run, gnnx_gather_edges)."""
import os

import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob, Subgraph
from gnn_model_explainer_amd.utils import synthetic
from gnn_model_explainer_amd.utils.graph_utils import KHopIndex

pytestmark = pytest.mark.gpu
TOL = 1e-5
WELL = 2e-6          # CPU-vs-CPU deviation up to which a target counts as well conditioned


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x.astype(np.float64)))


def _full(name):
    return np.load(os.path.join(helpers.GOLDEN, name))


def _per_target_err(eoff, got, want):
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return np.asarray([d[a:b].max() if b > a else 0.0 for a, b in zip(eoff[:-1], eoff[1:])])


def _run_node_config(name, iters):
    """Every motif node of a syn dataset as one job, device-side end to end.  -> (fixture, EdgeMasks, routes)"""
    ck, z = helpers.load_ckpt(name), _full(name + "_full_explain.npz")
    targets = z["targets"]
    idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
    graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])
    dn = engine.khop_device(graph, targets, 3)
    # the neighbour lists of the reference's own neighborhoods / extract_neighborhood, bit for bit
    assert np.array_equal(dn.nb_off.cpu().numpy(), z["nb_off"])
    assert np.array_equal(dn.nb_flat.cpu().numpy()[:len(z["nb_flat"])], z["nb_flat"])
    assert np.array_equal(dn.rows, z["node_idx_new"])
    job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"])
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
    job.launch(Hyper(num_iters=iters))
    em = job.fetch_edges()
    assert np.array_equal(em.eoff, z["eoff"])
    return z, em, job.route()


def _check(z, em_vals, em_feat, eoff, horizon, what, min_well, br=None, jump_max=None, list_name=None):
    """helpers.explained_outcome on the distance to the reference's output; the distance to the nearest pre-declared alternate outcome
    (helpers.branch_errors) is printed beside it."""
    early = horizon == "early"
    sfx = "_early" if early else ""
    cm, cf = z["cond_mask" + sfx], z["cond_feat" + sfx]
    err, ferr, matched = helpers.branch_errors(z, br, eoff, em_vals, _sig(em_feat), early)
    well = (cm <= WELL) & (cf <= WELL)
    assert well.sum() >= min_well, f"{what}: only {well.sum()} non-chaotic targets in the fixture"
    s_err, s_ferr, _ = helpers.branch_errors(z, None, eoff, em_vals, _sig(em_feat), early)        # strict: the reference's one output, no alternates
    ids_ = z["targets"] if "targets" in z.files else z["graphs"]
    if list_name is not None:
        calm = helpers.horizon_conditioning(list_name, cm, cf, early) <= WELL
        ok, msg, _ = helpers.explained_outcome(list_name, horizon, ids_, s_err, s_ferr, calm, jump_max)
    else:      # no decision fixture for these targets: every non-chaotic one within the jump bound, nothing more can be asserted about outcomes
        lim = helpers.BRANCH_JUMP_MAX if jump_max is None else jump_max
        worst = float(np.maximum(s_err, s_ferr)[well].max())
        ok, msg = worst <= lim, f"worst non-chaotic target {worst:.2e} (jump limit {lim:.0e})"
    strict = (s_err <= TOL) & (s_ferr <= TOL)
    print(f"{what} [{horizon}] three numbers: strict vs the reference's output {int((strict & well).sum())} / {int(well.sum())} non-chaotic "
          f"({int(strict.sum())} / {len(strict)} of all targets); with the pre-declared alternates {int((well & (err <= TOL) & (ferr <= TOL)).sum())} / {int(well.sum())}; "
          f"ungated chaotic targets {int((~well).sum())}, {int((strict & ~well).sum())} of them within 1e-5 anyway")
    bad = np.nonzero(well & ((err > TOL) | (ferr > TOL)))[0]
    ids = z["targets"] if "targets" in z.files else z["graphs"]
    print(f"{what} [{horizon}]: {msg}; {int((matched[well] >= 0).sum())} on an alternate branch; beyond 1e-5: "
          f"{[(int(ids[k]), float(max(err[k], ferr[k]))) for k in bad]}; chaotic targets: CPU-vs-CPU up to {cm.max():.2e}, "
          f"GPU-vs-reference up to {err[~well].max() if (~well).any() else 0:.2e}")
    dump = os.environ.get("GNNX_DUMP_OUTLIERS")
    if dump:       # measurement aid: which targets to give more perturbation trials (tests/golden/branch_watch.json)
        import json
        rec = json.load(open(dump)) if os.path.exists(dump) else {}
        rec[f"{what}:{horizon}"] = {"targets": [int(ids[k]) for k in bad], "mask_err": [float(err[k]) for k in bad], "feat_err": [float(ferr[k]) for k in bad]}
        json.dump(rec, open(dump, "w"), indent=1)
    assert ok, f"{what} [{horizon}]: {msg}"
    assert np.isfinite(em_vals).all() and em_vals.min() >= 0 and em_vals.max() <= 1
    return err, ferr, well


@pytest.mark.parametrize("name,min_well_full,min_well_early", [("syn1", 380, 390), ("syn4", 335, 345), ("syn5", 150, 530)])
def test_node_configs_every_motif_node_vs_reference(name, min_well_full, min_well_early):
    """BASELINE configs 2 (syn1: 400 targets) and 3 (syn4: 360, syn5: 720): ALL targets, 300 epochs and 50 epochs."""
    br = helpers.load_branches(name)
    z, em, route = _run_node_config(name, 300)
    print(name, "routes:", dict(zip(*np.unique(route, return_counts=True))))
    fails = []
    for horizon, iters, mw in (("full", 300, min_well_full), ("early", int(z["early_epochs"]), min_well_early)):
        if horizon == "early":
            z, em, _ = _run_node_config(name, iters)
        try:
            _check(z, em.masked_adj, em.feat_mask, em.eoff, horizon, name, mw, br, list_name=name)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, "\n".join(fails)


def test_sigmoid_saturation_bound_per_config():
    """SURVEY.md App. B4 / DESIGN.md: the update path uses d(entropy)/dS = -M, so it stays finite where the reference's
    log(1 - sigmoid(M)) would produce NaN (sigmoid(M) == 1.0 in fp32 needs M > 16.6).  The reference's own runs bound
    |M| on every config, so that difference is unreachable on them."""
    for name in ("syn1_full_explain.npz", "syn4_full_explain.npz", "syn5_full_explain.npz", "config4_explain.npz"):
        assert _full(name)["max_abs_mask"].max() < 12.0, name
    z = _full("ba100k_explain.npz")
    assert max(float(z[f"{t}:stats"][1]) for t in z["targets"]) < 12.0


def test_config4_graph_mode_512_of_4337_graphs_vs_reference():
    """BASELINE config 4: graph-level explanation; 512 size-stratified graphs of the 4337-graph job (tests/golden/config4_windows.npz: the
    LIVE reference's outputs with GcnEncoderGraph weights from the fixture), 300 epochs, on the edge-sparse graph-mode kernel AND - for every
    eighth graph - on the dense streaming kernels, so that a defect of one route cannot hide behind max-pool ties.
    Every full-horizon miss on a graph the two CPU implementations agree on must be EXPLAINED by the fixture's CPU-only window analysis: a
    50-epoch window of that graph in which a max-pool margin / ReLU gate comes within fp32 round-off of its boundary (or a 1-ulp perturbation /
    the other CPU implementation already diverges) - tests/golden/make_golden_windows.py; the windowed test (test_windowed_parity.py) shows
    the kernels follow the reference through those windows at 10-epoch granularity."""
    W = helpers.Windows("config4")
    z = W.z
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    gids = z["graphs"]
    A, X, nn, y = synthetic.molecule_like_graphs(int(gids.max()) + 1, seed=0)
    subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(g, A.shape[1]).numpy()) for g in gids]
    well = (z["cond_mask"] <= WELL) & (z["cond_feat"] <= WELL)
    assert well.sum() >= 180, well.sum()

    def run(sel, analyze, use_resident):
        job = MaskOptimJob([subs[k] for k in sel], sd, graph_mode=True, analyze=analyze)
        job.set_masks([subs[k].mask0 for k in sel])
        job.launch(Hyper(num_iters=int(z["full_epochs"]), use_resident=use_resident))
        em = job.fetch_edges()
        assert np.array_equal(np.diff(em.eoff), np.diff(z["eoff"])[sel])
        want = np.concatenate([z["vals"][z["eoff"][k]:z["eoff"][k + 1]] for k in sel])
        err = _per_target_err(em.eoff, em.masked_adj, want)
        ferr = np.abs(_sig(em.feat_mask) - z["feat_sig"][sel]).max(1)
        return np.maximum(err, ferr), job.route()

    allk = np.arange(len(gids))
    e_sparse, route = run(allk, True, True)
    assert set(route) <= {5, 6}, set(route)                      # the graph-mode classes of the sparse resident kernel
    sub8 = allk[::8]
    e_dense, route_d = run(sub8, False, False)
    # No percentage rule (VERDICT r4 #4b): the full-horizon GATE of graph mode is decision-based (test_decision_parity.py); what this test
    # asserts about OUTCOMES is that every miss on a graph two CPU implementations agree on has a window the CPU-only analysis flags - for the
    # edge-sparse route and, independently, for the dense streaming route on every eighth graph.
    inside = int((e_sparse[well] <= TOL).sum())
    miss = np.nonzero(well & (e_sparse > TOL))[0]
    unexplained = [int(gids[k]) for k in miss if not W.flagged[k].any()]
    wd = well[sub8]
    miss_d = [int(k) for k in sub8[wd & (e_dense > TOL)]]
    unexplained_d = [int(gids[k]) for k in miss_d if not W.flagged[k].any()]
    print(f"config4 [full, 512 graphs]: {inside} / {int(well.sum())} non-chaotic graphs within 1e-5 of the reference's output, worst {float(e_sparse[well].max()):.2e}; "
          f"misses {len(miss)}, of which {len(miss) - len(unexplained)} have a flagged window on the CPU, unexplained: "
          f"{unexplained}; dense streaming route on {len(sub8)} graphs: {int((e_dense[wd] <= TOL).sum())} / {int(wd.sum())} non-chaotic within 1e-5 "
          f"(sparse route on the same graphs: {int((e_sparse[sub8][wd] <= TOL).sum())}), its misses without a flagged window: {unexplained_d}")
    assert not unexplained, unexplained
    assert not unexplained_d, unexplained_d


def test_config4_64_graphs_against_the_reference_outcome_sets():
    """The 64 graphs of config4_explain.npz with the pre-declared alternate outcomes of the one sampler (make_golden_branches.py --what
    config4: 24 one-ulp trials per graph; tests/golden/config4_branches.npz - 41 of the 64 graphs move by more than 2e-6 under a 1-ulp
    perturbation of the initial mask, 37 by more than 1e-5, up to 6e-2: the max-pool ties of symmetric atoms).  Three numbers as for the
    node configs: strict / with alternates / ungated; the gate proper of config 4 is tests/test_decision_parity.py (every decision of
    every epoch against the live reference's), this one shows what the outcome-set argument alone buys in graph mode."""
    z = _full("config4_explain.npz")
    br = helpers.load_branches("config4")
    assert br is not None and len(br["alt_target"]) > 0
    sd = {k[2:]: z[k] for k in z.files if k.startswith("w:")}
    gids = z["graphs"]
    A, X, nn, y = synthetic.molecule_like_graphs(int(gids.max()) + 1, seed=0)
    subs = [Subgraph(A[g], X[g], int(y[g]), 0, None, helpers.seeded_mask0(g, A.shape[1]).numpy()) for g in gids]
    job = MaskOptimJob(subs, sd, graph_mode=True)
    job.set_masks([s.mask0 for s in subs])
    job.launch(Hyper(num_iters=int(z["epochs"])))
    em = job.fetch_edges()
    assert np.array_equal(em.eoff, z["eoff"])
    # (measured in round 4: 23 / 26 non-chaotic graphs strictly within 1e-5, the same 23 with the 208 alternates - in graph mode a flipped
    #  pool tie leads somewhere else every time, 24 trials do not enumerate the outcomes; hence the decision-based gate.  These 64 graphs have no
    #  decision fixture (4 of them are among the 512): the three numbers are printed, every non-chaotic graph must stay within the pool-tie jump.)
    _check(z, em.masked_adj, em.feat_mask, em.eoff, "full", "config4 (64 graphs)", 20, br, jump_max=helpers.CONFIG4_WINDOW_JUMP)


def test_config5_ba100k_route_stratified_targets_vs_reference():
    """BASELINE config 5 (BA-House x100k, 99 997 nodes): real targets from n = 6 to n > 4095, hubs of up to 749
    neighbours, one per kernel route - 64- / 256- / 512-thread sparse resident classes, k_sparse_large (n > 2000 and
    n > 4095, hub rows split over 64-entry slots) and, for the largest one, also the dense streaming kernels - against the
    reference's ExplainModule."""
    z = _full("ba100k_explain.npz")
    ck = helpers.load_ckpt("syn1")
    N, edges, label = synthetic.ba_house(42857, 11428, seed=0)
    csr = synthetic.csr_from_edges(N, edges)
    feat = np.ones((N, 10), np.float32)
    pred = synthetic.sparse_gcn_predict(csr, feat, ck["sd"])
    targets = z["targets"]
    graph = engine.device_graph(csr, feat, pred)
    dn = engine.khop_device(graph, targets, 3)
    for t, nb, row in zip(targets, dn.lists(), dn.rows):         # same graph, same sets as the fixture's sparse BFS
        assert np.array_equal(nb, z[f"{t}:neighbors"]) and row == z[f"{t}:meta"][0]
        assert np.array_equal(np.argmax(pred[nb], 1), z[f"{t}:pred_label"])
    job = MaskOptimJob.from_csr(graph, dn, None, label[targets], ck["sd"])
    route = job.route()
    print("ba100k routes:", {int(t): (int(n), int(r)) for t, n, r in zip(targets, dn.sizes, route)})
    assert (7 in route) and (6 in route) and ((8 in route) or (4 in route)) and 0 not in route, route
    assert all(r == 7 for r, nn in zip(route, dn.sizes) if nn > 4095) and max(dn.sizes) > 4095   # k_sparse_large takes them since round 2
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
    job.launch(Hyper(num_iters=int(z["epochs"])))
    em = job.fetch_edges()
    for k, t in enumerate(targets):
        a, b = em.eoff[k], em.eoff[k + 1]
        assert np.array_equal(em.rc[a:b], z[f"{t}:edges"].astype(np.int32)), t
        err = np.abs(em.masked_adj[a:b] - z[f"{t}:vals"]).max()
        ferr = np.abs(_sig(em.feat_mask[k]) - z[f"{t}:feat_sig"]).max()
        print(f"  target {t}: n={dn.sizes[k]} route={route[k]} err={err:.2e} feat={ferr:.2e}")
        assert err <= TOL and ferr <= TOL, (t, dn.sizes[k], route[k], err, ferr)
    # the largest target once more on the dense streaming kernels (what loss logging, mask_act = ReLU and --bn run on)
    k = int(np.argmax(dn.sizes))
    t = int(targets[k])
    nb = [dn.lists()[k]]
    job = MaskOptimJob.from_csr(graph, nb, [dn.rows[k]], label[[t]], ck["sd"], analyze=False)
    assert list(job.route()) == [0]
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes[k:k + 1], seeds=1000 + targets[k:k + 1]))
    job.launch(Hyper(num_iters=int(z["epochs"])))
    em = job.fetch_edges()
    err = np.abs(em.masked_adj - z[f"{t}:vals"]).max()
    ferr = np.abs(_sig(em.feat_mask[0]) - z[f"{t}:feat_sig"]).max()
    print(f"  target {t} on the streaming kernels: n={dn.sizes[k]} err={err:.2e} feat={ferr:.2e}")
    assert err <= TOL and ferr <= TOL


def test_edge_list_results_equal_dense_results():
    """gnnx_gather_edges returns exactly the non-zero entries of the dense result (and the final mask parameters)."""
    ck, gx = helpers.load_ckpt("syn1"), helpers.load_explain("syn1")
    subs = []
    for t in (302, 555, 309, 300):
        nb = gx[f"{t}:neighbors"]
        A, X, lab, yhat = helpers.subgraph(ck, nb)
        new = int(gx[f"{t}:node_idx_new"])
        subs.append(Subgraph(A, X, int(lab[new]), new, yhat, helpers.seeded_mask0(t, len(nb)).numpy()))
    job = MaskOptimJob(subs, ck["sd"])
    hy = Hyper(num_iters=10)
    dense = job.run([s.mask0 for s in subs], hy)
    em = job.fetch_edges(with_mask=True)
    for k, s in enumerate(subs):
        assert np.array_equal(em.dense(k, np.float32), dense.masked_adj[k])
        a, b = em.eoff[k], em.eoff[k + 1]
        r, c = em.rc[a:b, 0], em.rc[a:b, 1]
        rr, cc = np.nonzero(np.triu(s.adj, 1))
        assert np.array_equal(r, rr) and np.array_equal(c, cc)
        assert np.array_equal(em.mask_rc[a:b, 0], dense.mask[k][r, c]) and np.array_equal(em.mask_rc[a:b, 1], dense.mask[k][c, r])
