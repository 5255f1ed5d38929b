"""The options of the reference's CLI that round 3 moved onto the engine - `--opt sgd | rmsprop | adagrad`, `--opt-scheduler step | cos`
(utils/train_utils.py:7-22) and `unconstrained=True` (explain.py:688-691) - against outputs of the LIVE reference
(tests/golden/options_explain.npz, tests/golden/make_golden_options.py), on the emulator and on the GPU, and the reference's own
end-use number: the ROC-AUC explain_nodes_gnn_stats writes (explain.py:295-353)."""
import argparse
import os

import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, Subgraph
from gnn_model_explainer_amd.explainer import explain
from test_emu_kernels import _Backend, _node_case

TOL = 1e-5
Z = np.load(os.path.join(helpers.GOLDEN, "options_explain.npz"))


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def be(request):
    return _Backend(request.param)


def _args(**kw):
    a = argparse.Namespace(lr=0.1, opt="adam", opt_scheduler="none", num_epochs=100, opt_decay_step=30, opt_decay_rate=0.5, opt_restart=100)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _check(be, t, key, hy, unconstrained=False, analyze=True):
    ck, gx, sg = _node_case("syn1", t)
    true_adj = sg.adj
    if unconstrained:
        sg = Subgraph(1.0 - np.eye(len(true_adj), dtype=np.float32), sg.feat, sg.gt_label, sg.target_row, sg.pred_label, sg.mask0)
    job = be.job([sg], ck["sd"], analyze=analyze)
    if unconstrained:          # the route Explainer.explain(..., unconstrained=True) takes: complete graph, features unmasked
        job.set_masks([sg.mask0])
        job.launch(hy, state=explain._unmasked_features_state(job))
        res = job.fetch(hy)
        res.feat_mask[:] = explain._regulariser_only_feat_mask(_args(), job.D, hy.num_iters)
    else:
        res = job.run([sg.mask0], hy)
    r, c = np.nonzero(np.triu(true_adj, 1))
    got = (res.masked_adj[0].astype(np.float64) * true_adj)[r, c]
    em = np.abs(got - Z[key + ":masked_adj_edges"]).max()
    ef = np.abs(1 / (1 + np.exp(-res.feat_mask[0].astype(np.float64))) - Z[key + ":feat_sig"]).max()
    tol = TOL
    if unconstrained:
        # conditioning of this run, measured on the CPU alone: the closed-form oracle on the same complete graph vs the reference's output;
        # beyond 2e-6 the run is one of those that amplify round-off (bounded by the branch jump, as in helpers.explained_outcome)
        from oracle import closed_form
        o = closed_form.ClosedFormOracle(sg.adj, sg.feat, ck["sd"], sg.gt_label, sg.pred_label, sg.target_row, sg.mask0)
        o.f[:] = 40.0
        cond = np.abs((o.run(hy.num_iters) * true_adj)[r, c] - Z[key + ":masked_adj_edges"]).max()
        tol = TOL if cond <= helpers.WELL else helpers.BRANCH_JUMP_MAX
        print(f"{key}: closed form vs reference {cond:.2e} -> tolerance {tol:g}")
    print(f"{key} route {job.route()}: mask {em:.2e} feat {ef:.2e}")
    assert em <= tol and ef <= TOL, (key, em, ef)


@pytest.mark.parametrize("t", [302, 309])
@pytest.mark.parametrize("opt", ["sgd", "rmsprop", "adagrad"])
def test_other_optimizers_vs_reference(be, t, opt):
    hy = explain._hyper(_args(opt=opt))
    assert hy.opt == opt and hy.eps == explain.OPTIMIZER_EPS[opt]
    _check(be, t, f"opt:{opt}:{t}", hy)
    if t == 309:        # once more on the dense streaming kernels (what --bn, logging and large targets run on)
        hy.use_resident = False
        _check(be, t, f"opt:{opt}:{t}", hy, analyze=False)


@pytest.mark.parametrize("t", [302, 309])
@pytest.mark.parametrize("sched", ["step", "cos"])
def test_lr_schedulers_vs_reference(be, t, sched):
    a = _args(opt_scheduler=sched)
    hy = explain._hyper(a)
    assert np.array_equal(hy.lr_schedule, Z[f"sched:{sched}:{t}:lr"])      # torch's own scheduler: the reference's doubles, bit for bit
    _check(be, t, f"sched:{sched}:{t}", hy)
    if t == 309:
        hy = explain._hyper(a, use_resident=False)
        _check(be, t, f"sched:{sched}:{t}", hy, analyze=False)


@pytest.mark.parametrize("t", [302, 309])
def test_unconstrained_vs_reference(be, t):
    _check(be, t, f"unc:{t}", explain._hyper(_args()), unconstrained=True)


# ---------------------------------------------------------------- method="att" on k_att (csrc/gnnx_att.hpp) ----------------------------------------------------------------
def _att_case(t):
    """syn1 target t with the attention encoder the reference's train.py produced (tests/golden/make_golden_options.py)."""
    ck = helpers.load_ckpt("syn1")
    sd = {k[len("route:att:w:"):]: Z[k] for k in Z.files if k.startswith("route:att:w:")}
    nb = Z[f"route:att:{t}:neighbors"]
    A = ck["adj"][np.ix_(nb, nb)].astype(np.float32)
    X = ck["feat"][nb].astype(np.float32)
    new = int(np.searchsorted(nb, t))
    yhat = np.argmax(Z["route:att:pred"][nb], 1)
    m0 = helpers.seeded_mask0(t, len(nb)).numpy()
    return ck, sd, Subgraph(A, X, int(ck["label"][t]), new, yhat, m0)


@pytest.mark.parametrize("t", [302, 309])
def test_method_att_kernel_vs_reference(be, t):
    """method="att" (models.py:62-68) - adj * (x W_att)(x W_att)^T in every layer, forward and backward through the attention
    products - on k_att against the LIVE reference's output for an encoder its own train.py trained with --method att
    (100 epochs, as the fixture was made), at the 1e-5 of every other parity test (measured: 6e-8 on the GPU)."""
    if be.name == "emu" and t == 309:
        pytest.skip("n = 48 x 100 epochs takes the emulator 80 s: the GPU twin runs it; the emulator covers n = 6, the graph-mode fixture and the autograd check on n = 48")
    ck, sd, sg = _att_case(t)
    job = be.job([sg], sd)
    assert job.att is not None
    res = job.run([sg.mask0], explain._hyper(_args()))
    r, c = np.nonzero(np.triu(sg.adj, 1))
    em = np.abs(res.masked_adj[0][r, c].astype(np.float64) - Z[f"route:att:{t}:masked_adj_edges"]).max()
    ef = np.abs(1 / (1 + np.exp(-res.feat_mask[0].astype(np.float64))) - Z[f"route:att:{t}:feat_sig"]).max()
    print(f"method=att target {t}: n={len(sg.adj)} mask {em:.2e} feat {ef:.2e}")
    assert em <= TOL and ef <= TOL
    ma = res.masked_adj[0]
    assert np.array_equal(ma, ma.T) and np.all(ma[sg.adj == 0] == 0)


@pytest.mark.gpu
def test_method_att_kernel_at_scale_vs_reference():
    """k_att on the reference CLI's default node list range(400, 700, 5) and on syn1's largest neighbourhood (n = 310: the hub rows), 300
    epochs, one batched job, against the LIVE reference's explanations of the attention encoder its own train.py trained
    (tests/golden/make_golden_att_scale.py -> att_scale.npz).  Two horizons: the Adam state after the first 50 epochs (mask entries of both
    directions -> masked adjacency, sigmoid(feat_mask)) and the returned masks after 300.  Rule per target, as everywhere in round 4: within
    max(1e-5, 50 c) of the reference, c = the reference's own 1-ulp sensitivity at that horizon (measured on the reference itself: there is no
    closed-form oracle for this encoder), the 300-epoch horizon counted as six windows; every target beyond 1e-5 is listed."""
    z = np.load(os.path.join(helpers.GOLDEN, "att_scale.npz"))
    ck = helpers.load_ckpt("syn1")
    sd = {k[len("route:att:w:"):]: Z[k] for k in Z.files if k.startswith("route:att:w:")}
    subs = []
    for k, t in enumerate(z["targets"]):
        nb = z["nb_flat"][z["nb_off"][k]:z["nb_off"][k + 1]].astype(np.int64)
        A = ck["adj"][np.ix_(nb, nb)].astype(np.float32)
        subs.append(Subgraph(A, ck["feat"][nb].astype(np.float32), int(ck["label"][t]), int(z["node_idx_new"][k]), np.argmax(Z["route:att:pred"][nb], 1),
                             helpers.seeded_mask0(int(t), len(nb)).numpy()))
    job = engine.MaskOptimJob(subs, sd)
    assert job.att is not None
    args = _args()
    eoff = z["eoff"]
    per_target = lambda d: np.asarray([d[a:b].max() if b > a else 0.0 for a, b in zip(eoff[:-1], eoff[1:])])
    # (a) the first 50 epochs: the optimiser state itself
    args.num_epochs = int(z["early"])
    job.set_masks([s.mask0 for s in subs])
    job.launch(explain._hyper(args))
    st = job.fetch_edges(with_mask=True)
    assert np.array_equal(np.diff(st.eoff), np.diff(eoff))
    e50 = np.maximum(per_target(np.abs(helpers.abar_from_mask_rc(st.mask_rc) - helpers.abar_from_mask_rc(z["M50"]))),
                     np.abs(helpers._sig64(st.feat_mask) - helpers._sig64(z["f50"])).max(1))
    # (b) the full horizon: the returned masks
    args.num_epochs = int(z["epochs"])
    job.set_masks([s.mask0 for s in subs])
    job.launch(explain._hyper(args))
    em = job.fetch_edges()
    e300 = np.maximum(per_target(np.abs(em.masked_adj.astype(np.float64) - z["vals"])),
                      np.abs(helpers._sig64(em.feat_mask) - z["feat_sig"].astype(np.float64)).max(1))
    b50 = np.maximum(TOL, 50.0 * z["sens50"])
    b300 = 6.0 * np.maximum(TOL, 50.0 * z["sens300"])
    n = np.diff(z["nb_off"])
    print(f"method=att at scale: {len(subs)} targets (n = {n.min()} ... {n.max()}); after 50 epochs {int((e50 <= TOL).sum())} within 1e-5 of the reference's Adam state "
          f"(worst {e50.max():.2e}), after 300 epochs {int((e300 <= TOL).sum())} within 1e-5 of its output (worst {e300.max():.2e}); beyond 1e-5 (target, n, error, "
          f"the reference's own 1-ulp sensitivity): 50 epochs {[(int(t), int(nn), float('%.1e' % e), float('%.1e' % c)) for t, nn, e, c in zip(z['targets'], n, e50, z['sens50']) if e > TOL]}, "
          f"300 epochs {[(int(t), int(nn), float('%.1e' % e), float('%.1e' % c)) for t, nn, e, c in zip(z['targets'], n, e300, z['sens300']) if e > TOL]}")
    assert (e50 <= b50).all() and (e300 <= b300).all(), (e50.max(), e300.max())      # (the rule: every target within its own bound - no share)


ZG = np.load(os.path.join(helpers.GOLDEN, "attgraph_explain.npz"))


def test_method_att_graph_mode_kernel_vs_reference(be):
    """--method att in GRAPH mode (GcnEncoderGraph: per-layer max-pool over all rows, the direct gradient on the arg-max rows) on
    k_att against the LIVE reference (tests/golden/make_golden_att_graph.py: its GcnEncoderGraph with args.method = "att", four padded
    molecule-like graphs of 10-39 nodes, 60 epochs), all four graphs as one batch."""
    sd = {k[2:]: ZG[k] for k in ZG.files if k.startswith("w:")}
    graphs = [2] if be.name == "emu" else list(range(len(ZG["label"])))     # (the emulator steps one padded 40-row graph for 20 s)
    subs = [Subgraph(ZG["adj"][g], ZG["feat"][g], int(ZG["label"][g]), 0, None, helpers.seeded_mask0(g, ZG["adj"][g].shape[0]).numpy()) for g in graphs]
    job = be.job(subs, sd, graph_mode=True)
    assert job.att is not None
    a = argparse.Namespace(lr=0.1, opt="adam", opt_scheduler="none", num_epochs=int(ZG["epochs"]))
    res = job.run([s.mask0 for s in subs], explain._hyper(a))
    for k, g in enumerate(graphs):
        e = ZG["adj"][g] != 0
        em = np.abs(res.masked_adj[k].astype(np.float64) * ZG["adj"][g] - ZG[f"{g}:masked_adj"])[e].max()
        ef = np.abs(1 / (1 + np.exp(-res.feat_mask[k].astype(np.float64))) - ZG[f"{g}:feat_mask_sigmoid"]).max()
        print(f"method=att graph {g}: nodes {int(ZG['num_nodes'][g])} mask {em:.2e} feat {ef:.2e}")
        assert em <= TOL and ef <= TOL
        assert np.all(res.masked_adj[k][~e] == 0)


@pytest.mark.gpu
def test_method_att_graph_mode_through_the_explainer_api_on_gpu(tmp_path):
    """Explainer.explain(graph_idx=g, graph_mode=True) with an attention GcnEncoderGraph: the kernels, no PyTorch-ROCm route."""
    import warnings
    from gnn_model_explainer_amd import models
    from test_explainer_api import _args as api_args
    args = api_args(tmp_path, int(ZG["epochs"]), "syn1", method="att")
    args.bmname, args.graph_mode = "Mutagenicity", True
    model = models.GcnEncoderGraph(14, 20, 20, 2, 3, bn=False, args=args)
    model.load_state_dict({k[2:]: torch.tensor(ZG[k]) for k in ZG.files if k.startswith("w:")})
    ex = explain.Explainer(model, ZG["adj"], ZG["feat"], ZG["label"], ZG["pred"][None], None, args, writer=None, print_training=False,
                           graph_mode=True, graph_idx=0)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)          # the PyTorch-ROCm route would announce itself
        for g in (0, 3):
            torch.manual_seed(1000 + g)
            ma = ex.explain(node_idx=0, graph_idx=g, graph_mode=True)
            e = ZG["adj"][g] != 0
            assert np.abs(ma - ZG[f"{g}:masked_adj"])[e].max() <= TOL


def test_method_att_plans_refuse_the_base_encoder_entry_points(be, tmp_path):
    """gnnx_forward / gnnx_grad_baseline run the base encoder: on a plan with attention weights they fail instead of silently ignoring them;
    the mirror's gradient baseline says so before it gets there."""
    ck, sd, sg = _att_case(302)
    job = be.job([sg], sd)
    with pytest.raises(RuntimeError, match="method=att"):
        job.grad_baseline()
    with pytest.raises(RuntimeError, match="method=att"):
        job.forward([sg.mask0])
    if be.name == "emu":
        ck, ex = _route_explainer(tmp_path, "att", method="att")
        with pytest.raises(NotImplementedError, match="attention encoder"):
            ex.explain(302, model="grad")


def test_method_att_one_step_equals_autograd(be):
    """One Adam step of k_att against torch autograd through the mirror encoder (models.GcnEncoderNode with method="att") on the
    same inputs: the updated mask entries on the edges, the feature mask, and two targets in one batch == alone."""
    from gnn_model_explainer_amd import models
    from gnn_model_explainer_amd.explainer import torch_route
    ck, sd, sg = _att_case(309)
    _, _, sg2 = _att_case(302)
    args = argparse.Namespace(method="att", bias=True, num_gc_layers=3, mask_act="sigmoid", num_epochs=2, lr=0.1, opt="adam", opt_scheduler="none")
    model = models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args)
    model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    hy = explain._hyper(args)
    res = be.job([sg, sg2], sd).run([sg.mask0, sg2.mask0], hy)
    alone = be.job([sg], sd).run([sg.mask0], hy)
    assert np.array_equal(res.masked_adj[0], alone.masked_adj[0]) and np.array_equal(res.mask[0], alone.mask[0])
    for k, s in enumerate((sg, sg2)):
        mod = torch_route.TorchExplainModule(torch.tensor(s.adj[None]), torch.tensor(s.feat[None]), model, torch.tensor([[0] * s.target_row + [s.gt_label]]),
                                             args, explain.COEFFS, s.mask0, device="cpu")
        opt = torch.optim.Adam([mod.mask, mod.feat_mask], lr=0.1)
        model.eval()
        for _ in range(2):
            opt.zero_grad()
            pred, _ = mod(s.target_row)
            mod.loss(pred, s.pred_label, s.target_row).backward()
            opt.step()
        e = s.adj != 0
        assert np.abs(res.masked_adj[k] - mod.masked_adj[0].detach().numpy())[e].max() < 2e-6     # the second forward
        assert np.abs(res.mask[k] - mod.mask.detach().numpy())[e].max() < 2e-5                     # two steps of 0.1
        assert np.abs(res.feat_mask[k][:10] - mod.feat_mask.detach().numpy()).max() < 2e-5


def test_method_att_hub_rows_and_hop_pruning_equal_autograd(be):
    """k_att's round-5 structure against torch autograd through the mirror encoder on a graph built for it: a hub of 70 entries (three chunks of
    32: its partial sums go through the second pass) two hops from the target, its other leaves three hops away (rows the pruned phases skip:
    they only feed layer 1's gathers), a chain that reaches three hops on the other side, a row without entries beyond the diagonal.  Three Adam
    steps: the third forward has consumed gradients that passed every pruned phase twice."""
    from gnn_model_explainer_amd import models
    from gnn_model_explainer_amd.explainer import torch_route
    _, sd, _ = _att_case(302)
    rng = np.random.default_rng(5)
    n = 78
    A = np.zeros((n, n), np.float32)
    def link(i, j):
        A[i, j] = A[j, i] = 1.0
    t, hub = 0, 1
    for leaf in range(2, 72):
        link(hub, leaf)            # rows 2 .. 71: the hub's leaves
    link(t, 2)                     # t - leaf 2 - hub: the hub two hops away, the other leaves three
    link(t, 72); link(72, 73); link(73, 74)      # a chain: one, two, three hops
    link(72, 75); link(75, 76)     # (row 77 stays isolated: the reference's neighbourhoods never hold such a row, the kernel must not trip on it)
    link(3, 4); link(5, 6)         # entries between rows three hops away
    X = rng.normal(size=(n, 10)).astype(np.float32)
    yhat = rng.integers(0, 4, n)
    sg = Subgraph(A, X, 1, t, yhat, helpers.seeded_mask0(7, n).numpy())
    args = argparse.Namespace(method="att", bias=True, num_gc_layers=3, mask_act="sigmoid", num_epochs=3, lr=0.1, opt="adam", opt_scheduler="none")
    model = models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args)
    model.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    res = be.job([sg], sd).run([sg.mask0], explain._hyper(args))
    mod = torch_route.TorchExplainModule(torch.tensor(sg.adj[None]), torch.tensor(sg.feat[None]), model, torch.tensor([[0] * sg.target_row + [sg.gt_label]]),
                                         args, explain.COEFFS, sg.mask0, device="cpu")
    opt = torch.optim.Adam([mod.mask, mod.feat_mask], lr=0.1)
    model.eval()
    for _ in range(3):
        opt.zero_grad()
        pred, _ = mod(sg.target_row)
        mod.loss(pred, sg.pred_label, sg.target_row).backward()
        opt.step()
    e = sg.adj != 0
    em = np.abs(res.masked_adj[0] - mod.masked_adj[0].detach().numpy())[e].max()
    ep = np.abs(res.mask[0] - mod.mask.detach().numpy())[e].max()
    ef = np.abs(res.feat_mask[0][:10] - mod.feat_mask.detach().numpy()).max()
    print(f"method=att hub graph: masked adjacency {em:.2e}, mask parameter {ep:.2e}, feature mask {ef:.2e}")
    assert em < 2e-6 and ep < 3e-5 and ef < 3e-5


# ---------------------------------------------------------------- the PyTorch-ROCm route (explainer/torch_route.py) ----------------------------------------------------------------
def _route_explainer(tmp, tag, **kw):
    from gnn_model_explainer_amd import models
    from test_explainer_api import _args as api_args
    ck = helpers.load_ckpt("syn1")
    args = api_args(tmp, 100, "syn1", **kw)
    sd = {k[len(f"route:{tag}:w:"):]: torch.tensor(Z[k]) for k in Z.files if k.startswith(f"route:{tag}:w:")}
    model = models.GcnEncoderNode(10, 20, 20, 4, args.num_gc_layers, bn=False, args=args)
    model.load_state_dict(sd)                                   # the reference's state_dict keys (att_weight, conv_block.1.*) load as they are
    pred = Z[f"route:{tag}:pred"]
    ex = explain.Explainer(model, ck["adj"][None].astype(np.float64), ck["feat"][None].astype(np.float64), ck["label"][None], pred[None], None,
                           args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    return ck, ex


def _route_case(tmp_path, tag, kw, on_kernels=False, targets=(302, 309)):
    ck, ex = _route_explainer(tmp_path, tag, **kw)
    assert (explain._torch_route_reason(ex.args, ex.model) is None) == on_kernels
    for t in targets:
        torch.manual_seed(1000 + t)
        with pytest.warns(RuntimeWarning, match="PyTorch-ROCm route") if not (on_kernels or _already_warned(ex)) else _nullcontext():
            ma = ex.explain(t)
        nb = Z[f"route:{tag}:{t}:neighbors"]
        assert ma.dtype == np.float64 and ma.shape == (len(nb), len(nb))
        r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
        em = np.abs(ma[r, c] - Z[f"route:{tag}:{t}:masked_adj_edges"]).max()
        lr = ex.last_result
        fsig = lr.feat_mask_sigmoid[0] if hasattr(lr, "feat_mask_sigmoid") else 1 / (1 + np.exp(-np.asarray(lr.feat_mask, np.float64)[0][:10]))
        ef = np.abs(fsig - Z[f"route:{tag}:{t}:feat_sig"]).max()
        print(f"route {tag} target {t}: n={len(nb)} mask {em:.2e} feat {ef:.2e}")
        assert em <= 1e-4 and ef <= 1e-4          # torch ops in another order on another device; the reference run here is CPU


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _already_warned(ex):
    from gnn_model_explainer_amd.explainer import torch_route
    return explain._torch_route_reason(ex.args, ex.model) in torch_route._warned


@pytest.mark.parametrize("tag,kw", [("att+bn", dict(method="att")), ("l4", dict(num_gc_layers=4))])
def test_torch_route_vs_reference_on_the_cpu_hook(tmp_path, monkeypatch, tag, kw):
    """A 4-layer encoder (trained by the reference's train.py) through the drop-in API: the mirror models load the reference's
    state_dict, the explanation runs on explainer/torch_route.py (here on the CPU through the test hook) and matches the
    reference's own output.  method="att" runs on k_att; what k_att does not implement (loss logging here: print_training) still
    takes this route and gives the same numbers."""
    monkeypatch.setitem(explain._ENGINE, "device", "cpu")
    if tag == "att+bn":
        ck, ex = _route_explainer(tmp_path, "att", **kw)
        ex.print_training = True
        assert "loss logging" in explain._torch_route_reason(ex.args, ex.model, record_loss=True)
        torch.manual_seed(1000 + 309)
        with pytest.warns(RuntimeWarning, match="PyTorch-ROCm route"):
            ma = ex.explain(309)
        nb = Z["route:att:309:neighbors"]
        r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
        assert np.abs(ma[r, c] - Z["route:att:309:masked_adj_edges"]).max() <= 1e-4
        return
    _route_case(tmp_path, tag, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("l4", dict(num_gc_layers=4))])
def test_torch_route_vs_reference_on_gpu(tmp_path, tag, kw):
    _route_case(tmp_path, tag, kw)


def test_method_att_through_the_explainer_api_on_the_emulator(tmp_path, monkeypatch):
    """`--method att` through Explainer.explain: the kernels (k_att), not the PyTorch-ROCm route - the emulator build of the
    same sources stands in for the GPU here."""
    from emu.emu_engine import emu_library
    monkeypatch.setitem(explain._ENGINE, "lib", emu_library())
    monkeypatch.setitem(explain._ENGINE, "device", "cpu")
    _route_case(tmp_path, "att", dict(method="att"), on_kernels=True, targets=(302,))   # (the emulator steps n = 48 for a minute: 309 is test_method_att_kernel_vs_reference's)


@pytest.mark.gpu
def test_method_att_through_the_explainer_api_on_gpu(tmp_path):
    _route_case(tmp_path, "att", dict(method="att"), on_kernels=True)


def test_torch_route_refuses_a_cpu_only_host(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ck, ex = _route_explainer(tmp_path, "l4", num_gc_layers=4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ex.explain(302)


def test_unconstrained_runs_with_logging_or_relu_take_the_torch_route():
    """the kernels run unconstrained=True with the engine's feature mask pinned at sigma = 1 (the reference's regulariser-only
    feature mask is recovered on the host), so a loss LOGGED by them would carry feat_size = 1 where the reference logs mean(sigmoid(f));
    and the reference's unconstrained forward always applies the sigmoid (explain.py:689), whatever mask_act says.  Both combinations go to
    the PyTorch-ROCm route instead of differing silently; the plain unconstrained run stays on the kernels."""
    ck = helpers.load_ckpt("syn1")
    sd = {k: torch.tensor(v) for k, v in ck["sd"].items()}
    a = _args()
    a.method, a.mask_act = "base", "sigmoid"

    class _M:          # (only its modules() are looked at, for dropout)
        def modules(self):
            return []
    assert explain._torch_route_reason(a, _M(), state_dict=sd, unconstrained=True) is None
    assert "unconstrained" in explain._torch_route_reason(a, _M(), state_dict=sd, unconstrained=True, record_loss=True)
    a.mask_act = "ReLU"
    assert "unconstrained" in explain._torch_route_reason(a, _M(), state_dict=sd, unconstrained=True)
    assert explain._torch_route_reason(a, _M(), state_dict=sd) is None
