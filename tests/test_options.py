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
        # beyond 2e-6 the run is one of those that amplify round-off (the rule of helpers.parity_verdict: bounded by the branch jump)
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


def _route_case(tmp_path, tag, kw):
    ck, ex = _route_explainer(tmp_path, tag, **kw)
    assert explain._torch_route_reason(ex.args, ex.model) is not None
    for t in (302, 309):
        torch.manual_seed(1000 + t)
        with pytest.warns(RuntimeWarning, match="PyTorch-ROCm route") if not _already_warned(ex) else _nullcontext():
            ma = ex.explain(t)
        nb = Z[f"route:{tag}:{t}:neighbors"]
        assert ma.dtype == np.float64 and ma.shape == (len(nb), len(nb))
        r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
        em = np.abs(ma[r, c] - Z[f"route:{tag}:{t}:masked_adj_edges"]).max()
        ef = np.abs(ex.last_result.feat_mask_sigmoid[0] - Z[f"route:{tag}:{t}:feat_sig"]).max()
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


@pytest.mark.parametrize("tag,kw", [("att", dict(method="att")), ("l4", dict(num_gc_layers=4))])
def test_torch_route_vs_reference_on_the_cpu_hook(tmp_path, monkeypatch, tag, kw):
    """method="att" / a 4-layer encoder (trained by the reference's train.py) through the drop-in API: the mirror models load the
    reference's state_dict, the explanation runs on explainer/torch_route.py (here on the CPU through the test hook) and matches
    the reference's own output."""
    monkeypatch.setitem(explain._ENGINE, "device", "cpu")
    _route_case(tmp_path, tag, kw)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", [("att", dict(method="att")), ("l4", dict(num_gc_layers=4))])
def test_torch_route_vs_reference_on_gpu(tmp_path, tag, kw):
    _route_case(tmp_path, tag, kw)


def test_torch_route_refuses_a_cpu_only_host(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ck, ex = _route_explainer(tmp_path, "att", method="att")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ex.explain(302)
