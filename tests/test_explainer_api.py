"""Drop-in API tests: the Explainer/ExplainModule mirror of explainer/explain.py, seeded like the reference's
golden runs.  On CPU the engine is swapped for the emulator build of the same sources (tests/emu); on a GPU
box the same assertions run against libgnnx_hip.so (marked gpu)."""
import argparse
import os

import numpy as np
import pytest
import torch

import helpers
from gnn_model_explainer_amd import models
from gnn_model_explainer_amd.explainer import explain

TOL = 1e-5


def _args(tmp, epochs, dataset="syn1", **kw):
    a = argparse.Namespace(logdir=str(tmp), ckptdir=str(tmp), dataset=dataset, bmname=None, opt="adam",
                           opt_scheduler="none", lr=0.1, num_epochs=epochs, hidden_dim=20, output_dim=20,
                           num_gc_layers=3, method="base", name_suffix="", explainer_suffix="", graph_idx=-1,
                           mask_act="sigmoid", mask_bias=False, bn=False, bias=True, gpu=True)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def _explainer(tmp, epochs, name="syn1", **kw):
    ck = helpers.load_ckpt(name)
    args = _args(tmp, epochs, name, **kw)
    model = models.GcnEncoderNode(ck["feat"].shape[1], 20, 20, ck["pred"].shape[1], 3, bn=False, args=args)
    model.load_state_dict({k: torch.tensor(v) for k, v in ck["sd"].items()})
    ex = explain.Explainer(model, ck["adj"][None].astype(np.float64), ck["feat"][None].astype(np.float64),
                           ck["label"][None], ck["pred"][None], None, args, writer=None, print_training=False,
                           graph_mode=False, graph_idx=-1)
    return ck, args, ex


@pytest.fixture
def emu_engine(monkeypatch):
    """Point the Explainer / ExplainModule at the emulator build of the same engine sources (no GPU in this container)."""
    from emu.emu_engine import emu_library
    monkeypatch.setitem(explain._ENGINE, "lib", emu_library())
    monkeypatch.setitem(explain._ENGINE, "device", "cpu")


def _check_against_golden(ex, args, gx, t, tmp, epochs_full):
    torch.manual_seed(1000 + t)                                  # the golden seed protocol
    ma = ex.explain(t)
    assert ma.dtype == np.float64
    nb = gx[f"{t}:neighbors"]
    assert ma.shape == (len(nb), len(nb))
    f = os.path.join(str(tmp), "masked_adj_syn1_base_h20_o20_explainnode_idx_%dgraph_idx_-1.npy" % t)
    assert np.array_equal(np.load(f), ma)                        # same file name and content as the reference writes
    if epochs_full:
        rc = gx[f"{t}:edge_rc"]
        assert np.abs(ma[rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max() <= TOL
        fm = 1 / (1 + np.exp(-ex.last_result.feat_mask[0]))
        assert np.abs(fm - gx[f"{t}:feat_mask_sigmoid"]).max() <= TOL


def test_explain_single_node_matches_reference_emulated(tmp_path, emu_engine):
    gx = helpers.load_explain("syn1")
    ck, args, ex = _explainer(tmp_path, int(gx["epochs"]))
    _check_against_golden(ex, args, gx, 302, tmp_path, True)


def test_extract_neighborhood_contract(tmp_path, emu_engine):
    gx = helpers.load_explain("syn1")
    ck, args, ex = _explainer(tmp_path, 3)
    new, sub_adj, sub_feat, sub_label, nb = ex.extract_neighborhood(309)
    assert new == int(gx["309:node_idx_new"]) and np.array_equal(nb, gx["309:neighbors"])
    assert sub_adj.dtype == np.float64 and sub_adj.shape == (len(nb), len(nb))
    assert sub_feat.shape == (len(nb), 10) and np.array_equal(sub_label, ck["label"][nb])
    assert np.array_equal(ex.neighborhoods[0][309].nonzero()[0], nb)


def test_batched_list_api_equals_sequential_explains(tmp_path, emu_engine):
    """explain_nodes([...]) (one batched job) == [explain(i) ...] with the same RNG stream."""
    ck, args, ex = _explainer(tmp_path, 4)
    torch.manual_seed(7)
    seq = [ex.explain(v) for v in (302, 309)]
    torch.manual_seed(7)
    bat = ex.explain_nodes([302, 309], args)
    for a, b in zip(seq, bat):
        assert np.array_equal(a, b)


def test_unknown_options_fail_like_the_reference(tmp_path):
    """Every option the reference's CLI accepts runs (kernels, or explainer/torch_route.py: tests/test_options.py); names its
    build_optimizer does not know fail at construction, as there (utils/train_utils.py:9-22 leaves `optimizer` unbound)."""
    for kw in ({"opt": "lbfgs"}, {"opt_scheduler": "plateau"}):
        with pytest.raises(ValueError):
            _explainer(tmp_path, 3, **kw)
    if not torch.cuda.is_available():
        ck, args, ex = _explainer(tmp_path, 3)
        with pytest.raises(RuntimeError, match="no CPU fallback"):      # the torch route needs the HIP device too
            ex.explain(302, model="att")


def test_grad_baseline_and_mask_bias_through_the_api_emulated(tmp_path, emu_engine):
    """`explain(node, model="grad")` (explain.py:125-133) and `--mask-bias` (explain.py:657-661, 674-677) against the REAL
    reference's outputs under those options (tests/golden/flags_explain.npz)."""
    z = np.load(helpers.GOLDEN + "/flags_explain.npz")
    ck, args, ex = _explainer(tmp_path, 300)
    ma = ex.explain(309, model="grad")
    nb = z["grad:309:neighbors"]
    r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
    assert ma.dtype == np.float64 and ma.shape == (len(nb), len(nb)) and np.array_equal(ma, ma.T)
    assert np.abs(ma[r, c] - z["grad:309:masked_adj_edges"]).max() <= 1e-6
    assert os.path.exists(os.path.join(str(tmp_path), "masked_adj_syn1_base_h20_o20_explainnode_idx_309graph_idx_-1.npy"))
    # --mask-bias: the reference's bias mask stays exactly 0, its output equals the plain run bit for bit
    ck, args, ex = _explainer(tmp_path, 300, mask_bias=True)
    torch.manual_seed(1000 + 302)
    ma = ex.explain(302)
    nb = helpers.load_explain("syn1")["302:neighbors"]
    r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
    assert np.abs(ma[r, c] - z["mask_bias:302:masked_adj_edges"]).max() <= TOL
    gx = helpers.load_explain("syn1")                         # ... and the fixture of the flag run IS the plain run's output
    plain = np.zeros((len(nb), len(nb)), np.float32)
    plain[gx["302:edge_rc"][:, 0], gx["302:edge_rc"][:, 1]] = gx["302:masked_adj_edges"]
    assert np.array_equal(plain[r, c], z["mask_bias:302:masked_adj_edges"])


def test_gnn_stats_device_auc_and_denoise_equal_host_postprocessing_emulated(tmp_path, emu_engine, monkeypatch):
    """explain_nodes_gnn_stats (explain.py:295-353): the ROC-AUC from the device's pair counts equals sklearn's on the returned
    masks, and the device's denoised edge sets equal io_utils.denoise_graph(threshold_num=20) on them."""
    from sklearn.metrics import roc_auc_score
    from gnn_model_explainer_amd.utils import io_utils
    monkeypatch.chdir(tmp_path)
    ck, args, ex = _explainer(tmp_path, 3)
    nodes = [400, 405, 555, 600]
    torch.manual_seed(3)
    out = ex.explain_nodes_gnn_stats(nodes, args)
    pr = [ex.make_pred_real(ma, int(st)) for ma, st in zip(out, ex.last_rows)]
    want = roc_auc_score(np.concatenate([p[1] for p in pr]), np.concatenate([p[0] for p in pr]))
    assert abs(ex.last_auc - want) < 1e-12 and os.path.exists("log/pr/auc_syn1_exp.txt")
    keep, thr, stats = ex.last_result.denoised
    em = ex.last_result.edges
    for k, (ma, st) in enumerate(zip(out, ex.last_rows)):
        G = io_utils.denoise_graph(ma, int(st), threshold_num=20)
        a, b = em.eoff[k], em.eoff[k + 1]
        kept = em.rc[a:b][keep[a:b]]
        assert sorted(map(tuple, kept.tolist())) == sorted((min(u, v), max(u, v)) for u, v in G.edges())
        assert stats[k][0] == G.number_of_nodes() and stats[k][1] == G.number_of_edges()


def _explain_module_surface(tmp_path):
    gx = helpers.load_explain("syn1")
    ck, args, ex = _explainer(tmp_path, 5)
    t = 302
    new, sub_adj, sub_feat, sub_label, nb = ex.extract_neighborhood(t)
    pl = np.argmax(ck["pred"][nb], 1)
    torch.manual_seed(1000 + t)
    mod = explain.ExplainModule(torch.tensor(sub_adj[None], dtype=torch.float), torch.tensor(sub_feat[None], dtype=torch.float),
                                ex.model, torch.tensor(sub_label[None]), args, graph_idx=-1, node_idx=new, pred_label=pl)
    assert np.array_equal(mod.mask.detach().numpy(), gx[f"{t}:mask0"])       # same init stream as the reference
    pred, _ = mod.forward(new)
    assert abs(float(pred.sum()) - 1) < 1e-5 and mod.masked_adj.shape == (1, len(nb), len(nb))
    loss0 = float(mod.loss(pred, pl, new, 0))
    assert abs(loss0 - float(gx[f"{t}:loss"][0])) < 1e-4                     # reference's epoch-0 loss
    # mask_density (explain.py:822-825): sum of the masked adjacency over the number of edge entries of the sub-graph
    ma0 = mod.masked_adj.detach().cpu().numpy()[0]
    want = float(ma0.sum() / (sub_adj != 0).sum())
    assert abs(float(mod.mask_density()) - want) < 1e-6 and 0 < want < 1
    # .optimizer / .scheduler (explain.py:620-622): the torch objects build_optimizer returns, over the mirror's parameters
    assert isinstance(mod.optimizer, torch.optim.Adam) and mod.scheduler is None
    assert mod.optimizer.param_groups[0]["lr"] == args.lr and mod.optimizer.param_groups[0]["params"][0] is mod.mask
    mod.optimize(5)
    assert not np.array_equal(mod.mask.detach().numpy(), gx[f"{t}:mask0"])
    pred5, _ = mod.forward(new)
    assert float(mod.loss(pred5, pl, new, 5)) < loss0                         # five Adam steps lowered the loss
    # ... and after the run its state is the state the reference's Adam holds after five steps (the oracle runs torch.optim.Adam)
    from oracle import reference_restatement as rr
    o = rr.MaskOptimOracle(torch.tensor(sub_adj, dtype=torch.float), torch.tensor(sub_feat, dtype=torch.float),
                           {k: torch.tensor(v) for k, v in ck["sd"].items()}, int(sub_label[new]), pl, new, mask0=torch.tensor(gx[f"{t}:mask0"]))
    o.run(5)
    st, want = mod.optimizer.state[mod.mask], o.opt.state[o.mask]
    assert float(st["step"]) == 5 and st["exp_avg"].shape == (len(nb), len(nb))
    assert (st["exp_avg"] - want["exp_avg"]).abs().max() < 1e-6 and (st["exp_avg_sq"] - want["exp_avg_sq"]).abs().max() < 1e-7
    assert (mod.optimizer.state[mod.feat_mask]["exp_avg"] - o.opt.state[o.feat_mask]["exp_avg"]).abs().max() < 1e-6
    assert (mod.mask.detach() - o.mask.detach()).abs().max() < 1e-5
    # a second call (k != args.num_epochs) continues: moments, step count of the bias corrections - like the reference's optimiser
    mod.optimize(3)
    o.run(3)
    st, want = mod.optimizer.state[mod.mask], o.opt.state[o.mask]
    assert float(st["step"]) == 8
    assert (st["exp_avg"] - want["exp_avg"]).abs().max() < 1e-6 and (st["exp_avg_sq"] - want["exp_avg_sq"]).abs().max() < 1e-7
    assert (mod.mask.detach() - o.mask.detach()).abs().max() < 1e-5
    sgd = explain.ExplainModule(torch.tensor(sub_adj[None], dtype=torch.float), torch.tensor(sub_feat[None], dtype=torch.float), ex.model,
                                torch.tensor(sub_label[None]), _args(tmp_path, 5, opt="sgd", opt_scheduler="step", opt_decay_step=2, opt_decay_rate=0.5),
                                graph_idx=-1, node_idx=new, pred_label=pl)
    assert isinstance(sgd.optimizer, torch.optim.SGD) and isinstance(sgd.scheduler, torch.optim.lr_scheduler.StepLR)
    sgd.optimize(5)
    assert "momentum_buffer" in sgd.optimizer.state[sgd.mask] and abs(sgd.optimizer.param_groups[0]["lr"] - 0.1 * 0.25) < 1e-12
    sgd.optimize(3)        # the schedule continues at step 5: lr 0.1 * 0.5 ** (8 // 2) afterwards
    assert abs(sgd.optimizer.param_groups[0]["lr"] - 0.1 * 0.5 ** 4) < 1e-12


def test_explain_module_surface_emulated(tmp_path, emu_engine):
    _explain_module_surface(tmp_path)


@pytest.mark.gpu
def test_explain_module_surface_on_gpu(tmp_path):
    """forward / loss / mask_density / optimize of the ExplainModule mirror through libgnnx_hip.so."""
    _explain_module_surface(tmp_path)


@pytest.mark.gpu
def test_explain_matches_reference_on_gpu(tmp_path):
    gx = helpers.load_explain("syn1")
    ck, args, ex = _explainer(tmp_path, int(gx["epochs"]))
    for t in (302, 555):
        _check_against_golden(ex, args, gx, t, tmp_path, True)


@pytest.mark.gpu
def test_explain_nodes_gnn_stats_auc_on_gpu(tmp_path, monkeypatch):
    """The reference's only quantitative check (ROC-AUC vs motif ground truth, explain.py:328-351)."""
    monkeypatch.chdir(tmp_path)
    ck, args, ex = _explainer(tmp_path, 100)
    torch.manual_seed(0)
    out = ex.explain_nodes_gnn_stats(range(400, 700, 5), args)
    assert len(out) == 60 and ex.last_auc > 0.8
    assert os.path.exists("log/pr/auc_syn1_exp.txt")


@pytest.mark.gpu
@pytest.mark.parametrize("name,model", [("syn1", "exp"), ("syn1", "grad"), ("syn4", "exp"), ("syn4", "grad")])
def test_auc_equals_the_number_the_reference_prints(tmp_path, monkeypatch, name, model):
    """The reference's one quantitative output - the ROC-AUC explain_nodes_gnn_stats writes to log/pr/auc_<dataset>_<model>.txt
    (explain.py:325-351) - for the CLI's node range at the CLI's 100 epochs, same global seed: tests/golden/options_explain.npz holds
    the number the LIVE reference wrote (make_golden_options.py).  The masks come from the same RNG stream (one normal_ per target in
    list order), so the device AUC must agree to 1e-3 (and the edge scores it is computed from, in sum, to 1e-3 relative)."""
    monkeypatch.chdir(tmp_path)
    z = np.load(helpers.GOLDEN + "/options_explain.npz")
    ck, args, ex = _explainer(tmp_path, 100, name)
    nodes = [int(v) for v in z[f"auc:{name}:{model}:nodes"]]
    torch.manual_seed(0)
    out = ex.explain_nodes_gnn_stats(nodes, args, model=model)
    want = float(z[f"auc:{name}:{model}"])
    print(f"{name} {model}: AUC {ex.last_auc:.6f}, the reference wrote {want:.6f}")
    assert abs(ex.last_auc - want) <= 1e-3
    got_sum = sum(float(m[np.triu(m) > 0].sum()) for m in out)
    assert abs(got_sum - float(z[f"auc:{name}:{model}:pred_sum"])) <= 1e-3 * abs(float(z[f"auc:{name}:{model}:pred_sum"]))


@pytest.mark.gpu
def test_grad_baseline_batch_and_auc_on_gpu(tmp_path, monkeypatch):
    """The gradient baseline through the batched API on the GPU: golden parity on six targets (n = 6 ... 310) and the
    reference's AUC evaluation (explain_nodes_gnn_stats(..., model="grad"))."""
    monkeypatch.chdir(tmp_path)
    z = np.load(helpers.GOLDEN + "/flags_explain.npz")
    ck, args, ex = _explainer(tmp_path, 100)
    targets = [302, 309, 555, 330, 400, 300]
    for t, ma in zip(targets, ex.explain_grad(targets)):
        nb = z[f"grad:{t}:neighbors"]
        r, c = np.nonzero(np.triu(ck["adj"][np.ix_(nb, nb)], 1))
        assert np.abs(ma[r, c] - z[f"grad:{t}:masked_adj_edges"]).max() <= 1e-6
    out = ex.explain_nodes_gnn_stats(range(400, 700, 5), args, model="grad")
    assert len(out) == 60 and 0.5 < ex.last_auc <= 1.0
    assert os.path.exists("log/pr/auc_syn1_grad.txt")


def _write_reference_format_ckpt(tmp, name="syn1"):
    """A checkpoint laid out like the reference's io_utils.save_checkpoint (io_utils.py:81-103)."""
    ck = helpers.load_ckpt(name)
    d = os.path.join(str(tmp), "ckpt", f"{name}_base_h20_o20")
    os.makedirs(d, exist_ok=True)
    cg = {"adj": ck["adj"][None].astype(np.float64), "feat": ck["feat"][None].astype(np.float64),
          "label": ck["label"][None], "pred": ck["pred"][None], "train_idx": list(range(10))}
    torch.save({"epoch": -1, "model_type": "base", "optimizer": None, "optimizer_state": {},
                "model_state": {k: torch.tensor(v) for k, v in ck["sd"].items()}, "cg": cg}, d + ".pth.tar")
    return os.path.join(str(tmp), "ckpt")


def test_cli_single_node_like_reference_emulated(tmp_path, emu_engine, capsys):
    """`explainer_main.py --dataset=syn1 --explain-node=302 --epochs=300` against a reference-format checkpoint."""
    from gnn_model_explainer_amd import explainer_main
    ckptdir = _write_reference_format_ckpt(tmp_path)
    logdir = os.path.join(str(tmp_path), "log")
    torch.manual_seed(5)
    ma = explainer_main.main(["--dataset=syn1", "--explain-node=302", "--epochs=300", "--ckptdir", ckptdir,
                              "--logdir", logdir])
    # same RNG stream through the class API: the CLI builds the encoder (xavier init draws) before the mask init,
    # exactly like the reference's main() (explainer_main.py:225-237, 240-258)
    ck, args, ex = _explainer(tmp_path, 300)
    torch.manual_seed(5)
    models.GcnEncoderNode(10, 20, 20, 4, 3, bn=False, args=args)
    want = ex.explain(302)
    # the CLI prints the loss (-> streaming kernels with loss logging), the API call above took the on-chip-resident
    # path: same result up to summation order
    assert ma.dtype == np.float64 and np.abs(ma - want).max() <= 2e-6
    assert os.path.exists(os.path.join(logdir, "masked_adj_syn1_base_h20_o20_explainnode_idx_302graph_idx_-1.npy"))
    with pytest.raises(Exception, match="File not found"):
        explainer_main.main(["--dataset=syn4", "--explain-node=511", "--ckptdir", ckptdir, "--logdir", logdir])


@pytest.mark.gpu
def test_cli_default_mode_batched_on_gpu(tmp_path, monkeypatch):
    """Default CLI mode (nodes 400..695 step 5, explainer_main.py:309-313) as one batched GPU job."""
    from gnn_model_explainer_amd import explainer_main
    monkeypatch.chdir(tmp_path)
    ckptdir = _write_reference_format_ckpt(tmp_path)
    torch.manual_seed(0)
    out = explainer_main.main(["--dataset=syn1", "--epochs=100", "--ckptdir", ckptdir, "--logdir", str(tmp_path / "log")])
    assert len(out) == 60 and all(np.array_equal(m, m.T) for m in out)
    assert len([f for f in os.listdir(tmp_path / "log") if f.startswith("masked_adj_")]) == 60


def _printed_epochs(text):
    import re
    rows = []
    for line in text.splitlines():
        m = re.match(r"epoch:\s+(\d+)\s+; loss:\s+(\S+)\s+; mask density:\s+(\S+)\s+; pred:\s+\[(.*)\]", line)
        if m:
            rows.append((int(m.group(1)), float(m.group(2)), float(m.group(3)), np.asarray([float(v) for v in m.group(4).split()])))
    return rows


def _check_print_training(ex, capsys, targets):
    """print_training=True: one line per epoch with the reference's fields - "epoch: e ; loss: L ; mask density: d ; pred: [p ...]" (explain.py:149-159) -
    whose values are the LIVE reference's (tests/golden/logging_explain.npz: the density is the one AFTER the epoch's step, :142-148)."""
    z = np.load(os.path.join(helpers.GOLDEN, "logging_explain.npz"))
    ex.print_training = True
    for t in targets:
        torch.manual_seed(1000 + t)
        capsys.readouterr()
        ex.explain(t)
        rows = _printed_epochs(capsys.readouterr().out)
        assert [r[0] for r in rows] == list(range(int(z["epochs"])))
        assert np.allclose([r[1] for r in rows], z[f"{t}:loss"], rtol=1e-5)
        assert np.abs(np.asarray([r[2] for r in rows]) - z[f"{t}:density"]).max() < 1e-5
        assert np.abs(np.stack([r[3] for r in rows]) - z[f"{t}:pred"]).max() < 1e-5


def test_print_training_prints_the_references_fields_on_the_emulator(tmp_path, emu_engine, capsys):
    _, _, ex = _explainer(tmp_path, 40)
    _check_print_training(ex, capsys, (302, 309))


@pytest.mark.gpu
def test_print_training_prints_the_references_fields_on_gpu(tmp_path, capsys):
    _, _, ex = _explainer(tmp_path, 40)
    _check_print_training(ex, capsys, (302, 309, 300))


def _xl_through_the_api(tmp_path, monkeypatch, epochs, check_golden):
    """Explainer.explain / explain_nodes with the larger targets on the XL route (engine.XLJob: CSR-native, edge-list state; explain.XL_MIN_N lowered so
    that syn1's sub-graphs take it): the SAME generator stream as the all-dense path (an XL target's n x n draw passes through the caller's global
    generator like the reference's construct_edge_mask), results within round-off of the dense-packed routes, and of the reference's golden output."""
    gx = helpers.load_explain("syn1")
    ck, args, ex = _explainer(tmp_path, epochs)
    nodes = [309, 302, 330]                       # n = 64 (stays dense-packed at XL_MIN_N = 100), 168 and 129 (XL)
    monkeypatch.setattr(explain, "XL_MIN_N", 1 << 30)
    torch.manual_seed(11)
    dense = ex.explain_nodes(nodes, args)
    after_dense = torch.rand(1).item()
    monkeypatch.setattr(explain, "XL_MIN_N", 100)
    torch.manual_seed(11)
    mixed = ex.explain_nodes(nodes, args)
    assert torch.rand(1).item() == after_dense                      # the generator has advanced exactly as on the dense path
    for v, a, b in zip(nodes, dense, mixed):
        assert a.shape == b.shape and b.dtype == np.float64 and np.array_equal(b, b.T)
        assert np.array_equal(a != 0, b != 0)
        assert np.abs(a - b).max() < 2e-5, (v, np.abs(a - b).max())
    assert np.array_equal(mixed[0], dense[0])                       # the dense-packed target of the mixed batch: the same kernel, the same draw
    if check_golden:
        for t in (302,):
            torch.manual_seed(1000 + t)
            ma = ex.explain(t)
            rc = gx[f"{t}:edge_rc"]
            assert np.abs(ma[rc[:, 0], rc[:, 1]] - gx[f"{t}:masked_adj_edges"]).max() <= TOL
            fm = 1 / (1 + np.exp(-ex.last_result.feat_mask[0]))
            assert np.abs(fm - gx[f"{t}:feat_mask_sigmoid"]).max() <= TOL


def test_xl_route_through_the_api_emulated(tmp_path, emu_engine, monkeypatch):
    _xl_through_the_api(tmp_path, monkeypatch, 5, False)


@pytest.mark.gpu
def test_xl_route_through_the_api_on_gpu(tmp_path, monkeypatch):
    _xl_through_the_api(tmp_path, monkeypatch, 300, True)
