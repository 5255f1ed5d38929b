#!/usr/bin/env python
"""Golden outputs of the REAL reference (/root/reference) for the options round 3 moved onto the engine, and for the one
quantitative end-use number the reference produces.  Build container only; nothing is copied, the reference is imported and run.

    python tests/golden/make_golden_options.py [--work /tmp/gw/work]

options_explain.npz (syn1 checkpoint of make_golden.py, seed protocol torch.manual_seed(1000 + target) before each explanation):
  opt:<name>:<t>:masked_adj_edges / :feat_sig      --opt sgd | rmsprop | adagrad (utils/train_utils.py:11-16), 100 epochs
  sched:<name>:<t>:masked_adj_edges / :feat_sig / :lr   --opt-scheduler step (--opt-decay-step 30 --opt-decay-rate 0.5) | cos (--opt-restart 100)
                                                   with Adam (train_utils.py:19-22; stepped after every optimiser step, explain.py:144-146);
                                                   lr = the learning rate the scheduler leaves in optimizer.param_groups[0]["lr"] per epoch
  unc:<t>:masked_adj_edges / :feat_sig             Explainer.explain(t, unconstrained=True) (explain.py:688-691: the masked adjacency is
                                                   sym(sigmoid(mask)) * (1 - I), NOT multiplied by adj; the result still is, :209-211)
  route:att | l4:...                               method="att" and num_gc_layers=4 encoders trained by the reference's train.py: weights, predictions and
                                                   the explanations of two targets (the configurations explainer/torch_route.py serves)
  auc:<dataset>:<model>                            ROC-AUC the reference's explain_nodes_gnn_stats(range(400, 700, 5)) writes to
                                                   log/pr/auc_<dataset>_<model>.txt (explain.py:295-353) for syn1 / syn4, model exp | grad,
                                                   100 epochs (the CLI default), torch.manual_seed(0) before the call;
  auc:<dataset>:exp:pred_sum                       checksum of the concatenated edge scores that AUC was computed from
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def mg_adj(work, io_utils):
    with mg.quiet():
        return io_utils.load_ckpt(mg.explain_args("syn1", work, 1))["cg"]["adj"][0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--work", default="/tmp/gw/work")
    a = ap.parse_args()
    mg.install_shims()
    import torch
    torch.set_num_threads(1)
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    io_utils.log_graph = lambda *a_, **k_: None            # needs a TensorBoard writer
    for ds in ("syn1", "syn4"):
        if not os.path.exists(os.path.join(a.work, "ckpt", f"{ds}_base_h20_o20.pth.tar")):
            mg.mint_checkpoint(ds, a.work)
    os.makedirs(os.path.join(a.work, "log"), exist_ok=True)
    built = mg.capture_module(explain)
    out = {}

    def explainer(ds, epochs, **kw):
        args = mg.explain_args(ds, a.work, epochs)
        for k, v in kw.items():
            setattr(args, k, v)
        with mg.quiet():
            ckpt = io_utils.load_ckpt(mg.explain_args(ds, a.work, epochs))
        cg = ckpt["cg"]
        model = models.GcnEncoderNode(input_dim=10, hidden_dim=20, embedding_dim=20, label_dim=cg["pred"].shape[2], num_layers=3, bn=False, args=args)
        model.load_state_dict(ckpt["model_state"])
        with mg.quiet():
            ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"],
                                   train_idx=cg["train_idx"], args=args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
        return ex, args

    def one(t, epochs=100, unconstrained=False, **kw):
        ex, args = explainer("syn1", epochs, **kw)
        with mg.quiet():
            torch.manual_seed(1000 + t)
            _, sub_adj, _, _, nb = ex.extract_neighborhood(t)
            ma = ex.explain(t, unconstrained=unconstrained)
        mod = built[-1]
        r, c = np.nonzero(np.triu(sub_adj, 1))
        assert not np.isnan(ma).any()
        return ma[r, c].astype(np.float32), torch.sigmoid(mod.feat_mask).detach().numpy(), mod

    for t in (302, 309):
        for name in ("sgd", "rmsprop", "adagrad"):
            v, fs, _ = one(t, opt=name)
            out[f"opt:{name}:{t}:masked_adj_edges"], out[f"opt:{name}:{t}:feat_sig"] = v, fs
            print(f"target {t} --opt {name}: masked_adj in [{v.min():.4f}, {v.max():.4f}]")
        for name, kw in (("step", dict(opt_decay_step=30, opt_decay_rate=0.5)), ("cos", dict(opt_restart=100))):
            # the learning rate per epoch, read off the reference's own optimiser: wrap scheduler.step
            lrs = []
            orig = explain.ExplainModule.__init__

            def init(self, *aa, **kk):
                orig(self, *aa, **kk)
                opt, step = self.optimizer, self.optimizer.step

                def wrapped(*sa, **sk):
                    lrs.append(opt.param_groups[0]["lr"])
                    return step(*sa, **sk)
                opt.step = wrapped
            explain.ExplainModule.__init__ = init
            try:
                v, fs, _ = one(t, opt_scheduler=name, **kw)
            finally:
                explain.ExplainModule.__init__ = orig
            out[f"sched:{name}:{t}:masked_adj_edges"], out[f"sched:{name}:{t}:feat_sig"] = v, fs
            out[f"sched:{name}:{t}:lr"] = np.asarray(lrs, np.float64)
            print(f"target {t} --opt-scheduler {name}: lr {lrs[0]:.4f} -> {lrs[-1]:.6f}; masked_adj in [{v.min():.4f}, {v.max():.4f}]")
        v, fs, _ = one(t, unconstrained=True)
        out[f"unc:{t}:masked_adj_edges"], out[f"unc:{t}:feat_sig"] = v, fs
        print(f"target {t} unconstrained: masked_adj in [{v.min():.4f}, {v.max():.4f}]")

    # ---- configurations that take the PyTorch-ROCm route: method="att" (models.py:62-68) and a 4-layer encoder, each trained by the
    # reference's own train.py (syn_task1, 1000 epochs, seed 0) and explained for 100 epochs ----
    import random
    import train
    for tag, kw in (("att", dict(method="att")), ("l4", dict(num_gc_layers=4))):
        work = os.path.join(a.work, "route_" + tag)
        targs = mg.train_args("syn1", work)
        for k, v in kw.items():
            setattr(targs, k, v)
        np.random.seed(0)
        random.seed(0)
        torch.manual_seed(0)
        with mg.quiet():
            train.syn_task1(targs)
        eargs = mg.explain_args("syn1", work, 100)
        for k, v in kw.items():
            setattr(eargs, k, v)
        os.makedirs(eargs.logdir, exist_ok=True)
        with mg.quiet():
            ckpt = io_utils.load_ckpt(eargs)
        cg = ckpt["cg"]
        model = models.GcnEncoderNode(input_dim=10, hidden_dim=20, embedding_dim=20, label_dim=4, num_layers=eargs.num_gc_layers, bn=False, args=eargs)
        model.load_state_dict(ckpt["model_state"])
        for k, v in ckpt["model_state"].items():
            out[f"route:{tag}:w:{k}"] = v.detach().numpy().astype(np.float32)
        out[f"route:{tag}:pred"] = cg["pred"][0].astype(np.float32)
        assert np.array_equal(cg["adj"][0], mg_adj(a.work, io_utils)), "the graph of the route checkpoints must be syn1's"
        with mg.quiet():
            ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"], train_idx=cg["train_idx"],
                                   args=eargs, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
        for t in (302, 309):
            with mg.quiet():
                torch.manual_seed(1000 + t)
                _, sub_adj, _, _, nb = ex.extract_neighborhood(t)
                ma = ex.explain(t)
            r, c = np.nonzero(np.triu(sub_adj, 1))
            out[f"route:{tag}:{t}:neighbors"] = nb.astype(np.int32)
            out[f"route:{tag}:{t}:masked_adj_edges"] = ma[r, c].astype(np.float32)
            out[f"route:{tag}:{t}:feat_sig"] = torch.sigmoid(built[-1].feat_mask).detach().numpy()
            print(f"route {tag} target {t}: n={len(nb)} masked_adj in [{ma[r, c].min():.4f}, {ma[r, c].max():.4f}]")

    # ---- the reference's own end-use metric (explain.py:295-353) ----
    cwd = os.getcwd()
    os.makedirs(os.path.join(a.work, "aucrun", "log", "pr"), exist_ok=True)
    os.chdir(os.path.join(a.work, "aucrun"))
    try:
        for ds in ("syn1", "syn4"):
            for model_kind in ("exp", "grad"):
                ex, args = explainer(ds, 100)
                nodes = range(400, 700, 5) if ds == "syn1" else range(511, 871, 6)      # syn4: every 6th motif node (the CLI's range is syn1's)
                preds = []
                orig_mpr = ex.make_pred_real

                def mpr(adj, start, _o=orig_mpr):
                    p, r_ = _o(adj, start)
                    preds.append(p)
                    return p, r_
                ex.make_pred_real = mpr
                with mg.quiet():
                    torch.manual_seed(0)
                    ex.explain_nodes_gnn_stats(nodes, args, model=model_kind)
                txt = open(f"log/pr/auc_{ds}_{model_kind}.txt").read()
                auc = float(txt.strip().split("auc: ")[1])
                out[f"auc:{ds}:{model_kind}"] = np.float64(auc)
                out[f"auc:{ds}:{model_kind}:nodes"] = np.asarray(list(nodes), np.int64)
                out[f"auc:{ds}:{model_kind}:pred_sum"] = np.float64(np.concatenate(preds).astype(np.float64).sum())
                print(f"{ds} model={model_kind}: reference AUC over {len(list(nodes))} nodes = {auc:.6f}")
    finally:
        os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "options_explain.npz"), **out)


if __name__ == "__main__":
    main()
