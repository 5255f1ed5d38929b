#!/usr/bin/env python
"""Golden outputs of the REAL reference (/root/reference) under its optional flags, same seed
protocol as make_golden.py.  Build container only.

    python tests/golden/make_golden_flags.py [--work /tmp/gw/work]

flags_explain.npz:
  mask_bias:<t>:masked_adj_edges  --mask-bias run (explain.py:657-661, 674-677) - bit-identical to the plain run
                                  (the bias mask is initialised to 0 and ReLU6 has no gradient there)
  relu:<t>:nan_fraction           mask_act="ReLU" (explain.py:669-670, 757-760): the reference's entropy term takes
                                  log(1 - relu(M)) of entries > 1 -> NaN loss -> NaN masks after the first step
  bn:<t>:masked_adj_edges, bn:<t>:feat_sig, bn:<t>:cond
                                  --bn run (apply_bn, models.py:222-228, 241-253): a fresh BatchNorm1d(num_nodes) in training mode
                                  after the ReLU of the two hidden layers; cond = deviation of the closed-form fp32 oracle
  grad:<t>:masked_adj_edges       model="grad" baseline (explain.py:125-133, 717-738): sigmoid(|dL/dA| + |dL/dA|^T) * A
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


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
    if not os.path.exists(os.path.join(a.work, "ckpt", "syn1_base_h20_o20.pth.tar")):
        mg.mint_checkpoint("syn1", a.work)
    os.makedirs(os.path.join(a.work, "log"), exist_ok=True)
    out = {}

    def run(t, model_kind="exp", **kw):
        args = mg.explain_args("syn1", a.work, 300)
        for k, v in kw.items():
            setattr(args, k, v)
        with mg.quiet():
            ckpt = io_utils.load_ckpt(mg.explain_args("syn1", a.work, 300))
        cg = ckpt["cg"]
        model = models.GcnEncoderNode(input_dim=10, hidden_dim=20, embedding_dim=20, label_dim=4, num_layers=3, bn=args.bn, args=args)
        model.load_state_dict(ckpt["model_state"])
        with mg.quiet():
            ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"],
                                   train_idx=cg["train_idx"], args=args, writer=None, print_training=False, graph_mode=False,
                                   graph_idx=-1)
            torch.manual_seed(1000 + t)
            new_idx, sub_adj, _, _, nb = ex.extract_neighborhood(t)
            ma = ex.explain(t, model=model_kind)
        return ma, sub_adj, nb

    built = mg.capture_module(explain)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import closed_form, reference_restatement as rr
    for t in (302, 555):
        base, sub, nb = run(t)
        r, c = np.nonzero(np.triu(sub, 1))
        ma, _, _ = run(t, mask_bias=True)
        assert np.array_equal(ma, base), "the reference's --mask-bias run differs from its plain run"
        out[f"mask_bias:{t}:masked_adj_edges"] = ma[r, c].astype(np.float32)
        ma, _, _ = run(t, mask_act="ReLU")
        out[f"relu:{t}:nan_fraction"] = np.float64(np.isnan(ma).mean())
        print(f"target {t}: --mask-bias bit-identical to the plain run; mask_act=ReLU -> {np.isnan(ma).mean():.0%} NaN")
    ck = dict(np.load(os.path.join(HERE, "syn1_ckpt.npz")))
    sd = {k[2:]: v for k, v in ck.items() if k.startswith("w:")}
    for t in (302, 309, 555, 400):
        ma, sub, nb = run(t, bn=True)
        mod = built[-1]
        r, c = np.nonzero(np.triu(sub, 1))
        fsig = torch.sigmoid(mod.feat_mask).detach().numpy()
        new = int(np.searchsorted(nb, t))
        pl = np.argmax(ck["pred"][nb], 1)
        o = closed_form.ClosedFormOracle(sub.astype(np.float32), ck["feat"][nb], sd, int(ck["label"][t]), pl, new, mod.mask0.numpy(), bn=True)
        got = o.run(300)
        cm = float(np.abs(got - ma).max())
        cf = float(np.abs(1 / (1 + np.exp(-o.f.astype(np.float64))) - fsig).max())
        # the torch restatement with bn=True must reproduce the reference bit for bit
        oo = rr.MaskOptimOracle(torch.tensor(sub.astype(np.float32)), torch.tensor(ck["feat"][nb]), {k: torch.tensor(v) for k, v in sd.items()},
                                int(ck["label"][t]), pl, new, mask0=mod.mask0.clone(), bn=True)
        assert np.array_equal(oo.run(300), ma), "restatement (bn) is not bit-identical to the reference"
        out[f"bn:{t}:neighbors"] = nb.astype(np.int32)
        out[f"bn:{t}:masked_adj_edges"] = ma[r, c].astype(np.float32)
        out[f"bn:{t}:feat_sig"] = fsig
        out[f"bn:{t}:cond"] = np.asarray([cm, cf], np.float32)
        print(f"target {t}: --bn n={len(nb)} loss {mod.loss_trace[0]:.4f} -> {mod.loss_trace[-1]:.4f}; closed form vs reference {cm:.2e} / {cf:.2e}; "
              f"restatement bit-identical")
    for t in (302, 309, 555, 330, 400, 300):
        ma, sub, nb = run(t, model_kind="grad")
        r, c = np.nonzero(np.triu(sub, 1))
        assert np.array_equal(ma, ma.T) and not np.isnan(ma).any()
        out[f"grad:{t}:neighbors"] = nb.astype(np.int32)
        out[f"grad:{t}:masked_adj_edges"] = ma[r, c].astype(np.float32)
        print(f"target {t}: grad baseline n={len(nb)} edges={len(r)} range [{ma[r, c].min():.4f}, {ma[r, c].max():.4f}]")
    np.savez_compressed(os.path.join(HERE, "flags_explain.npz"), **out)


if __name__ == "__main__":
    main()
