#!/usr/bin/env python
"""Golden fixture for what the reference PRINTS every epoch (explain.py:148-159): the mask density (ExplainModule.mask_density, :680-683 -
SURVEY.md 8(a) row a12) and the class probabilities `ypred` (ExplainModule.forward, :710-714), next to the loss - recorded from the LIVE
reference (/root/reference, build container only) on the encoder weights of the committed syn1 fixture, seed protocol of SURVEY.md 8(d).

    python tests/golden/make_golden_logging.py        # -> tests/golden/logging_explain.npz (a few seconds)

Per target t: `t:density` [epochs] f32, `t:pred` [epochs, C] f32, `t:loss` [epochs] f32 (the same run's totals)."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg

TARGETS = (302, 309, 300)      # n = 6, 48 and the largest neighbourhood of the motif nodes (310): the three kernel classes of one mixed launch
EPOCHS = 40


def main():
    mg.install_shims()
    import models
    from explainer import explain
    ck = np.load(os.path.join(HERE, "syn1_ckpt.npz"))
    N = int(ck["num_nodes"])
    adj = np.zeros((N, N))
    adj[ck["edges"][:, 0], ck["edges"][:, 1]] = 1.0
    adj = adj + adj.T
    work = tempfile.mkdtemp(prefix="gnnx_golden_logging_")
    args = mg.explain_args("syn1", work, EPOCHS)
    os.makedirs(args.logdir, exist_ok=True)
    model = models.GcnEncoderNode(input_dim=ck["feat"].shape[1], hidden_dim=20, embedding_dim=20, label_dim=ck["pred"].shape[1], num_layers=3,
                                  bn=False, args=args)
    model.load_state_dict({k[2:]: torch.tensor(ck[k]) for k in ck.files if k.startswith("w:")})
    rec = {}
    cls = explain.ExplainModule
    orig_density, orig_forward, orig_loss = cls.mask_density, cls.forward, cls.loss

    def density(self):
        out = orig_density(self)
        rec["density"].append(float(out))
        return out

    def forward(self, *a, **k):
        res, att = orig_forward(self, *a, **k)
        rec["pred"].append(res.detach().numpy().astype(np.float32).copy())
        return res, att

    def loss(self, *a, **k):
        out = orig_loss(self, *a, **k)
        rec["loss"].append(float(out))
        return out

    cls.mask_density, cls.forward, cls.loss = density, forward, loss
    with mg.quiet():
        ex = explain.Explainer(model=model, adj=adj[None], feat=ck["feat"][None].astype(np.float64), label=ck["label"][None], pred=ck["pred"][None],
                               train_idx=None, args=args, writer=None, print_training=False, graph_mode=False, graph_idx=-1)
    out = dict(epochs=np.int64(EPOCHS), targets=np.asarray(TARGETS, np.int64))
    for t in TARGETS:
        rec.update(density=[], pred=[], loss=[])
        torch.manual_seed(1000 + t)
        with mg.quiet():
            ex.explain(t)
        assert len(rec["density"]) == EPOCHS and len(rec["pred"]) == EPOCHS
        out[f"{t}:density"] = np.asarray(rec["density"], np.float32)
        out[f"{t}:pred"] = np.stack(rec["pred"]).astype(np.float32)
        out[f"{t}:loss"] = np.asarray(rec["loss"], np.float32)
        print(f"target {t}: density {rec['density'][0]:.6f} -> {rec['density'][-1]:.6f}, pred[0] {rec['pred'][0]}, loss {rec['loss'][0]:.4f} -> {rec['loss'][-1]:.4f}")
    np.savez_compressed(os.path.join(HERE, "logging_explain.npz"), **out)


if __name__ == "__main__":
    main()
