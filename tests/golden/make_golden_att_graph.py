#!/usr/bin/env python
"""TEST INFRASTRUCTURE (build container only): golden outputs of the LIVE reference for `--method att` in GRAPH mode.

    python tests/golden/make_golden_att_graph.py        -> tests/golden/attgraph_explain.npz

Imports /root/reference unmodified (through the shims of make_golden.py), builds its GcnEncoderGraph with args.method = "att"
(models.py:36-37, 62-68; seeded xavier initialisation, non-zero biases so that padded rows matter), explains four small
molecule-like graphs with its own Explainer.explain(graph_mode=True) under the seed protocol (torch.manual_seed(1000 + g) before
each graph) and stores the weights, the inputs and its outputs.  Nothing of the reference is copied."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (sets up sys.path for /root/reference and its shims)


def main():
    import tempfile
    work = tempfile.mkdtemp(prefix="attgraph_")
    mg.install_shims()
    import models
    from explainer import explain
    epochs = 60
    args = mg.explain_args("syn1", work, epochs)
    args.bmname = "Mutagenicity"
    args.graph_mode = True
    args.method = "att"
    os.makedirs(args.logdir, exist_ok=True)
    rng = np.random.default_rng(7)
    torch.manual_seed(7)
    model = models.GcnEncoderGraph(input_dim=14, hidden_dim=20, embedding_dim=20, label_dim=2, num_layers=3, bn=False, args=args)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    graphs = [mg.molecule_like(rng, max_nodes=40) for _ in range(4)]
    adj = torch.tensor(np.stack([g[0] for g in graphs]))
    feat = torch.tensor(np.stack([g[1] for g in graphs]))
    label = torch.tensor(rng.integers(0, 2, len(graphs)), dtype=torch.long)
    model.eval()
    with torch.no_grad():
        pred = model(feat, adj)[0].numpy()[None]
    built = mg.capture_module(explain)
    ex = explain.Explainer(model=model, adj=adj, feat=feat, label=label, pred=pred, train_idx=None, args=args, writer=None,
                           print_training=False, graph_mode=True, graph_idx=0)
    out = dict(epochs=np.int64(epochs), adj=adj.numpy(), feat=feat.numpy(), label=label.numpy(), pred=pred[0],
               num_nodes=np.asarray([g[2] for g in graphs], np.int64))
    for k, v in model.state_dict().items():
        out["w:" + k] = v.detach().numpy().astype(np.float32)
    for g in range(len(graphs)):
        torch.manual_seed(1000 + g)
        with mg.quiet():
            ma = ex.explain(node_idx=0, graph_idx=g, graph_mode=True)
        mod = built[-1]
        assert not np.isnan(ma).any()
        out[f"{g}:masked_adj"] = ma.astype(np.float32)
        out[f"{g}:feat_mask_sigmoid"] = torch.sigmoid(mod.feat_mask).detach().numpy()
        print(f"  graph {g}: nodes={graphs[g][2]} masked_adj on edges in [{ma[adj[g].numpy() > 0].min():.4f}, {ma[adj[g].numpy() > 0].max():.4f}]")
    np.savez_compressed(os.path.join(HERE, "attgraph_explain.npz"), **out)
    print("wrote attgraph_explain.npz")


if __name__ == "__main__":
    main()
