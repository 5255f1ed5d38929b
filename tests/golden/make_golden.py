#!/usr/bin/env python
"""Generate the committed golden fixtures by running the REAL reference (/root/reference).

Runs only in the build container (the reference does not exist on the GPU box).
Nothing from the reference is copied: it is imported, executed on fixed seeds and
its inputs/outputs are stored as small .npz files next to this script.

    python tests/golden/make_golden.py            # everything (about 2 minutes)

Seed protocol:
  * np.random.seed(0); random.seed(0); torch.manual_seed(0) before graph generation + training;
  * torch.manual_seed(1000 + target_id) immediately before each Explainer.explain call.

Fixtures written:
  syn1_ckpt.npz / syn4_ckpt.npz / syn5_ckpt.npz   graph (edge list), features, labels, model predictions, encoder weights
                                   as minted by the reference's own train.py (syn_task1 / syn_task4)
  syn1_explain.npz / syn4_explain.npz / syn5_explain.npz
                                   per target: sub-graph node ids, edge-entry values of the returned
                                   masked_adj (300 epochs), final sigma(feat_mask), final mask at edge
                                   entries, per-epoch loss, and for two small targets the initial mask
  graphmode_explain.npz            graph-mode (GcnEncoderGraph, random-init weights, synthetic padded
                                   molecule-like graphs): inputs + returned masked_adj (100 epochs)
"""
import argparse
import contextlib
import io
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_shims():
    """Harness shims of SURVEY.md §8c (missing optional deps, networkx 3 renames)."""
    tb = types.ModuleType("tensorboardX")
    tbu = types.ModuleType("tensorboardX.utils")

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return lambda *a, **k: None

    tb.SummaryWriter = _Writer
    tbu.figure_to_image = lambda *a, **k: None
    tb.utils = tbu
    sys.modules["tensorboardX"] = tb
    sys.modules["tensorboardX.utils"] = tbu
    sys.modules["seaborn"] = types.ModuleType("seaborn")
    sys.modules["cv2"] = types.ModuleType("cv2")
    import networkx as nx
    nx.to_numpy_matrix = lambda G, *a, **k: np.asmatrix(nx.to_numpy_array(G, *a, **k))
    nx.from_numpy_matrix = nx.from_numpy_array
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _load = torch.load
    torch.load = lambda f, *a, **k: _load(f, *a, **{**k, "weights_only": False})


def train_args(dataset, work):
    return argparse.Namespace(
        datadir="data", logdir=os.path.join(work, "log"), ckptdir=os.path.join(work, "ckpt"), dataset=dataset,
        opt="adam", opt_scheduler="none", max_nodes=100, cuda="0", feature_type="default", lr=0.001, clip=2.0,
        batch_size=20, num_epochs=1000, train_ratio=0.8, test_ratio=0.1, num_workers=1, input_dim=10,
        hidden_dim=20, output_dim=20, num_classes=2, num_gc_layers=3, dropout=0.0, weight_decay=0.005,
        method="base", name_suffix="", assign_ratio=0.1, gpu=False, bn=False, bias=True, bmname=None,
        pkl_fname=None, linkpred=False)


def explain_args(dataset, work, epochs):
    return argparse.Namespace(
        logdir=os.path.join(work, "log"), ckptdir=os.path.join(work, "ckpt"), dataset=dataset, opt="adam",
        opt_scheduler="none", cuda="0", lr=0.1, clip=2.0, batch_size=20, num_epochs=epochs, hidden_dim=20,
        output_dim=20, num_gc_layers=3, dropout=0.0, method="base", name_suffix="", explainer_suffix="",
        align_steps=1000, explain_node=None, graph_idx=-1, mask_act="sigmoid", multigraph_class=-1,
        multinode_class=-1, gpu=False, bn=False, bias=True, bmname=None, mask_bias=False, writer=False,
        graph_mode=False)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def mint_checkpoint(dataset, work):
    import train
    import utils.io_utils as io_utils
    io_utils.log_graph = lambda *a, **k: None       # gen_syn4 calls it with args=None and crashes
    np.random.seed(0)
    random.seed(0)
    torch.manual_seed(0)
    with quiet():
        getattr(train, "syn_task" + dataset[-1])(train_args(dataset, work))


def capture_module(explain_mod):
    """Record every ExplainModule the reference builds and every loss it computes."""
    built = []
    cls = explain_mod.ExplainModule
    if getattr(cls, "_gnnx_recording", False):
        return cls._gnnx_built
    orig_init, orig_loss = cls.__init__, cls.loss

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self.loss_trace = []
        self.mask0 = self.mask.detach().clone()
        built.append(self)

    def loss(self, *a, **k):
        out = orig_loss(self, *a, **k)
        self.loss_trace.append(float(out))
        return out

    cls.__init__, cls.loss = init, loss
    cls._gnnx_recording, cls._gnnx_built = True, built
    return built


def node_fixture(dataset, targets, work, epochs=300, keep_mask0=()):
    import models
    import utils.io_utils as io_utils
    from explainer import explain
    args = explain_args(dataset, work, epochs)
    os.makedirs(args.logdir, exist_ok=True)
    with quiet():
        ckpt = io_utils.load_ckpt(args)
    cg = ckpt["cg"]
    D, C = cg["feat"].shape[2], cg["pred"].shape[2]
    model = models.GcnEncoderNode(input_dim=D, hidden_dim=20, embedding_dim=20, label_dim=C, num_layers=3,
                                  bn=False, args=args)
    model.load_state_dict(ckpt["model_state"])
    adj = cg["adj"][0]
    iu = np.triu_indices_from(adj, 1)
    sel = adj[iu] != 0
    edges = np.stack([iu[0][sel], iu[1][sel]], 1).astype(np.int32)
    assert np.array_equal(adj, adj.T) and set(np.unique(adj)) <= {0.0, 1.0}
    ck = dict(num_nodes=np.int64(adj.shape[0]), edges=edges, feat=cg["feat"][0].astype(np.float32),
              label=cg["label"][0].astype(np.int64), pred=cg["pred"][0].astype(np.float32))
    for k, v in ckpt["model_state"].items():
        ck["w:" + k] = v.detach().numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, dataset + "_ckpt.npz"), **ck)

    built = capture_module(explain)
    with quiet():
        ex = explain.Explainer(model=model, adj=cg["adj"], feat=cg["feat"], label=cg["label"], pred=cg["pred"],
                               train_idx=cg["train_idx"], args=args, writer=None, print_training=False,
                               graph_mode=False, graph_idx=-1)
    out = dict(epochs=np.int64(epochs), targets=np.asarray(targets, np.int64))
    for t in targets:
        torch.manual_seed(1000 + t)
        with quiet():
            new_idx, sub_adj, _, _, nb = ex.extract_neighborhood(t)
            ma = ex.explain(t)
        mod = built[-1]
        r, c = np.nonzero(sub_adj)
        assert ma.dtype == np.float64 and not np.isnan(ma).any()
        assert np.all(ma[sub_adj == 0] == 0)
        out[f"{t}:neighbors"] = nb.astype(np.int32)
        out[f"{t}:node_idx_new"] = np.int64(new_idx)
        out[f"{t}:edge_rc"] = np.stack([r, c], 1).astype(np.int32)
        out[f"{t}:masked_adj_edges"] = ma[r, c].astype(np.float32)     # exact: f32 value times 0/1
        out[f"{t}:feat_mask_sigmoid"] = torch.sigmoid(mod.feat_mask).detach().numpy()
        out[f"{t}:final_mask_edges"] = mod.mask.detach().numpy()[r, c]
        out[f"{t}:loss"] = np.asarray(mod.loss_trace, np.float32)
        with quiet():   # reference post-processing of explain_nodes_gnn_stats (explain.py:306-308)
            G = io_utils.denoise_graph(ma, new_idx, ex.feat[0][nb], threshold_num=20)
        out[f"{t}:denoised_nodes"] = np.asarray(sorted(G.nodes()), np.int32)
        out[f"{t}:denoised_edges"] = np.asarray(sorted((min(u, v), max(u, v)) for u, v in G.edges()), np.int32).reshape(-1, 2)
        if t in keep_mask0:
            out[f"{t}:mask0"] = mod.mask0.numpy()
        print(f"  {dataset} target {t}: n={len(nb)} edges={len(r)//2} loss[0]={mod.loss_trace[0]:.4f} "
              f"loss[-1]={mod.loss_trace[-1]:.4f}")
    np.savez_compressed(os.path.join(HERE, dataset + "_explain.npz"), **out)


def molecule_like(rng, max_nodes=100, num_feat=14):
    """Random tree + a few ring closures, 10..100 nodes, one-hot node labels, padded to max_nodes."""
    n = int(rng.integers(10, max_nodes + 1))
    A = np.zeros((max_nodes, max_nodes), np.float32)
    for v in range(1, n):
        u = int(rng.integers(max(0, v - 4), v))
        A[u, v] = A[v, u] = 1
    for _ in range(max(1, n // 8)):
        u, v = rng.integers(0, n, 2)
        if u != v:
            A[u, v] = A[v, u] = 1
    X = np.zeros((max_nodes, num_feat), np.float32)
    X[np.arange(n), rng.integers(0, num_feat, n)] = 1
    return A, X, n


def graph_fixture(work, num_graphs=6, epochs=100):
    import models
    from explainer import explain
    args = explain_args("syn1", work, epochs)
    args.bmname = "Mutagenicity"
    args.graph_mode = True
    os.makedirs(args.logdir, exist_ok=True)
    rng = np.random.default_rng(0)
    torch.manual_seed(0)
    model = models.GcnEncoderGraph(input_dim=14, hidden_dim=20, embedding_dim=20, label_dim=2, num_layers=3,
                                   bn=False, args=args)
    with torch.no_grad():                                       # non-zero conv biases so padded rows matter
        for k, v in model.state_dict().items():
            if k.endswith("bias"):
                v.normal_(0, 0.1)
    graphs = [molecule_like(rng) for _ in range(num_graphs)]
    adj = torch.tensor(np.stack([g[0] for g in graphs]))
    feat = torch.tensor(np.stack([g[1] for g in graphs]))
    label = torch.tensor(rng.integers(0, 2, num_graphs), dtype=torch.long)
    model.eval()
    with torch.no_grad():
        pred = model(feat, adj)[0].numpy()[None]                # cg["pred"] layout [1, G, C] (train.py:255-257)
    built = capture_module(explain)
    ex = explain.Explainer(model=model, adj=adj, feat=feat, label=label, pred=pred, train_idx=None, args=args,
                           writer=None, print_training=False, graph_mode=True, graph_idx=0)
    out = dict(epochs=np.int64(epochs), adj=adj.numpy(), feat=feat.numpy(), label=label.numpy(), pred=pred[0],
               num_nodes=np.asarray([g[2] for g in graphs], np.int64))
    for k, v in model.state_dict().items():
        out["w:" + k] = v.detach().numpy().astype(np.float32)
    for g in range(num_graphs):
        torch.manual_seed(1000 + g)
        with quiet():
            ma = ex.explain(node_idx=0, graph_idx=g, graph_mode=True)
        mod = built[-1]
        assert not np.isnan(ma).any()
        out[f"{g}:masked_adj"] = ma.astype(np.float32)
        out[f"{g}:feat_mask_sigmoid"] = torch.sigmoid(mod.feat_mask).detach().numpy()
        out[f"{g}:loss"] = np.asarray(mod.loss_trace, np.float32)
        print(f"  graph {g}: nodes={graphs[g][2]} loss[0]={mod.loss_trace[0]:.4f} loss[-1]={mod.loss_trace[-1]:.4f}")
    np.savez_compressed(os.path.join(HERE, "graphmode_explain.npz"), **out)


def main():
    install_shims()
    work = tempfile.mkdtemp(prefix="gnnx_golden_")
    torch.set_num_threads(1)
    try:
        print("syn1: training with the reference train.py ...")
        mint_checkpoint("syn1", work)
        node_fixture("syn1", [302, 309, 330, 555, 400, 300], work, keep_mask0=(302, 309))
        print("syn4: training with the reference train.py ...")
        mint_checkpoint("syn4", work)
        node_fixture("syn4", [511, 520, 700, 870], work, keep_mask0=(511,))
        print("syn5: training with the reference train.py ...")
        mint_checkpoint("syn5", work)
        node_fixture("syn5", [511, 515, 1000, 1230], work)
        print("graph mode ...")
        graph_fixture(work)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
