"""explainer_main.py — same command line as the reference CLI (explainer_main.py:23-168 arg_parse, :171-313 main),
running the mask optimisation on the MI355X engine.

    python -m gnn_model_explainer_amd.explainer_main --dataset=syn1 --explain-node=300 --epochs=300
    python -m gnn_model_explainer_amd.explainer_main --dataset=syn1                 # nodes 400..695 step 5, batched
    python -m gnn_model_explainer_amd.explainer_main --bmname=Mutagenicity --graph-idx=1

Differences from the reference, all deliberate: no TensorBoard writer / plots (out of scope), `--gpu` is implied
(the engine has no CPU path), multi-target modes run as ONE batched GPU job, and unsupported options raise
NotImplementedError instead of silently taking another code path.
"""
import argparse
import os

from . import models
from .explainer import explain
from .utils import io_utils


def arg_parse(argv=None):
    parser = argparse.ArgumentParser(description="GNN Explainer arguments.")
    io_parser = parser.add_mutually_exclusive_group(required=False)
    io_parser.add_argument("--dataset", dest="dataset", help="Input dataset.")
    io_parser.add_argument("--bmname", dest="bmname", help="Name of the benchmark dataset")
    io_parser.add_argument("--pkl", dest="pkl_fname", help="Name of the pkl data file")
    # utils/parser_utils.py:7-23
    parser.add_argument("--opt", dest="opt", type=str, help="Type of optimizer")
    parser.add_argument("--opt-scheduler", dest="opt_scheduler", type=str, help="Type of optimizer scheduler.")
    parser.add_argument("--opt-restart", dest="opt_restart", type=int)
    parser.add_argument("--opt-decay-step", dest="opt_decay_step", type=int)
    parser.add_argument("--opt-decay-rate", dest="opt_decay_rate", type=float)
    parser.add_argument("--lr", dest="lr", type=float, help="Learning rate.")
    parser.add_argument("--clip", dest="clip", type=float, help="Gradient clipping.")
    parser.add_argument("--clean-log", action="store_true")
    parser.add_argument("--logdir", dest="logdir", help="Log directory (masked_adj_*.npy files)")
    parser.add_argument("--ckptdir", dest="ckptdir", help="Model checkpoint directory")
    parser.add_argument("--cuda", dest="cuda", help="Device index.")
    parser.add_argument("--gpu", dest="gpu", action="store_const", const=True, default=True)
    parser.add_argument("--epochs", dest="num_epochs", type=int, help="Number of mask-optimisation epochs.")
    parser.add_argument("--hidden-dim", dest="hidden_dim", type=int)
    parser.add_argument("--output-dim", dest="output_dim", type=int)
    parser.add_argument("--num-gc-layers", dest="num_gc_layers", type=int)
    parser.add_argument("--bn", dest="bn", action="store_const", const=True, default=False)
    parser.add_argument("--dropout", dest="dropout", type=float)
    parser.add_argument("--nobias", dest="bias", action="store_const", const=False, default=True)
    parser.add_argument("--no-writer", dest="writer", action="store_const", const=False, default=True)
    parser.add_argument("--mask-act", dest="mask_act", type=str, help="sigmoid, ReLU.")
    parser.add_argument("--mask-bias", dest="mask_bias", action="store_const", const=True, default=False)
    parser.add_argument("--explain-node", dest="explain_node", type=int, help="Node to explain.")
    parser.add_argument("--graph-idx", dest="graph_idx", type=int, help="Graph to explain.")
    parser.add_argument("--graph-mode", dest="graph_mode", action="store_const", const=True, default=False)
    parser.add_argument("--multigraph-class", dest="multigraph_class", type=int)
    parser.add_argument("--multinode-class", dest="multinode_class", type=int)
    parser.add_argument("--align-steps", dest="align_steps", type=int)
    parser.add_argument("--method", dest="method", type=str)
    parser.add_argument("--name-suffix", dest="name_suffix")
    parser.add_argument("--explainer-suffix", dest="explainer_suffix")
    parser.set_defaults(                                   # explainer_main.py:143-167
        logdir="log", ckptdir="ckpt", dataset="syn1", opt="adam", opt_scheduler="none", cuda="0", lr=0.1, clip=2.0,
        batch_size=20, num_epochs=100, hidden_dim=20, output_dim=20, num_gc_layers=3, dropout=0.0, method="base",
        name_suffix="", explainer_suffix="", align_steps=1000, explain_node=None, graph_idx=-1, mask_act="sigmoid",
        multigraph_class=-1, multinode_class=-1)
    return parser.parse_args(argv)


def main(argv=None):
    prog_args = arg_parse(argv)
    os.makedirs(prog_args.logdir, exist_ok=True)
    ckpt = io_utils.load_ckpt(prog_args)
    cg_dict = ckpt["cg"]
    input_dim = cg_dict["feat"].shape[2]
    num_classes = cg_dict["pred"].shape[2]
    print("Loaded model from {}".format(prog_args.ckptdir))
    print("input dim: ", input_dim, "; num classes: ", num_classes)
    graph_mode = prog_args.graph_mode or prog_args.multigraph_class >= 0 or prog_args.graph_idx >= 0
    cls = models.GcnEncoderGraph if graph_mode else models.GcnEncoderNode
    model = cls(input_dim=input_dim, hidden_dim=prog_args.hidden_dim, embedding_dim=prog_args.output_dim,
                label_dim=num_classes, num_layers=prog_args.num_gc_layers, bn=prog_args.bn, args=prog_args)
    model.load_state_dict(ckpt["model_state"])
    explainer = explain.Explainer(model=model, adj=cg_dict["adj"], feat=cg_dict["feat"], label=cg_dict["label"],
                                  pred=cg_dict["pred"], train_idx=cg_dict["train_idx"], args=prog_args, writer=None,
                                  print_training=True, graph_mode=graph_mode, graph_idx=prog_args.graph_idx)
    if prog_args.explain_node is not None:
        return explainer.explain(prog_args.explain_node, unconstrained=False)
    if graph_mode:
        if prog_args.multigraph_class >= 0:
            labels = cg_dict["label"].numpy()
            graph_indices = []
            for i, l in enumerate(labels):
                if l == prog_args.multigraph_class:
                    graph_indices.append(i)
                if len(graph_indices) > 30:
                    break
            return explainer.explain_graphs(graph_indices=graph_indices)
        if prog_args.graph_idx == -1:
            return explainer.explain_graphs(graph_indices=[1, 2, 3, 4])
        return explainer.explain(node_idx=0, graph_idx=prog_args.graph_idx, graph_mode=True, unconstrained=False)
    if prog_args.multinode_class >= 0:
        labels = cg_dict["label"][0]
        node_indices = []
        for i, l in enumerate(labels):
            if len(node_indices) > 4:
                break
            if l == prog_args.multinode_class:
                node_indices.append(i)
        return explainer.explain_nodes(node_indices, prog_args)
    return explainer.explain_nodes_gnn_stats(range(400, 700, 5), prog_args)


if __name__ == "__main__":
    # this process exists to run the engine: eight HIP queues, CPU pool sized to the container's quota, one NUMA node - an explicit call
    # (a caller that imports the package, or calls main() from its own process, gets none of it)
    from . import tune_process
    tune_process()
    main()
