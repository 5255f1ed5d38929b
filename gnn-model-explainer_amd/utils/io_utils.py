"""Checkpoint / file-name helpers of the explainer boundary (host side, not the hot path).

Same naming rules and on-disk formats as the reference so checkpoints and `.npy` outputs interchange:
  gen_prefix / gen_explainer_prefix   utils/io_utils.py:37-60
  create_filename                     utils/io_utils.py:63-78
  load_ckpt                           utils/io_utils.py:106-125 (torch >= 2.6 needs weights_only=False: the
                                      reference pickles the optimizer object and numpy arrays)
"""
import os

import numpy as np
import torch


def gen_prefix(args):
    name = args.bmname if getattr(args, "bmname", None) is not None else args.dataset
    name += "_" + args.method
    name += "_h" + str(args.hidden_dim) + "_o" + str(args.output_dim)
    if not args.bias:
        name += "_nobias"
    if len(args.name_suffix) > 0:
        name += "_" + args.name_suffix
    return name


def gen_explainer_prefix(args):
    name = gen_prefix(args) + "_explain"
    if len(args.explainer_suffix) > 0:
        name += "_" + args.explainer_suffix
    return name


def create_filename(save_dir, args, isbest=False, num_epochs=-1):
    filename = os.path.join(save_dir, gen_prefix(args))
    os.makedirs(filename, exist_ok=True)
    if isbest:
        filename = os.path.join(filename, "best")
    elif num_epochs > 0:
        filename = os.path.join(filename, str(num_epochs))
    return filename + ".pth.tar"


def load_ckpt(args, isbest=False):
    filename = create_filename(args.ckptdir, args, isbest)
    if not os.path.isfile(filename):
        print("Checkpoint does not exist: {}".format(filename))
        print("Train one with the reference:  python train.py --dataset=DATASET_NAME")
        raise Exception("File not found.")          # same error behaviour as the reference (io_utils.py:124)
    return torch.load(filename, map_location="cpu", weights_only=False)


def denoise_graph(adj, node_idx, feat=None, label=None, threshold=None, threshold_num=None, max_component=True):
    """Threshold an explanation into a small networkx graph — the reference's post-processing step
    (utils/io_utils.py:193-245, used by explain_nodes_gnn_stats / explain_graphs, explain.py:306-308, 364-370).

    threshold_num keeps the `threshold_num` heaviest undirected edges (the symmetric matrix stores each twice);
    max_component keeps the largest connected component, otherwise isolated nodes are dropped."""
    import networkx as nx
    adj = np.asarray(adj)
    num_nodes = adj.shape[-1]
    G = nx.Graph()
    G.add_nodes_from(range(num_nodes))
    G.nodes[node_idx]["self"] = 1
    if feat is not None:
        for node in G.nodes():
            G.nodes[node]["feat"] = feat[node]
    if label is not None:
        for node in G.nodes():
            G.nodes[node]["label"] = label[node]
    if threshold_num is not None:
        pos = adj[adj > 0]
        k = min(len(pos), threshold_num * 2)
        threshold = np.sort(pos)[-k]
    if threshold is not None:
        rows, cols = np.nonzero(adj >= threshold)
    else:
        rows, cols = np.nonzero(adj > 1e-6)
    G.add_weighted_edges_from((int(i), int(j), adj[i, j]) for i, j in zip(rows, cols))
    if max_component:
        G = G.subgraph(max(nx.connected_components(G), key=len)).copy()
    else:
        G.remove_nodes_from(list(nx.isolates(G)))
    return G
