"""Host-side synthetic inputs for the benchmark configurations (setup only, not the hot path).

`ba_house` restates the reference's BA-House construction so the 100k-node configuration of BASELINE.json
(configs[4]) can be generated without the reference:
  gengraph.py:106-138 gen_syn1 (BA basis + 'house' motifs + 1 % random edges, gengraph.py:28-50 perturb),
  utils/synthetic_structsim.py:155-175 ba, :178-204 house, :263-343 build_graph (regular plug-in spacing,
  roles: basis 0, house 1/1/2/2/3).
syn1 itself is NOT regenerated here: parity fixtures come from the reference's own generator + RNG stream.
`sparse_gcn_predict` evaluates the frozen encoder on the full graph with sparse products to obtain the
predicted labels the Laplacian term needs (models.py:58-80, 230-267, 375 restated on scipy.sparse).
"""
import math

import networkx as nx
import numpy as np
import scipy.sparse as sp


def ba_house(width_basis=300, nb_shapes=80, m=5, seed=0, perturb=0.01):
    """-> (num_nodes, edges [E,2] int64 with u < v, labels [N] int64)."""
    rng = np.random.RandomState(seed)
    g = nx.barabasi_albert_graph(width_basis, m, seed=seed)
    edges = [(min(u, v), max(u, v)) for u, v in g.edges()]
    labels = [0] * width_basis
    spacing = math.floor(width_basis / nb_shapes)
    start = width_basis
    for k in range(nb_shapes):
        s = start
        edges += [(s, s + 1), (s + 1, s + 2), (s + 2, s + 3), (s, s + 3), (s, s + 4), (s + 1, s + 4)]
        edges.append((int(k * spacing), s))
        labels += [1, 1, 2, 2, 3]
        start += 5
    n = start
    eset = set(edges)
    for _ in range(int(len(eset) * perturb)):
        while True:
            u, v = int(rng.randint(0, n)), int(rng.randint(0, n))
            if u != v and (min(u, v), max(u, v)) not in eset:
                break
        eset.add((min(u, v), max(u, v)))
    e = np.asarray(sorted(eset), np.int64)
    return n, e, np.asarray(labels, np.int64)


def csr_from_edges(n, edges):
    e = np.asarray(edges)
    data = np.ones(len(e) * 2, np.float32)
    a = sp.csr_matrix((data, (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n))
    a.sum_duplicates()
    return a


def sparse_gcn_predict(csr, feat, sd):
    """Logits [N, C] of the 3-layer node encoder on the full (unmasked) graph."""
    f32 = np.float32
    x = np.asarray(feat, f32)
    hs = []
    for l, k in enumerate(("conv_first", "conv_block.0", "conv_last")):
        y = (csr @ x) @ np.asarray(sd[k + ".weight"], f32) + np.asarray(sd[k + ".bias"], f32)
        y = y / np.maximum(np.linalg.norm(y, axis=1, keepdims=True), 1e-12)
        x = np.maximum(y, 0) if l < 2 else y
        hs.append(x)
    return np.concatenate(hs, 1) @ np.asarray(sd["pred_model.weight"], f32).T + np.asarray(sd["pred_model.bias"], f32)


def molecule_like_graphs(num_graphs, seed=0, max_nodes=100, num_feat=14):
    """BASELINE.json configs[3] stand-in (the real Mutagenicity TU files are not available offline): molecule-like
    graphs - a random tree plus a few ring closures, 10..max_nodes atoms, one-hot atom types, binary class label -
    padded to max_nodes x max_nodes like the reference's GraphSampler (utils/graph_utils.py:11-145).
    -> (adj [G, max_nodes, max_nodes] f32, feat [G, max_nodes, num_feat] f32, num_nodes [G], label [G]); graph g is the
    same whatever num_graphs is (one generator, consumed in graph order)."""
    rng = np.random.default_rng(seed)
    A = np.zeros((num_graphs, max_nodes, max_nodes), np.float32)
    X = np.zeros((num_graphs, max_nodes, num_feat), np.float32)
    nn_, y = np.zeros(num_graphs, np.int64), np.zeros(num_graphs, np.int64)
    for g in range(num_graphs):
        n = int(rng.integers(10, max_nodes + 1))
        for v in range(1, n):
            u = int(rng.integers(max(0, v - 4), v))
            A[g, u, v] = A[g, v, u] = 1
        for _ in range(max(1, n // 8)):
            u, v = rng.integers(0, n, 2)
            if u != v:
                A[g, u, v] = A[g, v, u] = 1
        X[g, np.arange(n), rng.integers(0, num_feat, n)] = 1
        nn_[g] = n
        y[g] = int(rng.integers(0, 2))
    return A, X, nn_, y
