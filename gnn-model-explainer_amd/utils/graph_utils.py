"""Sparse k-hop neighbourhoods and sub-graph extraction (host side, setup of the hot path).

Reference behaviour reproduced (paths relative to the reference root):
  * utils/graph_utils.py:147-158 `neighborhoods`: hop = ((A + A^2 + ... + A^k) > 0) by DENSE matmul on the
    whole graph — O(N^3) time, O(N^2) memory (40 GB per matrix at N = 100k).  Here: the same set, row by
    row, from a CSR adjacency by walking k frontiers (O(size of the neighbourhood) per target).
    Note the reference semantics: a node is in its own neighbourhood only through a walk of length
    2..k back to itself (no explicit self term), so an isolated node has an EMPTY neighbourhood.
  * explainer/explain.py:492-501 `extract_neighborhood`: neighbours in ascending id order,
    node_idx_new = number of neighbours with a smaller id, dense sub-adjacency / features / labels.
"""
import numpy as np
import scipy.sparse as sp


class KHopIndex:
    """CSR adjacency + k-hop walk sets. adj: dense ndarray [N,N], scipy sparse, or (num_nodes, edges[E,2])."""

    def __init__(self, adj, n_hops):
        if isinstance(adj, tuple):
            n, e = adj
            e = np.asarray(e)
            data = np.ones(len(e) * 2, np.float32)
            csr = sp.csr_matrix((data, (np.r_[e[:, 0], e[:, 1]], np.r_[e[:, 1], e[:, 0]])), shape=(n, n))
            csr.sum_duplicates()
        elif sp.issparse(adj):
            csr = adj.tocsr()
        else:
            csr = sp.csr_matrix(np.asarray(adj))
        csr.eliminate_zeros()
        self.csr = csr
        self.n_hops = int(n_hops)
        self.num_nodes = csr.shape[0]

    def neighbors(self, node):
        """Ascending ids of {v : (A + ... + A^k)[node, v] > 0} (assumes non-negative weights, as the reference)."""
        indptr, indices = self.csr.indptr, self.csr.indices
        seen = np.zeros(0, np.int64)
        frontier = np.asarray([node], np.int64)
        for _ in range(self.n_hops):
            if frontier.size == 0:
                break
            nxt = np.unique(np.concatenate([indices[indptr[u]:indptr[u + 1]] for u in frontier])) \
                if frontier.size else frontier
            seen = np.union1d(seen, nxt)
            frontier = nxt          # walks, not shortest paths: revisiting is what puts `node` itself in the set
        return seen.astype(np.int64)

    def neighbors_batch(self, nodes):
        """k-hop walk sets of many targets at once: rows of S + S.A + ... + S.A^(k-1), S = A[nodes] (sparse
        products in C, no Python loop per hop).  -> list of ascending id arrays, same sets as `neighbors`."""
        nodes = np.asarray(nodes, np.int64)
        pattern = self.csr.astype(bool).astype(np.float32)
        walk = pattern[nodes]
        reach = walk.copy()
        for _ in range(self.n_hops - 1):
            walk = (walk @ pattern)
            walk.data[:] = 1.0          # keep counts from overflowing / growing
            reach = reach + walk
        reach = reach.tocsr()
        reach.sort_indices()
        return [reach.indices[reach.indptr[r]:reach.indptr[r + 1]].astype(np.int64) for r in range(len(nodes))]

    def sizes(self, nodes):
        return np.asarray([len(nb) for nb in self.neighbors_batch(nodes)], np.int64)

    def extract_batch(self, nodes):
        """-> list of (node_idx_new, sub_adj [n,n] float32, neighbors) for many targets (vectorised k-hop)."""
        out = []
        for v, nb in zip(nodes, self.neighbors_batch(nodes)):
            out.append((int(np.searchsorted(nb, v)), self.sub_adjacency(nb), nb))
        return out

    def sub_adjacency(self, nb):
        return np.asarray(self.csr[nb][:, nb].todense(), dtype=np.float32)

    def extract(self, node):
        """-> (node_idx_new, sub_adj [n,n] float32, neighbors [n]) with the reference's ordering rules."""
        nb = self.neighbors(node)
        node_idx_new = int(np.searchsorted(nb, node))      # == sum(row[:node]) of the reference
        return node_idx_new, self.sub_adjacency(nb), nb


def neighborhoods_dense(adj, n_hops):
    """Dense restatement of the reference function (small graphs / tests only)."""
    a = np.asarray(adj, np.float32)
    hop = power = a
    for _ in range(n_hops - 1):
        power = power @ a
        hop = ((hop + power) > 0).astype(np.float32)
    return hop.astype(int)
