"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, on torch-autograd fp32, of the reference's per-target mask
optimisation loop.  It performs the same tensor operations in the same order as
the reference so that, on the same torch build and the same seed, its output is
bit-identical to `/root/reference` (checked by tests/test_oracle_golden.py
against fixtures produced by tests/golden/make_golden.py, which imports the real
reference).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.

Reference lines restated here (paths relative to /root/reference):
  * explainer/explain.py:645-663   edge-mask init  N(1, sqrt(2/n)) from the CPU generator
  * explainer/explain.py:633-643   feature-mask init (constant 0)
  * explainer/explain.py:665-678   masked adjacency  adj * (s + s^T)/2 * (1 - I)
  * explainer/explain.py:685-715   forward under the mask, softmax of the target row
  * explainer/explain.py:740-820   loss = pred + size + lap + mask_ent + feat_size
  * explainer/explain.py:137-146   the epoch loop (zero_grad, forward, loss, backward, step)
  * explainer/explain.py:208-211   result = masked_adj of the LAST forward * sub_adj
  * models.py:58-80                GraphConv: normalize((adj @ x) @ W + b)
  * models.py:230-267, 363-376     node encoder: 3 convs, ReLU on first two, concat, Linear
  * models.py:269-316              graph encoder: per-layer max over rows, concat, Linear
  * utils/train_utils.py:7-23      Adam(lr=args.lr) with torch defaults
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

COEFF_SIZE = 0.005       # explain.py:625
COEFF_FEAT_SIZE = 1.0    # explain.py:626
COEFF_ENT = 1.0          # explain.py:627
COEFF_LAP = 1.0          # explain.py:630


def init_edge_mask(n, generator=None):
    """explain.py:645-652: one normal_ draw of n*n values, mean 1, std gain('relu')*sqrt(2/(n+n))."""
    std = math.sqrt(2.0) * math.sqrt(2.0 / (n + n))
    m = torch.empty(n, n, dtype=torch.float32)
    m.normal_(1.0, std, generator=generator)
    return m


def graph_conv(x, adj, w, b):
    """models.py:70-78 with normalize_embedding=True, no att/add_self/dropout."""
    y = torch.matmul(adj, x)
    y = torch.matmul(y, w)
    y = y + b
    return F.normalize(y, p=2, dim=2)


def apply_bn(x):
    """models.py:222-228: a FRESH nn.BatchNorm1d(num_nodes) (training mode, weight 1, bias 0) applied to [1, n, d]: every
    node is a channel, standardised over its d features with the biased variance, eps 1e-5."""
    return F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)


def encoder_node(x, adj, wts, bn=False):
    """models.py:230-267 + 375: returns logits for every row, [1, n, C]."""
    h1 = torch.relu(graph_conv(x, adj, wts["conv_first.weight"], wts["conv_first.bias"]))
    if bn:
        h1 = apply_bn(h1)
    h2 = torch.relu(graph_conv(h1, adj, wts["conv_block.0.weight"], wts["conv_block.0.bias"]))
    if bn:
        h2 = apply_bn(h2)
    h3 = graph_conv(h2, adj, wts["conv_last.weight"], wts["conv_last.bias"])
    emb = torch.cat([h1, h2, h3], dim=2)
    return F.linear(emb, wts["pred_model.weight"], wts["pred_model.bias"])


def encoder_graph(x, adj, wts, bn=False):
    """models.py:269-316 (num_aggs=1, concat): logits [1, C]."""
    h1 = torch.relu(graph_conv(x, adj, wts["conv_first.weight"], wts["conv_first.bias"]))
    if bn:
        h1 = apply_bn(h1)
    o1, _ = torch.max(h1, dim=1)
    h2 = torch.relu(graph_conv(h1, adj, wts["conv_block.0.weight"], wts["conv_block.0.bias"]))
    if bn:
        h2 = apply_bn(h2)
    o2, _ = torch.max(h2, dim=1)
    h3 = graph_conv(h2, adj, wts["conv_last.weight"], wts["conv_last.bias"])
    o3, _ = torch.max(h3, dim=1)
    out = torch.cat([o1, o2, o3], dim=1)
    return F.linear(out, wts["pred_model.weight"], wts["pred_model.bias"])


class MaskOptimOracle:
    """One target's optimisation, reference order of operations.

    adj [n,n], x [n,D] float32 tensors; wts: dict of float32 tensors keyed like the
    reference state_dict; gt_label: int (node: label of the target row; graph: graph
    label); pred_label: int array [n] (node mode; used only by the Laplacian term);
    node_idx: row of the target in the subgraph; graph_mode: bool.
    """

    def __init__(self, adj, x, wts, gt_label, pred_label, node_idx, graph_mode=False,
                 lr=0.1, mask0=None, mask_act="sigmoid", bn=False):
        self.bn = bool(bn)                # args.bn (models.py:222-228, 241-253)
        self.mask_act = mask_act          # explain.py:603, 667-670, 757-760: "sigmoid" or "ReLU"
        self.adj = adj.reshape(1, *adj.shape).float()
        self.x = x.reshape(1, *x.shape).float()
        self.wts = {k: v.float() for k, v in wts.items()}
        self.graph_mode = graph_mode
        self.gt_label = int(gt_label)
        self.node_idx = int(node_idx)
        n = adj.shape[0]
        self.n = n
        self.mask = torch.nn.Parameter(init_edge_mask(n) if mask0 is None else mask0.clone().float())
        self.feat_mask = torch.nn.Parameter(torch.zeros(x.shape[-1], dtype=torch.float32))
        self.diag_mask = torch.ones(n, n) - torch.eye(n)
        self.opt = torch.optim.Adam([self.mask, self.feat_mask], lr=lr)
        if not graph_mode:
            self.pred_label_t = torch.tensor(np.asarray(pred_label), dtype=torch.float)
        self.masked_adj = None

    def _act(self):
        return torch.sigmoid(self.mask) if self.mask_act == "sigmoid" else torch.relu(self.mask)

    def _masked_adj(self):
        s = self._act()
        s = (s + s.t()) / 2
        return (self.adj * s) * self.diag_mask

    def forward(self):
        self.masked_adj = self._masked_adj()
        x = self.x * torch.sigmoid(self.feat_mask)
        if self.graph_mode:
            logits = encoder_graph(x, self.masked_adj, self.wts, self.bn)
            return torch.softmax(logits[0], dim=0)
        logits = encoder_node(x, self.masked_adj, self.wts, self.bn)
        return torch.softmax(logits[-1, self.node_idx, :], dim=0)

    def loss(self, pred):
        pred_loss = -torch.log(pred[self.gt_label])
        m = self._act()
        size_loss = COEFF_SIZE * torch.sum(m)
        fm = torch.sigmoid(self.feat_mask)
        feat_size_loss = COEFF_FEAT_SIZE * torch.mean(fm)
        ent = -m * torch.log(m) - (1 - m) * torch.log(1 - m)
        ent_loss = COEFF_ENT * torch.mean(ent)
        if self.graph_mode:
            lap_loss = 0
        else:
            deg = torch.diag(torch.sum(self.masked_adj[0], 0))
            lap = deg - self.masked_adj[-1]
            lap_loss = COEFF_LAP * (self.pred_label_t @ lap @ self.pred_label_t) / self.adj.numel()
        total = pred_loss + size_loss + lap_loss + ent_loss + feat_size_loss
        terms = tuple(float(v.detach()) if torch.is_tensor(v) else float(v)
                      for v in (pred_loss, size_loss, lap_loss, ent_loss, feat_size_loss))
        return total, terms

    def run(self, num_epochs, record=False):
        """explain.py:137-146, 208-211. Returns masked_adj [n,n] float64 (mask of the last forward x adj)."""
        trace = []
        for _ in range(num_epochs):
            self.opt.zero_grad()
            pred = self.forward()
            loss, terms = self.loss(pred)
            loss.backward()
            self.opt.step()
            if record:
                trace.append((float(loss.detach()),) + terms)
        out = self.masked_adj[0].detach().numpy() * self.adj[0].numpy().astype(np.float64)
        self.trace = np.asarray(trace, dtype=np.float64)
        return out
