"""Parameter containers with the reference's class names and state_dict keys (models.py:83-380).

The HIP engine only needs the frozen weights (it reads `model.state_dict()`), so `explainer_main.py`
must be able to rebuild the encoder and `load_state_dict` a reference checkpoint without importing the
reference.  `forward` is a plain dense restatement used for predictions outside the hot path
(models.py:58-80 GraphConv, :230-267 gcn_forward, :269-316 graph head, :363-376 node head); only the
default configuration is provided: 3 layers, bias, normalize_embedding=True, concat, optional bn, no att/dropout.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GraphConv(nn.Module):
    def __init__(self, input_dim, output_dim, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(input_dim, output_dim))
        self.bias = nn.Parameter(torch.zeros(output_dim)) if bias else None
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, x, adj):
        y = torch.matmul(torch.matmul(adj, x), self.weight)
        if self.bias is not None:
            y = y + self.bias
        return F.normalize(y, p=2, dim=2)


class GcnEncoderGraph(nn.Module):
    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims=(), concat=True,
                 bn=False, dropout=0.0, add_self=False, args=None):
        super().__init__()
        if num_layers != 3 or len(pred_hidden_dims) or not concat or dropout > 0 or add_self:
            raise NotImplementedError("only the explainer_main.py default encoder (3 layers, concat) is provided")
        self.bn = bool(bn)
        if args is not None and getattr(args, "method", "base") == "att":
            raise NotImplementedError("method='att' is outside the accelerated path")
        bias = True if args is None else getattr(args, "bias", True)
        self.conv_first = GraphConv(input_dim, hidden_dim, bias)
        self.conv_block = nn.ModuleList([GraphConv(hidden_dim, hidden_dim, bias)])
        self.conv_last = GraphConv(hidden_dim, embedding_dim, bias)
        self.pred_model = nn.Linear(hidden_dim * 2 + embedding_dim, label_dim)
        self.att = False

    def apply_bn(self, x):
        """models.py:222-228: a fresh BatchNorm1d(num_nodes) in training mode - every node standardised over its features."""
        return F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)

    def _layers(self, x, adj):
        h1 = torch.relu(self.conv_first(x, adj))
        if self.bn:
            h1 = self.apply_bn(h1)
        h2 = torch.relu(self.conv_block[0](h1, adj))
        if self.bn:
            h2 = self.apply_bn(h2)
        return h1, h2, self.conv_last(h2, adj)

    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        hs = self._layers(x, adj)
        out = torch.cat([h.max(dim=1)[0] for h in hs], dim=1)
        return self.pred_model(out), None


class GcnEncoderNode(GcnEncoderGraph):
    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        return self.pred_model(torch.cat(self._layers(x, adj), dim=2)), None
