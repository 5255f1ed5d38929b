"""The encoders of the reference with its class names, constructor arguments and state_dict keys (models.py:8-80 GraphConv,
:83-316 GcnEncoderGraph, :330-380 GcnEncoderNode), so that `explainer_main.py` can rebuild an encoder and `load_state_dict` a
reference checkpoint without importing the reference.

The HIP engine reads only the frozen weights of the default configuration (3 layers, bias, method "base": `model.state_dict()`).
`forward` is a plain torch restatement of the reference's forward: predictions outside the hot path, and the model the torch
route of the explainer (explainer/torch_route.py) differentiates through for the configurations the kernels do not implement -
`method="att"` (models.py:62-68: adj * (x W_att)(x W_att)^T per layer), any number of layers, `add_self`, dropout.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class GraphConv(nn.Module):
    """y = normalize((adj' x) W (+ x W_self) + b), adj' = adj * att with att = (x W_att)(x W_att)^T when `att` (models.py:58-80)."""

    def __init__(self, input_dim, output_dim, add_self=False, normalize_embedding=False, dropout=0.0, bias=True, gpu=True, att=False):
        super().__init__()
        self.att, self.add_self, self.dropout, self.normalize_embedding = att, add_self, dropout, normalize_embedding
        self.input_dim, self.output_dim = input_dim, output_dim
        if dropout > 0.001:
            self.dropout_layer = nn.Dropout(p=dropout)
        self.weight = nn.Parameter(torch.empty(input_dim, output_dim))
        if add_self:
            self.self_weight = nn.Parameter(torch.empty(input_dim, output_dim))
        if att:
            self.att_weight = nn.Parameter(torch.empty(input_dim, input_dim))
        self.bias = nn.Parameter(torch.zeros(output_dim)) if bias else None

    def forward(self, x, adj):
        if self.dropout > 0.001:
            x = self.dropout_layer(x)
        if self.att:
            x_att = torch.matmul(x, self.att_weight)
            adj = adj * (x_att @ x_att.permute(0, 2, 1))
        y = torch.matmul(torch.matmul(adj, x), self.weight)
        if self.add_self:
            y = y + torch.matmul(x, self.self_weight)
        if self.bias is not None:
            y = y + self.bias
        if self.normalize_embedding:
            y = F.normalize(y, p=2, dim=2)
        return y, adj


class GcnEncoderGraph(nn.Module):
    def __init__(self, input_dim, hidden_dim, embedding_dim, label_dim, num_layers, pred_hidden_dims=(), concat=True,
                 bn=True, dropout=0.0, add_self=False, args=None):
        super().__init__()
        self.concat, self.bn, self.num_layers, self.num_aggs = concat, bool(bn), num_layers, 1
        self.bias = True if args is None else getattr(args, "bias", True)
        self.att = args is not None and getattr(args, "method", "base") == "att"
        gc = lambda i, o, dp=0.0: GraphConv(i, o, add_self=add_self, normalize_embedding=True, dropout=dp, bias=self.bias, att=self.att)
        self.conv_first = gc(input_dim, hidden_dim)
        self.conv_block = nn.ModuleList([gc(hidden_dim, hidden_dim, dropout) for _ in range(num_layers - 2)])
        self.conv_last = gc(hidden_dim, embedding_dim)
        self.act = nn.ReLU()
        self.label_dim = label_dim
        self.pred_input_dim = hidden_dim * (num_layers - 1) + embedding_dim if concat else embedding_dim
        if len(pred_hidden_dims) == 0:
            self.pred_model = nn.Linear(self.pred_input_dim, label_dim)
        else:
            layers, d = [], self.pred_input_dim
            for h in pred_hidden_dims:
                layers += [nn.Linear(d, h), self.act]
                d = h
            layers.append(nn.Linear(d, label_dim))
            self.pred_model = nn.Sequential(*layers)
        gain = nn.init.calculate_gain("relu")
        for m in self.modules():
            if isinstance(m, GraphConv):
                nn.init.xavier_uniform_(m.weight.data, gain=gain)
                if m.att:
                    nn.init.xavier_uniform_(m.att_weight.data, gain=gain)
                if m.add_self:
                    nn.init.xavier_uniform_(m.self_weight.data, gain=gain)

    def apply_bn(self, x):
        """models.py:222-228: a fresh BatchNorm1d(num_nodes) in training mode - every node standardised over its features."""
        return F.batch_norm(x, None, None, None, None, True, 0.1, 1e-5)

    def gcn_forward(self, x, adj):
        """models.py:230-267: all layers, concatenated embeddings [1, n, H (L - 1) + O] and the stacked attention adjacencies (the
        block layers re-append conv_first's, as the reference does at :252)."""
        x, adj_att = self.conv_first(x, adj)
        x = self.act(x)
        if self.bn:
            x = self.apply_bn(x)
        x_all, att_all = [x], [adj_att]
        for conv in self.conv_block:
            x, _ = conv(x, adj)
            x = self.act(x)
            if self.bn:
                x = self.apply_bn(x)
            x_all.append(x)
            att_all.append(adj_att)
        x, adj_att = self.conv_last(x, adj)
        x_all.append(x)
        att_all.append(adj_att)
        return torch.cat(x_all, dim=2), torch.stack(att_all, dim=3)

    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        """models.py:269-316: per-layer max over all rows, concat, linear head."""
        x, adj_att = self.conv_first(x, adj)
        x = self.act(x)
        if self.bn:
            x = self.apply_bn(x)
        outs, atts = [x.max(dim=1)[0]], [adj_att]
        for conv in self.conv_block:
            x, adj_att = conv(x, adj)
            x = self.act(x)
            if self.bn:
                x = self.apply_bn(x)
            outs.append(x.max(dim=1)[0])
            atts.append(adj_att)
        x, adj_att = self.conv_last(x, adj)
        atts.append(adj_att)
        outs.append(x.max(dim=1)[0])
        output = torch.cat(outs, dim=1) if self.concat else outs[-1]
        return self.pred_model(output), torch.stack(atts, dim=3)


class GcnEncoderNode(GcnEncoderGraph):
    def forward(self, x, adj, batch_num_nodes=None, **kwargs):
        """models.py:363-376: the head on every row."""
        emb, adj_att = self.gcn_forward(x, adj)
        return self.pred_model(emb), adj_att
