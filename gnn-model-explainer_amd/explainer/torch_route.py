"""The PyTorch-ROCm route for the configurations the HIP kernels do not implement ("fall back to a
PyTorch-ROCm restatement of the reference path rather than silently differ"): `model="att"` (the attention baseline),
`num_gc_layers != 3`, encoders with `add_self` / dropout / a hidden prediction head, node explanations on a graph-mode Explainer,
`method="att"` together with `--bn` / a ReLU mask / loss logging / unconstrained (`method="att"` itself runs on k_att), and
`ExplainModule.forward(marginalize=True / mask_features=False)`.

It is the reference's algorithm - ExplainModule (explainer/explain.py:582-820) and the loop of Explainer.explain (:137-146,
:200-211) - written with torch tensors on the HIP device and autograd through the CALLER's model (`model(x, masked_adj)`), one
target at a time.  It is NOT the accelerated path: it exists so that every flag the reference's CLI accepts keeps working after
the import swap, and it says so once per process (warnings).  On a CPU-only host it refuses to run unless a test points the
engine at the CPU on purpose (`explain._ENGINE["device"] = "cpu"`), like the kernels.
"""
import warnings

import numpy as np
import torch
import torch.nn as nn

_warned = set()


def _note(reason):
    if reason not in _warned:
        _warned.add(reason)
        warnings.warn("gnn_model_explainer_amd: %s is not implemented by the HIP kernels - this explanation runs on the PyTorch-ROCm "
                      "route (torch autograd on the GPU, one target at a time), not on the accelerated path" % reason, RuntimeWarning, stacklevel=3)


def device_for(engine_device):
    if engine_device is not None:
        return torch.device(engine_device)
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the explainer runs on MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class TorchExplainModule(nn.Module):
    """explain.py:582-820 on `device`.  mask0: the initial mask drawn by the caller (CPU generator, explain.py:645-652)."""

    def __init__(self, adj, x, model, label, args, coeffs, mask0, graph_idx=0, graph_mode=False, device=None):
        super().__init__()
        self.adj, self.x, self.label = adj.to(device), x.to(device), label
        self.model, self.args, self.graph_idx, self.graph_mode = model.to(device), args, graph_idx, graph_mode
        self.mask_act = getattr(args, "mask_act", "sigmoid")
        self.mask = nn.Parameter(torch.as_tensor(mask0, dtype=torch.float32).clone().to(device))
        self.feat_mask = nn.Parameter(torch.zeros(x.size(-1), device=device))
        n = adj.size()[1]
        self.mask_bias = nn.Parameter(torch.zeros(n, n, device=device)) if getattr(args, "mask_bias", False) else None
        self.diag_mask = (torch.ones(n, n) - torch.eye(n)).to(device)
        self.coeffs = coeffs
        self.masked_adj = None

    def _masked_adj(self):                                   # explain.py:665-678
        s = self.mask
        if self.mask_act == "sigmoid":
            s = torch.sigmoid(self.mask)
        elif self.mask_act == "ReLU":
            s = torch.relu(self.mask)
        s = (s + s.t()) / 2
        ma = self.adj * s
        if self.mask_bias is not None:
            b = (self.mask_bias + self.mask_bias.t()) / 2
            b = nn.functional.relu6(b * 6) / 6
            ma = ma + (b + b.t()) / 2
        return ma * self.diag_mask

    def forward(self, node_idx, unconstrained=False, mask_features=True, marginalize=False):   # explain.py:685-715
        x = self.x
        if unconstrained:
            s = torch.sigmoid(self.mask)
            self.masked_adj = torch.unsqueeze((s + s.t()) / 2, 0) * self.diag_mask
        else:
            self.masked_adj = self._masked_adj()
            if mask_features:
                fm = torch.sigmoid(self.feat_mask)
                if marginalize:
                    z = torch.normal(mean=torch.zeros_like(x) - x, std=torch.ones_like(x) / 2)
                    x = x + z * (1 - fm)
                else:
                    x = x * fm
        ypred, adj_att = self.model(x, self.masked_adj)
        node_pred = ypred[0] if self.graph_mode else ypred[self.graph_idx, node_idx, :]
        return torch.softmax(node_pred, dim=0), adj_att

    def loss(self, pred, pred_label, node_idx):              # explain.py:740-808
        gt = self.label if self.graph_mode else self.label[0][node_idx]
        out = -torch.log(pred[int(gt)])
        m = torch.sigmoid(self.mask) if self.mask_act == "sigmoid" else torch.relu(self.mask) if self.mask_act == "ReLU" else self.mask
        out = out + self.coeffs["size"] * torch.sum(m)
        out = out + self.coeffs["feat_size"] * torch.mean(torch.sigmoid(self.feat_mask))
        out = out + self.coeffs["ent"] * torch.mean(-m * torch.log(m) - (1 - m) * torch.log(1 - m))
        if not self.graph_mode:
            y = torch.tensor(np.asarray(pred_label), dtype=torch.float, device=self.mask.device)
            ma = self.masked_adj[0] if self.masked_adj.dim() == 3 else self.masked_adj
            lap = torch.diag(torch.sum(ma, 0)) - ma
            out = out + self.coeffs["lap"] * (y @ lap @ y) / self.adj.numel()
        return out


def explain_one(model, sub_adj, sub_feat, sub_label, pred_label, node_idx_new, args, coeffs, make_optimizer, mask0, graph_idx=0,
                graph_mode=False, unconstrained=False, kind="exp", device=None, reason="this configuration"):
    """The body of Explainer.explain (explain.py:94-146, 200-211) for one target -> float64 [n, n] masked adjacency * sub_adj."""
    _note(reason)
    adj = torch.tensor(np.asarray(sub_adj)[None], dtype=torch.float)
    x = torch.tensor(np.asarray(sub_feat)[None], dtype=torch.float)
    label = torch.as_tensor(np.asarray(sub_label)) if graph_mode else torch.as_tensor(np.asarray(sub_label)[None], dtype=torch.long)
    was_training = model.training
    mod = TorchExplainModule(adj, x, model, label, args, coeffs, mask0, graph_idx, graph_mode, device)
    params = [mod.mask, mod.feat_mask] + ([mod.mask_bias] if mod.mask_bias is not None else [])
    sched, opt = make_optimizer(args, params)
    model.eval()
    mod.train()                      # explain.py:135: this flips the wrapped model to train mode too (dropout)
    adj_atts = None
    for _ in range(int(args.num_epochs)):
        mod.zero_grad()
        opt.zero_grad()
        ypred, adj_atts = mod(node_idx_new, unconstrained=unconstrained)
        loss = mod.loss(ypred, pred_label, node_idx_new)
        loss.backward()
        opt.step()
        if sched is not None:
            sched.step()
        if kind != "exp":            # explain.py:200-201: the attention baseline only needs one forward
            break
    model.train(was_training)
    sub = np.asarray(sub_adj, np.float64)
    if kind == "exp":
        out = mod.masked_adj[0].detach().cpu().numpy() * sub
    else:                            # explain.py:207-208
        out = torch.sigmoid(adj_atts).squeeze().detach().cpu().numpy() * sub
    return out, torch.sigmoid(mod.feat_mask).detach().cpu().numpy()
