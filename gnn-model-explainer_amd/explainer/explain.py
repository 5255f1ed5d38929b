"""Drop-in `Explainer` / `ExplainModule` behind which the mask optimisation runs on the MI355X engine.

Same constructor and method signatures, argument meaning, return types, output files and error behaviour
as the reference's explainer/explain.py (Explainer :42-579, ExplainModule :582-820), for the
mask-optimisation path (`model="exp"`, `unconstrained=False`, sigmoid mask, Adam):

  * `explain()` returns the float64 [n, n] `masked_adj * sub_adj` of the LAST forward and writes
    `<logdir>/masked_adj_<prefix>node_idx_<i>graph_idx_<g>.npy` (explain.py:208-221);
  * `explain_nodes`, `explain_nodes_gnn_stats`, `explain_graphs` return `list[np.ndarray]` — but the list
    comprehension over targets (explain.py:234-236, 296-299, 362-363) is ONE batched GPU job;
  * the initial edge masks are drawn from the caller's global torch CPU generator, one `normal_` per target
    in list order, exactly like `construct_edge_mask` (explain.py:645-652), so seeded runs reproduce the
    reference's masks.

`--mask-bias` is accepted (the reference's bias mask provably stays exactly 0, see _check_supported); `mask_act="ReLU"` runs on
the dense streaming kernels and reproduces the reference's NaN behaviour; `--bn` runs on the dense streaming kernels.
`--opt sgd | rmsprop | adagrad`, `--opt-scheduler step | cos` and `unconstrained=True` run on the same kernels (round 3);
`--method att` (the attention encoder, models.py:62-68) runs on k_att (csrc/gnnx_att.hpp), node and graph mode.
Configurations the kernels do not implement - `model="att"` (the attention baseline), num_gc_layers != 3, encoders with add_self /
dropout / a hidden head, a node explanation on a graph-mode Explainer, `--method att` together with --bn / loss logging - run on
explainer/torch_route.py: the reference's algorithm as torch
autograd on the HIP device through the caller's model, announced by a RuntimeWarning; never a silent difference.  `model="grad"` (the gradient baseline, explain.py:125-133)
runs on the engine too (Explainer.explain_grad).
Plotting / TensorBoard / alignment post-processing of the reference is out of scope.
"""
import os
import time

import numpy as np
import torch
import torch.nn as nn

from ..engine import (FEAT_STRIDE, Hyper, MaskOptimJob, Subgraph, XLJob, device_graph, init_edge_mask, init_edge_masks_raw,
                      khop_device)
from ..utils import io_utils
from ..utils.graph_utils import KHopIndex

COEFFS = {"size": 0.005, "feat_size": 1.0, "ent": 1.0, "feat_ent": 0.1, "grad": 0, "lap": 1.0}  # explain.py:624-631
# sub-graphs of more nodes than this take the XL route in Explainer.explain_batch (engine.XLJob; GNNX_XL_MIN_N overrides).  The default keeps every
# reference-sized dataset on the LDS-resident classes and sends what they cannot hold to the CSR-native kernel instead of dense n x n blocks.
XL_MIN_N = int(os.environ.get("GNNX_XL_MIN_N", "512"))


def _check_supported(args):
    # mask_act: "sigmoid", or "ReLU" (explain.py:669-670, 757-760) on the dense streaming kernels.  Like the reference, ReLU
    # yields NaN masks whenever an initial mask entry lies outside (0, 1] (its entropy term takes log(1 - relu(M))) - i.e.
    # always with the reference's own N(1, .) initialisation (tests/golden/flags_explain.npz: 100 % NaN).
    # --mask-bias (explain.py:657-661, 674-677) is accepted: the reference creates mask_bias = 0 and adds
    # sym(ReLU6(6 sym(mask_bias)) / 6) to the masked adjacency.  ReLU6 has zero gradient at 0 (torch: hardtanh backward is
    # strict), so mask_bias never receives a gradient, Adam leaves it at exactly 0 and the added term is exactly 0 in every
    # epoch: the reference's outputs with and without the flag are bit-identical (pinned by tests/golden/flags_explain.npz,
    # produced by running the reference with mask_bias=True).  The same kernels therefore serve both settings.
    # --bn (apply_bn, models.py:222-228, 241-253) runs on the dense streaming kernels (forward and backward through the row-wise
    # standardisation; pinned to the reference by tests/golden/flags_explain.npz)
    # --opt adam | sgd | rmsprop | adagrad and --opt-scheduler none | step | cos (utils/train_utils.py:7-22) all run in the kernels'
    # per-entry update (gnnx_hyper.opt, .lr_schedule); anything else makes the reference's build_optimizer fail too
    if getattr(args, "opt", "adam") not in OPTIMIZER_EPS:
        raise ValueError("opt=%r: the reference's build_optimizer knows adam, sgd, rmsprop, adagrad" % args.opt)
    if (getattr(args, "opt_scheduler", "none") or "none") not in ("none", "step", "cos"):
        raise ValueError("opt_scheduler=%r: the reference's build_optimizer knows none, step, cos" % args.opt_scheduler)


OPTIMIZER_EPS = {"adam": 1e-8, "sgd": 0.0, "rmsprop": 1e-8, "adagrad": 1e-10}      # torch defaults of the optimisers train_utils.py:9-16 builds


def _torch_optimizer(args, params):
    """The optimiser / scheduler pair the reference's build_optimizer returns for `args` (utils/train_utils.py:7-22), built over
    `params`: the `.optimizer` / `.scheduler` surface of ExplainModule, and the source of the per-epoch learning rates."""
    import torch.optim as optim
    lr = float(args.lr)
    make = {"adam": lambda: optim.Adam(params, lr=lr, weight_decay=0.0), "sgd": lambda: optim.SGD(params, lr=lr, momentum=0.95, weight_decay=0.0),
            "rmsprop": lambda: optim.RMSprop(params, lr=lr, weight_decay=0.0), "adagrad": lambda: optim.Adagrad(params, lr=lr, weight_decay=0.0)}
    opt = make[getattr(args, "opt", "adam")]()
    kind = getattr(args, "opt_scheduler", "none") or "none"
    sched = None
    if kind == "step":
        sched = optim.lr_scheduler.StepLR(opt, step_size=args.opt_decay_step, gamma=args.opt_decay_rate)
    elif kind == "cos":
        sched = optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.opt_restart)
    return sched, opt


def _lr_schedule(args, num_iters):
    """Learning rate of every epoch under --opt-scheduler: torch's own scheduler stepped on a one-parameter dummy, once after every
    optimiser step as the reference's loop does (explain.py:144-146), so the doubles are the ones the reference's optimiser sees."""
    if (getattr(args, "opt_scheduler", "none") or "none") == "none":
        return None
    import warnings
    sched, opt = _torch_optimizer(args, [nn.Parameter(torch.zeros(1))])
    lrs = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(int(num_iters)):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
    return np.asarray(lrs, np.float64)


def _unmasked_features_state(job):
    """ExplainModule.forward(unconstrained=True) does NOT apply the feature mask (explain.py:688-707: `x * feat_mask` sits in the other
    branch): start the engine with feat_mask = 40 - sigmoid(40) == 1.0f exactly and its gradient factor phi (1 - phi) == 0, so the
    features stay unmasked for the whole run."""
    from ..engine import AdamState
    f = torch.zeros(job.T, 3, FEAT_STRIDE, dtype=torch.float32)
    f[:, 0, :job.D] = 40.0
    return AdamState(0, None, None, f.to(job.device))


def _regulariser_only_feat_mask(args, D, num_iters):
    """What the reference's feat_mask parameter does in an unconstrained run: only the size term mean(sigmoid(f)) reaches it
    (explain.py:763-766), a scalar recursion through the optimiser - evaluated with the very torch optimiser (host, D values)."""
    f = nn.Parameter(torch.zeros(D))
    sched, opt = _torch_optimizer(args, [f])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(int(num_iters)):
            opt.zero_grad()
            (COEFFS["feat_size"] * torch.mean(torch.sigmoid(f))).backward()
            opt.step()
            if sched is not None:
                sched.step()
    return f.detach().numpy()


def _torch_route_reason(args, model, state_dict=None, graph_mode=False, record_loss=False, unconstrained=False):
    """None when the HIP kernels implement this configuration, else what makes it take explainer/torch_route.py (SURVEY.md section 8(b):
    configurations the kernels do not cover run on a PyTorch-ROCm restatement of the reference path instead of silently differing).
    method="att" (models.py:62-68) runs on k_att (csrc/gnnx_att.hpp), node and graph mode, with the sigmoid mask; its other
    combinations (--bn, mask_act=ReLU, loss logging, unconstrained) take the PyTorch-ROCm route."""
    method = getattr(args, "method", "base")
    if method not in ("base", "att"):
        return "method=%r" % method
    if getattr(args, "mask_act", "sigmoid") not in ("sigmoid", "ReLU"):
        return "mask_act=%r" % args.mask_act
    sd = state_dict if state_dict is not None else model.state_dict()
    att = any(k.endswith("att_weight") for k in sd)
    if att or method == "att":
        for what, on in (("--bn", bool(getattr(args, "bn", False))), ("loss logging", record_loss),
                         ("mask_act=ReLU", getattr(args, "mask_act", "sigmoid") == "ReLU"), ("unconstrained", unconstrained),
                         ("attention weights in some layers only", not all(k + ".att_weight" in sd for k in ("conv_first", "conv_block.0", "conv_last")))):
            if on:
                return "method='att' (models.py:62-68) with %s" % what
    if unconstrained and record_loss:
        # the kernels run the unconstrained forward with the engine's feature mask pinned at sigma = 1 and recover the reference's
        # regulariser-only feature mask on the host afterwards: the masks are the reference's, the LOGGED feat_size term would not be
        return "unconstrained=True with loss logging (explain.py:688-691 with :808-819)"
    if unconstrained and getattr(args, "mask_act", "sigmoid") == "ReLU":
        return "unconstrained=True with mask_act=ReLU (the reference's unconstrained forward always applies the sigmoid, explain.py:689)"
    extra = [k for k in sd if k.startswith("conv_block.") and not k.startswith("conv_block.0.")]
    if extra or "conv_block.0.weight" not in sd:
        return "an encoder with %d graph-convolution layers (the kernels implement 3)" % (2 + len({k.split(".")[1] for k in sd if k.startswith("conv_block.")}))
    if any(k.endswith("self_weight") for k in sd):
        return "an encoder with add_self"
    if "pred_model.weight" not in sd or "conv_first.bias" not in sd:
        return "an encoder without bias or with a hidden prediction head"
    if sd["pred_model.weight"].shape[1] != sd["conv_first.weight"].shape[1] + sd["conv_block.0.weight"].shape[1] + sd["conv_last.weight"].shape[1]:
        return "an encoder without concatenated embeddings"
    if max(sd["conv_first.weight"].shape + sd["conv_last.weight"].shape) > FEAT_STRIDE or sd["pred_model.weight"].shape[0] > 32:
        return "feature / hidden / class widths beyond 32"
    if any(getattr(m, "dropout", 0.0) and m.dropout > 0.001 for m in model.modules()):
        return "dropout in the encoder (active during the optimisation: explain.py:135)"
    return None


def _hyper(args, **kw):
    opt = getattr(args, "opt", "adam")
    n = int(kw.pop("num_iters", args.num_epochs))
    return Hyper(num_iters=n, lr=float(args.lr), eps=OPTIMIZER_EPS[opt], opt=opt, lr_schedule=_lr_schedule(args, n),
                 c_size=COEFFS["size"], c_feat_size=COEFFS["feat_size"], c_ent=COEFFS["ent"], c_lap=COEFFS["lap"], **kw)


def _np(a):
    return a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)


# which build of the engine the classes below drive: None = libgnnx_hip.so on the current HIP device.  The CPU-only tests
# point this at the emulator build of the same sources (tests/emu) with host memory standing in for device memory.
_ENGINE = {"lib": None, "device": None}


class _Result:
    """What the most recent batch left behind besides the returned masks."""

    def __init__(self, feat_mask, loss, edges):
        self.feat_mask = feat_mask      # [T, D] final feature-mask parameters (the reference discards them, explain.py:108, 221)
        self.loss = loss                # [T, iters, 8] loss terms when logging was on
        self.edges = edges              # engine.EdgeMasks of the batch (binary adjacency)
        self.denoised = None            # (keep [E], threshold [T], stats [T, 3]) of engine.MaskOptimJob.denoise, when requested
        self.auc = None                 # ROC-AUC against the motif ground truth computed on the device, when requested


class Explainer:
    def __init__(self, model, adj, feat, label, pred, train_idx, args, writer=None, print_training=True,
                 graph_mode=False, graph_idx=False):
        self.model = model
        self.model.eval()
        self.adj = adj
        self.feat = feat
        self.label = label
        self.pred = pred
        self.train_idx = train_idx
        self.n_hops = args.num_gc_layers
        self.graph_mode = graph_mode
        self.graph_idx = graph_idx
        self.args = args
        self.writer = writer
        self.print_training = print_training
        _check_supported(args)
        self._khop = {}             # graph index -> KHopIndex (sparse; replaces the dense utils/graph_utils.py:147-158)
        self._dev_graph = {}        # graph index -> engine.DeviceGraph (CSR + features on the GPU, uploaded once)
        self.last_result = None     # JobResult of the most recent batch (feature masks, loss traces)
        self.last_time = None

    # -- neighbourhoods ------------------------------------------------------------------------
    def _index(self, graph_idx):
        if graph_idx not in self._khop:
            self._khop[graph_idx] = KHopIndex(_np(self.adj[graph_idx]), self.n_hops)
        return self._khop[graph_idx]

    @property
    def neighborhoods(self):
        """Dense [G, N, N] int matrix like the reference attribute — built on demand, small graphs only."""
        if self.graph_mode:
            return None
        out = []
        for g in range(len(self.adj)):
            idx = self._index(g)
            m = np.zeros((idx.num_nodes, idx.num_nodes), dtype=int)
            for v in range(idx.num_nodes):
                m[v, idx.neighbors(v)] = 1
            out.append(m)
        return np.stack(out)

    def extract_neighborhood(self, node_idx, graph_idx=0):
        """Returns (node_idx_new, sub_adj, sub_feat, sub_label, neighbors) — explain.py:492-501."""
        node_idx_new, sub_adj, neighbors = self._index(graph_idx).extract(node_idx)
        sub_adj = sub_adj.astype(_np(self.adj).dtype)
        sub_feat = _np(self.feat)[graph_idx, neighbors]
        sub_label = _np(self.label)[graph_idx][neighbors]
        return node_idx_new, sub_adj, sub_feat, sub_label, neighbors

    # -- batched core --------------------------------------------------------------------------
    def _node_subgraph(self, node_idx, graph_idx):
        node_idx_new, sub_adj, sub_feat, sub_label, neighbors = self.extract_neighborhood(node_idx, graph_idx)
        if len(neighbors) == 0:
            raise IndexError("node %d has an empty %d-hop neighbourhood" % (node_idx, self.n_hops))
        pred_label = np.argmax(_np(self.pred)[graph_idx][neighbors], axis=1)          # explain.py:105
        gt = int(sub_label[node_idx_new])                                              # explain.py:751
        return Subgraph(np.asarray(sub_adj, np.float32), np.asarray(sub_feat, np.float32), gt, int(node_idx_new),
                        pred_label, None), sub_adj

    def _node_subgraphs(self, nodes, graph_idx):
        """Batched form of _node_subgraph: one vectorised k-hop pass for all targets."""
        feat, label, pred = _np(self.feat), _np(self.label), _np(self.pred)
        adj_dtype = _np(self.adj).dtype
        out = []
        for v, (new, sub, nb) in zip(nodes, self._index(graph_idx).extract_batch(nodes)):
            if len(nb) == 0:
                raise IndexError("node %d has an empty %d-hop neighbourhood" % (v, self.n_hops))
            pl = np.argmax(pred[graph_idx][nb], axis=1)
            out.append((Subgraph(sub, np.asarray(feat[graph_idx, nb], np.float32), int(label[graph_idx][nb][new]), new,
                                 pl, None), sub.astype(adj_dtype)))
        return out

    def _graph_subgraph(self, graph_idx):
        sub_adj = _np(self.adj[graph_idx])                                             # explain.py:82-85
        sub_feat = _np(self.feat[graph_idx])
        gt = int(_np(self.label[graph_idx]))
        return Subgraph(np.asarray(sub_adj, np.float32), np.asarray(sub_feat, np.float32), gt, 0, None, None), sub_adj

    def _device_graph(self, graph_idx):
        if graph_idx not in self._dev_graph:
            self._dev_graph[graph_idx] = device_graph(self._index(graph_idx).csr, _np(self.feat)[graph_idx],
                                                      _np(self.pred)[graph_idx], device=_ENGINE["device"])
        return self._dev_graph[graph_idx]

    def explain_batch(self, node_indices=None, graph_indices=None, graph_idx=0, record_loss=False, use_graph=False, stats=False,
                      unconstrained=False):
        """All targets as ONE job on the GPU. Returns list of float64 masked adjacencies (reference layout).

        Node mode runs device-side end to end: the k-hop walk sets (gnnx_khop), the dense sub-adjacencies, feature rows
        and predicted labels (gnnx_pack_csr) all come from the CSR graph resident on the GPU; the host only draws the
        initial masks (the caller's torch CPU generator, one normal_ per target in list order, like
        construct_edge_mask) and receives the masks as edge lists (gnnx_gather_edges).  With torch.distributed
        initialised (one process per GPU) the targets are sharded over the ranks by modelled GPU time (parallel.target_cost, parallel.run_sharded) and
        every rank returns the full list.

        unconstrained=True (explain.py:688-691): the optimised masked adjacency is sym(sigmoid(mask)) * (1 - I) - not multiplied by
        the sub-graph's adjacency - i.e. the same optimisation on the COMPLETE graph over the sub-graph's nodes; the result is still
        multiplied by sub_adj (explain.py:209-211).  Runs on the same kernels with the packed adjacency replaced by 1 - I."""
        begin = time.time()
        lib, device = _ENGINE["lib"], _ENGINE["device"]
        reason = _torch_route_reason(self.args, self.model, graph_mode=graph_indices is not None, record_loss=record_loss,
                                     unconstrained=unconstrained)
        if reason is None and self.graph_mode and graph_indices is None:
            # --graph-idx N --explain-node K (explainer_main.py:257-258 with a graph-mode Explainer): the node head on a graph encoder
            reason = "a node explanation on a graph-mode Explainer"
        if reason is not None:
            out = self._torch_route(node_indices, graph_indices, graph_idx, unconstrained, "exp", reason)
            self.last_time = time.time() - begin
            return out
        sd = self.model.state_dict()
        relu = getattr(self.args, "mask_act", "sigmoid") == "ReLU"
        bn = bool(getattr(self.args, "bn", False))
        record_loss = record_loss and not relu          # the reference's loss is NaN there; nothing to log
        if graph_indices is not None:
            if not self.graph_mode:
                raise ValueError("graph_indices given to a node-mode Explainer")
            targets = list(graph_indices)
            built = [self._graph_subgraph(g) for g in targets]
            subs = [b[0] for b in built]
            masks = [init_edge_mask(s.adj.shape[0]) for s in subs]     # same RNG stream as ExplainModule.__init__ per target
            if unconstrained:
                subs = [Subgraph(1.0 - np.eye(s.adj.shape[0], dtype=np.float32), s.feat, s.gt_label, 0, None, None) for s in subs]
            job = MaskOptimJob(subs, sd, graph_mode=True, device=device, lib=lib, mask_relu=relu, bn=bn)
            hy = _hyper(self.args, record_loss=record_loss, use_graph=use_graph and len(targets) > 1)
            job.set_masks(masks)
            job.launch(hy, state=_unmasked_features_state(job) if unconstrained else None)
            res = job.fetch(hy)
            if unconstrained:
                res.feat_mask[:] = _regulariser_only_feat_mask(self.args, job.D, hy.num_iters)
            job.close()
            self.last_time = time.time() - begin
            self.last_result = res
            # explain.py:209-211: float32 mask * float64 sub_adj -> float64
            return [ma.astype(np.float64) * np.asarray(b[1], np.float64) for ma, b in zip(res.masked_adj, built)]
        targets = np.asarray([int(v) for v in node_indices], np.int64)
        graph = self._device_graph(graph_idx)
        dn = khop_device(graph, targets, self.n_hops, lib=lib)
        for v, n in zip(targets, dn.sizes):
            if n == 0:
                raise IndexError("node %d has an empty %d-hop neighbourhood" % (v, self.n_hops))
        labels = _np(self.label)[graph_idx][targets]                    # explain.py:751
        hy = _hyper(self.args, record_loss=record_loss, use_graph=use_graph and len(targets) > 1)
        # Round 6: targets of more than XL_MIN_N sub-graph nodes take the XL route (engine.XLJob: sub-graph CSRs from the resident graph, masks and
        # results as edge lists - no dense n x n block on the device; beyond 16 383 nodes the dense routes would stream 28 n^2 bytes per iteration).
        # Plain configurations only; everything else stays on the dense-packed job.
        xl_sel = np.zeros(len(targets), bool)
        if (graph.binary and not (relu or bn or record_loss or unconstrained) and getattr(self.args, "method", "base") == "base" and
                int(np.asarray(_np(self.pred)).shape[-1]) <= 8):
            xl_sel = dn.sizes > XL_MIN_N
        # The initial masks: ONE normal_ draw of n x n values per target from the caller's global generator, in target order, like
        # construct_edge_mask (explain.py:645-652).  The dense-packed targets' draws land in one buffer; an XL target's draw is a temporary of which only
        # the values on its edges are kept (its n^2 values still pass through the generator: the stream every later target sees is the reference's).
        xl_pos = np.nonzero(xl_sel)[0]
        xl_all = XLJob(graph, dn.subset(xl_pos), None, labels[xl_pos], sd, lib=lib) if len(xl_pos) else None
        xl_vals = None
        if xl_all is not None:
            xl_eoff, xl_rc = xl_all.edge_ids()
            xl_rc = xl_rc.cpu().numpy()
            xl_vals = np.zeros((int(xl_eoff[-1]), 2), np.float32)
        dense_sizes = np.where(xl_sel, 0, dn.sizes).astype(np.int64)
        raw_off = np.zeros(len(targets) + 1, np.int64)
        np.cumsum(dense_sizes ** 2, out=raw_off[1:])
        raw = torch.empty(int(raw_off[-1]), dtype=torch.float32)
        import math
        xk = 0
        for k, n_k in enumerate(dn.sizes):
            std = math.sqrt(2.0) * math.sqrt(2.0 / (int(n_k) + int(n_k)))
            if xl_sel[k]:
                m0 = torch.empty(int(n_k), int(n_k)).normal_(1.0, std).numpy()
                e = xl_rc[int(xl_eoff[xk]):int(xl_eoff[xk + 1])]
                xl_vals[int(xl_eoff[xk]):int(xl_eoff[xk + 1]), 0] = m0[e[:, 0], e[:, 1]]
                xl_vals[int(xl_eoff[xk]):int(xl_eoff[xk + 1]), 1] = m0[e[:, 1], e[:, 0]]
                del m0
                xk += 1
            else:
                raw[raw_off[k]:raw_off[k + 1]].normal_(1.0, std)
        last = {}

        def compute_xl(idxs):
            """The XL targets among idxs (positions in `targets`) -> {position: float64 masked adjacency}"""
            pos = [i for i in idxs if xl_sel[i]]
            if not pos:
                return {}
            where = {int(p_): j for j, p_ in enumerate(xl_pos)}
            if len(pos) == len(xl_pos):
                xj, sel = xl_all, np.arange(len(xl_pos))
            else:
                sel = np.asarray([where[int(i)] for i in pos], np.int64)
                xj = XLJob(graph, dn.subset(np.asarray(pos, np.int64)), None, labels[pos], sd, lib=lib)
            vals = np.concatenate([xl_vals[int(xl_eoff[j]):int(xl_eoff[j + 1])] for j in sel]) if len(sel) else np.zeros((0, 2), np.float32)
            xj.set_masks_on_edges(torch.from_numpy(np.ascontiguousarray(vals)))
            xj.launch(hy)
            em = xj.fetch_edges()
            last.setdefault("xl_feat", {}).update({int(p_): em.feat_mask[j] for j, p_ in enumerate(pos)})
            return {int(p_): em.dense(j) for j, p_ in enumerate(pos)}

        def compute(idxs):
            """This rank's shard (all targets on one GPU): -> one float64 masked adjacency per target."""
            xl_out = compute_xl(idxs)
            all_idxs = list(idxs)
            idxs = [i for i in all_idxs if not xl_sel[i]]
            if not idxs:
                last.update(feat_mask=np.stack([last["xl_feat"][int(i)] for i in all_idxs]), loss=None, edges=None, rows=dn.rows[all_idxs])
                return [xl_out[int(i)] for i in all_idxs]
            if xl_out:
                dense_out = compute_dense(idxs)
                merged, it = [], iter(dense_out)
                fm_dense = iter(last["feat_mask"])
                fms = []
                for i in all_idxs:
                    if xl_sel[i]:
                        merged.append(xl_out[int(i)])
                        fms.append(last["xl_feat"][int(i)])
                    else:
                        merged.append(next(it))
                        fms.append(next(fm_dense))
                last.update(feat_mask=np.stack(fms), edges=None, rows=dn.rows[all_idxs], denoised=None, auc=None)
                return merged
            return compute_dense(idxs)

        def compute_dense(idxs):
            if len(idxs) == len(targets):
                sub_dn, sub_raw = dn, raw
            else:
                sub_dn = dn.subset(np.asarray(idxs, np.int64))
                sub_raw = torch.cat([raw[raw_off[i]:raw_off[i + 1]] for i in idxs])
            job = MaskOptimJob.from_csr(graph, sub_dn, None, labels[idxs], sd, lib=lib, mask_relu=relu, bn=bn, analyze=not unconstrained)
            true_adj = None
            if unconstrained:
                true_adj = job.adjacency()                 # sub_adj of every target (explain.py:209-211 multiplies the result by it)
                job.set_complete_graphs()                  # the optimisation sees 1 - I
                job.analyze()
            job.set_masks_raw(sub_raw)
            job.launch(hy, state=_unmasked_features_state(job) if unconstrained else None)
            if unconstrained:
                res = job.fetch(hy)
                res.feat_mask[:] = _regulariser_only_feat_mask(self.args, job.D, hy.num_iters)
                out = [ma.astype(np.float64) * a.astype(np.float64) for ma, a in zip(res.masked_adj, true_adj)]
                last.update(feat_mask=res.feat_mask, loss=res.loss, edges=None, rows=sub_dn.rows)
            elif graph.binary:            # explain.py:209-211 multiplies by sub_adj: a no-op for a 0/1 adjacency
                em = job.fetch_edges()
                out = [em.dense(k) for k in range(len(idxs))]
                last.update(feat_mask=em.feat_mask, loss=job.loss.cpu().numpy() if record_loss else None, edges=em, rows=sub_dn.rows)
                # the device AUC scores EVERY upper-triangle edge; make_pred_real only the entries with masked_adj > 0 (explain.py:544):
                # the two agree while every edge value is finite and positive - checked, else the host path below decides
                if stats and len(idxs) == len(targets) and not relu and bool(np.isfinite(em.masked_adj).all() and (em.masked_adj > 0).all()):
                    # explain.py:306-351 on the device, on the edge lists: denoise_graph(threshold_num=20) per target and the
                    # ROC-AUC of all targets' edge scores against the motif ground truth
                    real = self._motif_truth(em, sub_dn.rows)
                    last.update(denoised=job.denoise(20), auc=job.auc(real)[0])
            else:
                res = job.fetch(hy)
                out = [ma.astype(np.float64) * a.astype(np.float64) for ma, a in zip(res.masked_adj, job.adjacency())]
                last.update(feat_mask=res.feat_mask, loss=res.loss, edges=None, rows=sub_dn.rows)
            job.close()
            return out

        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            from ..parallel import run_sharded, target_cost
            got = run_sharded(list(range(len(targets))), target_cost(dn.sizes), compute)
            out = [got[i] for i in range(len(targets))]
        else:
            out = compute(list(range(len(targets))))
        self.last_time = time.time() - begin
        self.last_result = _Result(last.get("feat_mask"), last.get("loss"), last.get("edges"))
        self.last_result.denoised, self.last_result.auc = last.get("denoised"), last.get("auc")
        self.last_rows = dn.rows
        return out

    def _torch_route(self, node_indices, graph_indices, graph_idx, unconstrained, kind, reason):
        """One target after the other through explainer/torch_route.py (torch autograd on the HIP device through self.model)."""
        from . import torch_route
        device = torch_route.device_for(_ENGINE["device"])
        out, fms = [], []
        if graph_indices is not None:
            for g in graph_indices:
                sub, sub_adj = self._graph_subgraph(g)
                ma, fm = torch_route.explain_one(self.model, sub_adj, sub.feat, _np(self.label[g]), None, 0, self.args, dict(COEFFS), _torch_optimizer,
                                                 init_edge_mask(sub.adj.shape[0]), self.graph_idx, True, unconstrained, kind, device, reason)
                out.append(ma)
                fms.append(fm)
        else:
            for v in node_indices:
                if self.graph_mode:                       # explain(node, graph_idx=g) on a graph-mode Explainer: the whole graph g, node v
                    new, sub_adj, nb = int(v), _np(self.adj[graph_idx]), np.arange(_np(self.adj[graph_idx]).shape[0])
                    sub_feat, sub_label = _np(self.feat[graph_idx]), np.full(len(nb), int(_np(self.label[graph_idx])))
                    pred_label = np.full(len(nb), int(np.argmax(_np(self.pred)[0][graph_idx])))
                else:
                    new, sub_adj, sub_feat, sub_label, nb = self.extract_neighborhood(int(v), graph_idx)
                    if len(nb) == 0:
                        raise IndexError("node %d has an empty %d-hop neighbourhood" % (v, self.n_hops))
                    pred_label = np.argmax(_np(self.pred)[graph_idx][nb], axis=1)
                ma, fm = torch_route.explain_one(self.model, sub_adj, sub_feat, sub_label, pred_label, int(new), self.args, dict(COEFFS),
                                                 _torch_optimizer, init_edge_mask(len(nb)), self.graph_idx, False, unconstrained, kind, device, reason)
                out.append(ma)
                fms.append(fm)
        self.last_result = _Result(None, None, None)
        self.last_result.feat_mask_sigmoid = np.stack(fms)
        self.last_rows = None
        return out

    def explain_grad(self, node_indices, graph_idx=0, graph_mode=False):
        """The gradient baseline of the reference (`explain(..., model="grad")`, explain.py:125-133 with adj_feat_grad
        :717-738) for a list of targets as ONE batched job: one forward + backward of the encoder on every unmasked
        sub-graph, loss = -log softmax(logits[node])[predicted label], result sigmoid(|dL/dA| + |dL/dA|^T) * sub_adj
        (float64, like the reference's float32 tensor times the float64 sub_adj)."""
        if graph_mode or self.graph_mode:
            # the reference indexes pred_label[node_idx_new] (explain.py:130), which only exists in node mode
            raise NotImplementedError("the gradient baseline is a node-mode path")
        if getattr(self.args, "bn", False):
            raise NotImplementedError("the gradient baseline with --bn is not implemented on the HIP path")
        sd = self.model.state_dict()
        if any(k.endswith("att_weight") for k in sd) or _torch_route_reason(self.args, self.model, state_dict=sd) is not None:
            raise NotImplementedError("the gradient baseline (gnnx_grad_baseline) implements the base 3-layer encoder only: %s" %
                                      (_torch_route_reason(self.args, self.model, state_dict=sd) or "an attention encoder (method='att')"))
        lib = _ENGINE["lib"]
        targets = np.asarray([int(v) for v in node_indices], np.int64)
        graph = self._device_graph(graph_idx)
        dn = khop_device(graph, targets, self.n_hops, lib=lib)
        for v, n in zip(targets, dn.sizes):
            if n == 0:
                raise IndexError("node %d has an empty %d-hop neighbourhood" % (v, self.n_hops))
        pred_label = np.argmax(_np(self.pred)[graph_idx][targets], axis=1)          # explain.py:105, 130: pred_label[node_idx_new]
        job = MaskOptimJob.from_csr(graph, dn, None, pred_label, self.model.state_dict(), lib=lib, analyze=False)
        out = [g.astype(np.float64) for g in job.grad_baseline()]                    # already multiplied by sub_adj on the device
        job.close()
        return out

    def _save(self, masked_adj, node_idx):
        fname = "masked_adj_" + io_utils.gen_explainer_prefix(self.args) + (
            "node_idx_" + str(node_idx) + "graph_idx_" + str(self.graph_idx) + ".npy")
        with open(os.path.join(self.args.logdir, fname), "wb") as outfile:
            np.save(outfile, np.asarray(masked_adj.copy()))
        return fname

    # -- reference API ---------------------------------------------------------------------------
    def explain(self, node_idx, graph_idx=0, graph_mode=False, unconstrained=False, model="exp"):
        """Explain a single node (or graph) prediction — explain.py:74-221."""
        if model == "grad":
            masked_adj = self.explain_grad([node_idx], graph_idx=graph_idx, graph_mode=graph_mode)[0]
            fname = self._save(masked_adj, node_idx)
            print("Saved adjacency matrix to ", fname)
            return masked_adj
        if model != "exp":                                     # explain.py:200-208: any other name is the attention baseline
            targets = dict(graph_indices=[graph_idx], node_indices=None) if graph_mode else dict(node_indices=[node_idx], graph_indices=None)
            masked_adj = self._torch_route(graph_idx=graph_idx, unconstrained=unconstrained, kind=model,
                                           reason="model=%r (the attention baseline, explain.py:200-208)" % model, **targets)[0]
            fname = self._save(masked_adj, node_idx)
            print("Saved adjacency matrix to ", fname)
            return masked_adj
        if graph_mode:
            masked_adj = self.explain_batch(graph_indices=[graph_idx], record_loss=self.print_training, unconstrained=unconstrained)[0]
        else:
            masked_adj = self.explain_batch(node_indices=[node_idx], graph_idx=graph_idx,
                                            record_loss=self.print_training, unconstrained=unconstrained)[0]
        if self.print_training and self.last_result.loss is not None:
            tr = self.last_result.loss[0]        # logged by the kernel the optimisation ran on (the resident kernel's logging form)
            # explain.py:149-159 prints "epoch, loss, mask density, pred" every epoch: the five terms of ExplainModule.loss (:808-819), the density
            # ExplainModule.mask_density takes AFTER the epoch's step (:142-148, 680-683) and the class probabilities of the epoch's forward
            # (:710-714) - all logged by the kernels (include/gnnx.h: GNNX_LOSS_TERMS), no second pass, no device sync per epoch
            C = min(int(_np(self.pred).shape[-1]), 8)
            for epoch in range(len(tr)):
                print("epoch: ", epoch, "; loss: ", float(tr[epoch, :5].sum()), "; mask density: ", float(tr[epoch, 5]), "; pred: ", tr[epoch, 8:8 + C])
        print("finished training in ", self.last_time)
        fname = self._save(masked_adj, node_idx)
        print("Saved adjacency matrix to ", fname)
        return masked_adj

    def explain_nodes(self, node_indices, args, graph_idx=0):
        """explain.py:225-292 without the alignment / plotting post-processing."""
        masked_adjs = self.explain_batch(node_indices=node_indices, graph_idx=graph_idx)
        for v, ma in zip(node_indices, masked_adjs):
            self._save(ma, v)
        return masked_adjs

    def explain_nodes_gnn_stats(self, node_indices, args, graph_idx=0, model="exp"):
        """explain.py:295-353: batch explanation + the ROC-AUC text file (no plots).  Like the reference this needs a
        dataset with motif ground truth (make_pred_real: syn1 / syn2 / syn4) and raises otherwise."""
        if model not in ("exp", "grad"):
            raise NotImplementedError("model=%r: 'exp' and 'grad' are implemented" % model)
        node_indices = list(node_indices)
        if model == "grad":        # explain.py:296-299 with model="grad": the baseline's masks through the same AUC evaluation
            masked_adjs = self.explain_grad(node_indices, graph_idx=graph_idx)
            nbs = khop_device(self._device_graph(graph_idx), np.asarray(node_indices, np.int64), self.n_hops, lib=_ENGINE["lib"])
            self.last_rows = nbs.rows
        else:
            masked_adjs = self.explain_batch(node_indices=node_indices, graph_idx=graph_idx, stats=True)
        for v, ma in zip(node_indices, masked_adjs):
            self._save(ma, v)
        if model == "exp" and self.last_result.auc is not None:
            self.last_auc = float(self.last_result.auc)                               # pair counts from the device (gnnx_auc_counts)
        else:                                                                         # sharded / weighted / grad runs: on the host
            pred_all, real_all = [], []
            for ma, new_idx in zip(masked_adjs, self.last_rows):                      # node_idx_new came with the k-hop lists
                pred, real = self.make_pred_real(ma, int(new_idx))
                pred_all.append(pred)
                real_all.append(real)
            from sklearn.metrics import roc_auc_score
            self.last_auc = float(roc_auc_score(np.concatenate(real_all), np.concatenate(pred_all)))   # raises on single-class labels, as the reference
        os.makedirs("log/pr", exist_ok=True)
        with open("log/pr/auc_" + self.args.dataset + "_" + model + ".txt", "w") as f:
            f.write("dataset: {}, model: {}, auc: {}\n".format(self.args.dataset, "exp", str(self.last_auc)))   # "exp" is hard-coded in the reference too (explain.py:349)
        return masked_adjs

    def explain_graphs(self, graph_indices):
        """explain.py:356-402 without denoise/plot logging."""
        graph_indices = list(graph_indices)
        masked_adjs = self.explain_batch(graph_indices=graph_indices)
        for g, ma in zip(graph_indices, masked_adjs):
            self._save(ma, 0)
        return masked_adjs

    def _motif(self):
        ds = getattr(self.args, "dataset", None)
        if ds in ("syn1", "syn2"):
            return [(0, 1), (1, 2), (2, 3), (0, 3), (0, 4), (1, 4)]
        if ds == "syn4":
            return [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]
        raise ValueError("no motif ground truth for dataset %r (explain.py:535-579 covers syn1, syn2, syn4)" % ds)

    def _motif_truth(self, em, rows):
        """make_pred_real (explain.py:535-579) on edge lists: 0/1 per upper-triangle edge of every target."""
        motif = self._motif()
        span = max(max(m) for m in motif)
        real = np.zeros(int(em.eoff[-1]), np.uint8)
        for k, st in enumerate(rows):
            if st + span >= em.n[k]:
                raise IndexError("motif of node_idx_new=%d reaches beyond the %d-node sub-graph" % (st, em.n[k]))
            a, b = int(em.eoff[k]), int(em.eoff[k + 1])
            r, c = em.rc[a:b, 0] - st, em.rc[a:b, 1] - st
            hit = np.zeros(b - a, bool)
            for x, y in motif:
                hit |= (r == x) & (c == y)
            real[a:b] = hit
        return real

    def make_pred_real(self, adj, start):
        """Motif ground truth for syn1/syn2 (house) and syn4 (cycle) — explain.py:535-579.  Other datasets have none:
        the reference fails there (its `real` is never bound); so does this."""
        ds = getattr(self.args, "dataset", None)
        if ds in ("syn1", "syn2"):
            motif = [(0, 1), (1, 2), (2, 3), (0, 3), (0, 4), (1, 4)]
        elif ds == "syn4":
            motif = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]
        else:
            raise ValueError("no motif ground truth for dataset %r (explain.py:535-579 covers syn1, syn2, syn4)" % ds)
        if start + max(max(m) for m in motif) >= adj.shape[0]:
            raise IndexError("motif of node_idx_new=%d reaches beyond the %d-node sub-graph" % (start, adj.shape[0]))
        pred = adj[np.triu(adj) > 0]
        real = adj.copy()
        for a, b in motif:
            if real[start + a][start + b] > 0:
                real[start + a][start + b] = 10
        real = real[np.triu(real) > 0]
        real[real != 10] = 0
        real[real == 10] = 1
        return pred, real


class ExplainModule(nn.Module):
    """One target's mask parameters with the reference's attribute surface (explain.py:582-715).

    `mask`, `feat_mask`, `masked_adj`, `forward`, `loss`, `mask_density` behave like the reference; the
    `num_epochs` loop itself is `optimize()` — one fused call into the HIP engine instead of
    zero_grad / forward / loss / backward / optimizer.step per epoch (explain.py:137-146)."""

    def __init__(self, adj, x, model, label, args, graph_idx=0, writer=None, use_sigmoid=True, graph_mode=False,
                 node_idx=0, pred_label=None):
        super().__init__()
        _check_supported(args)
        if not use_sigmoid:
            raise NotImplementedError("use_sigmoid=False is not implemented on the HIP path")
        if any(k.endswith("att_weight") for k in model.state_dict()):
            raise NotImplementedError("the ExplainModule mirror steps the base encoder only: explain an attention encoder (method='att') "
                                      "through Explainer.explain / explain_nodes, which run it on k_att")
        self.adj, self.x, self.model, self.label = adj, x, model, label
        self.graph_idx, self.args, self.writer, self.graph_mode = graph_idx, args, writer, graph_mode
        self.mask_act = args.mask_act
        n = adj.size()[1]
        self.mask = nn.Parameter(torch.from_numpy(init_edge_mask(n)))          # explain.py:645-652
        self.feat_mask = nn.Parameter(torch.zeros(x.size(-1)))                  # explain.py:639-641
        # explain.py:657-661: a zero Parameter that provably never moves (see _check_supported)
        self.mask_bias = nn.Parameter(torch.zeros(n, n)) if getattr(args, "mask_bias", False) else None
        self.diag_mask = torch.ones(n, n) - torch.eye(n)
        self.coeffs = dict(COEFFS)
        # explain.py:620-622: build_optimizer(args, params) - a real torch optimiser / scheduler over the mirror's parameters: same
        # `.optimizer` / `.scheduler` attributes, param_groups and (after optimize()) state as the reference's
        params = [self.mask, self.feat_mask] + ([self.mask_bias] if self.mask_bias is not None else [])
        self.scheduler, self.optimizer = _torch_optimizer(args, params)
        self.masked_adj = None
        self._node_idx = int(node_idx)
        lab = _np(label)
        gt = int(lab) if graph_mode else int(lab.reshape(-1)[self._node_idx])
        self._sub = Subgraph(_np(adj)[0].astype(np.float32), _np(x)[0].astype(np.float32), gt, self._node_idx,
                             None if graph_mode else np.asarray(pred_label), None)
        self._job = MaskOptimJob([self._sub], model.state_dict(), graph_mode=graph_mode, device=_ENGINE["device"], lib=_ENGINE["lib"],
                                 mask_relu=(args.mask_act == "ReLU"), bn=bool(getattr(args, "bn", False)))

    def forward(self, node_idx, unconstrained=False, mask_features=True, marginalize=False):
        if unconstrained or marginalize or not mask_features:
            raise NotImplementedError("only the default forward (explain.py:693-707) is implemented")
        if not self.graph_mode and int(node_idx) != self._node_idx:
            raise ValueError("ExplainModule was built for node_idx=%d" % self._node_idx)
        probs, ma = self._job.forward([self.mask.detach().numpy()], self.feat_mask.detach().numpy()[None])
        self.masked_adj = torch.from_numpy(ma[0])[None]
        return torch.from_numpy(probs[0].copy()), None

    def mask_density(self):
        if self.masked_adj is None:
            self.forward(self._node_idx)
        return self.masked_adj.sum() / torch.as_tensor(_np(self.adj)).sum()      # explain.py:680-683

    def loss(self, pred, pred_label, node_idx, epoch):
        """Scalar loss of the current parameters (explain.py:740-808), evaluated on the host from the engine's
        forward; logging only — the optimisation itself is `optimize()`."""
        m = torch.sigmoid(self.mask.detach()) if self.mask_act == "sigmoid" else torch.relu(self.mask.detach())   # explain.py:757-760
        fm = torch.sigmoid(self.feat_mask.detach())
        gt = self._sub.gt_label
        out = -torch.log(pred[gt]) + self.coeffs["size"] * m.sum() + self.coeffs["feat_size"] * fm.mean()
        out = out + self.coeffs["ent"] * (-m * torch.log(m) - (1 - m) * torch.log(1 - m)).mean()
        if not self.graph_mode:
            y = torch.tensor(np.asarray(pred_label), dtype=torch.float)
            ma = self.masked_adj[0]
            out = out + self.coeffs["lap"] * (y @ (torch.diag(ma.sum(0)) - ma) @ y) / ma.numel()
        return out

    def optimize(self, num_epochs=None, record_loss=False):
        """The hot loop (explain.py:137-146) on the GPU; updates mask / feat_mask / masked_adj in place."""
        # the streaming kernels keep EVERY entry of the dense mask up to date (the edge-sparse resident kernels only the
        # entries on edges), so `self.mask` - and a later `loss()` over all n^2 entries - match the reference's parameter
        # A second call continues where the first stopped - moments, step count for the bias corrections and position in the learning-rate
        # schedule - like the torch optimiser / scheduler the reference keeps in the module (explain.py:620-622).
        n = int(num_epochs) if num_epochs is not None else int(self.args.num_epochs)
        prev = getattr(self, "_state", None)
        first = 0 if prev is None else int(prev.first_iter)
        hy = _hyper(self.args, num_iters=n, record_loss=record_loss, use_resident=False)
        if hy.lr_schedule is not None:
            hy.lr_schedule = _lr_schedule(self.args, first + n)[first:]
        job = self._job
        job.set_masks([self.mask.detach().numpy()])
        job.launch(hy, state=prev, keep_state=True)
        self._state = job.state_out
        res = job.fetch(hy)
        with torch.no_grad():
            self.mask.copy_(torch.from_numpy(res.mask[0]))
            self.feat_mask.copy_(torch.from_numpy(res.feat_mask[0]))
        self.masked_adj = torch.from_numpy(res.masked_adj[0])[None]
        # the optimiser state the reference's torch optimiser would hold now (the engine's moments, dense: the streaming kernels keep
        # every entry), and the learning rate its scheduler would have left
        n = self._sub.adj.shape[0]
        st = job.state_out
        sq = lambda a: torch.from_numpy(job._square_views(a.cpu().numpy())[0][:n, :n].copy())
        fs = st.feat.cpu().numpy()[0, :, :self.feat_mask.numel()]
        steps = torch.tensor(float(first + hy.num_iters))
        names = {"adam": ("exp_avg", "exp_avg_sq"), "sgd": ("momentum_buffer", None), "rmsprop": (None, "square_avg"),
                 "adagrad": (None, "sum")}[hy.opt]
        for prm, m, v in ((self.mask, sq(st.m), sq(st.v)), (self.feat_mask, torch.from_numpy(fs[1].copy()), torch.from_numpy(fs[2].copy()))):
            state = self.optimizer.state[prm]
            if hy.opt != "sgd":
                state["step"] = steps.clone()
            if names[0]:
                state[names[0]] = m
            if names[1]:
                state[names[1]] = v
        if hy.lr_schedule is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = float(_lr_schedule(self.args, first + hy.num_iters + 1)[-1])
        return res
