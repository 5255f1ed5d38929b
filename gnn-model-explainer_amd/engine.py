"""ctypes binding of libgnnx_hip.so (include/gnnx.h) + the batched mask-optimisation job.

This is the host side of the hot path: it packs many targets' sub-graphs into the segmented
device layout, hands raw device pointers and the current HIP stream to the C ABI, and unpacks
the masks.  PyTorch is used only for device memory and streams.  There is NO CPU fallback: if the
HIP library is missing or no GPU is visible the constructor raises.

Replaces, for a whole list of targets at once, the body of the reference's
Explainer.explain (explainer/explain.py:94-146, 208-211) and everything it calls
(ExplainModule, explain.py:582-820; models.GcnEncoderNode/Graph forward, models.py:230-376;
torch autograd; torch.optim.Adam via utils/train_utils.py:9-10).
"""
import ctypes
import math
import os
from contextlib import nullcontext as _nullcontext
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

FEAT_STRIDE = 32
MAX_CLASSES = 32
LOSS_TERMS = 16      # [0..4] pred, size, lap, ent, feat_size; [5] mask density after the step; [8..15] class probabilities (include/gnnx.h)
_DEVICE_TYPE = "cuda"          # PyTorch-ROCm exposes HIP devices as "cuda"
_LIB_NAME = "libgnnx_hip.so"


_ENGINE_STREAMS = {}


def _engine_stream(dev):
    """The stream a new job's work goes to: the caller's CURRENT stream when that is not the default stream (a pipeline stage
    running under `with torch.cuda.stream(s)`: pipeline.BatchPipeline), else one process-wide engine stream per device."""
    cur = torch.cuda.current_stream(dev)
    if cur != torch.cuda.default_stream(dev):
        return cur
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _ENGINE_STREAMS:
        _ENGINE_STREAMS[key] = torch.cuda.Stream(dev)
    return _ENGINE_STREAMS[key]


class _Problem(ctypes.Structure):
    _fields_ = [("num_targets", ctypes.c_int32), ("n", ctypes.POINTER(ctypes.c_int32)),
                ("target_row", ctypes.POINTER(ctypes.c_int32)), ("gt_label", ctypes.POINTER(ctypes.c_int32)),
                ("D", ctypes.c_int32), ("H", ctypes.c_int32), ("O", ctypes.c_int32), ("C", ctypes.c_int32),
                ("graph_mode", ctypes.c_int32), ("mask_relu", ctypes.c_int32), ("bn", ctypes.c_int32)]


class _Model(ctypes.Structure):
    _fields_ = [("W", ctypes.POINTER(ctypes.c_float) * 3), ("b", ctypes.POINTER(ctypes.c_float) * 3),
                ("Wp", ctypes.POINTER(ctypes.c_float)), ("bp", ctypes.POINTER(ctypes.c_float))]


class _Hyper(ctypes.Structure):
    _fields_ = [("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("c_size", ctypes.c_float), ("c_feat_size", ctypes.c_float), ("c_ent", ctypes.c_float),
                ("c_lap", ctypes.c_float), ("num_iters", ctypes.c_int32), ("record_loss", ctypes.c_int32),
                ("use_graph", ctypes.c_int32), ("use_resident", ctypes.c_int32), ("opt", ctypes.c_int32), ("edge_results_only", ctypes.c_int32),
                ("momentum", ctypes.c_double), ("alpha", ctypes.c_double), ("lr_schedule", ctypes.POINTER(ctypes.c_double))]


OPTIMIZERS = {"adam": 0, "sgd": 1, "rmsprop": 2, "adagrad": 3}      # utils/train_utils.py:9-16


class _Resume(ctypes.Structure):
    _fields_ = [("first_iter", ctypes.c_int32), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p), ("feat", ctypes.c_void_p),
                ("m_out", ctypes.c_void_p), ("v_out", ctypes.c_void_p), ("feat_out", ctypes.c_void_p)]


class _XlState(ctypes.Structure):
    _fields_ = [("first_iter", ctypes.c_int32), ("M_e", ctypes.c_void_p), ("m_e", ctypes.c_void_p), ("v_e", ctypes.c_void_p), ("feat", ctypes.c_void_p),
                ("m_out_e", ctypes.c_void_p), ("v_out_e", ctypes.c_void_p), ("feat_out", ctypes.c_void_p)]


@dataclass
class AdamState:
    """Optimiser state of a batch (gnnx_resume, include/gnnx.h): what torch.optim.Adam keeps for ExplainModule's two
    parameters (explain.py:622) after `first_iter` steps.  Device tensors in the packed layouts; None = zeros."""
    first_iter: int = 0
    m: Optional[torch.Tensor] = None       # [Q] exp_avg of the mask
    v: Optional[torch.Tensor] = None       # [Q] exp_avg_sq
    feat: Optional[torch.Tensor] = None    # [T, 3, 32]: feat_mask, exp_avg, exp_avg_sq


@dataclass
class Hyper:
    """Adam (utils/train_utils.py:9-10, torch defaults) + ExplainModule.coeffs (explain.py:624-631)."""
    num_iters: int = 100
    lr: float = 0.1
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    c_size: float = 0.005
    c_feat_size: float = 1.0
    c_ent: float = 1.0
    c_lap: float = 1.0
    record_loss: bool = False
    use_graph: bool = False
    use_resident: bool = True     # the on-chip-resident kernels for the targets the plan routed to them
    opt: str = "adam"             # "adam" | "sgd" (momentum) | "rmsprop" (alpha, eps) | "adagrad" (eps): utils/train_utils.py:9-16
    momentum: float = 0.95        # train_utils.py:12
    alpha: float = 0.99           # torch.optim.RMSprop default
    lr_schedule: Optional[np.ndarray] = None   # [num_iters] learning rate of every iteration (an LR scheduler's trace), None = constant
    edge_results_only: bool = False   # the result is fetched as edge lists only (fetch_edges / gather_edges_device): no dense Abar blocks

    def c(self):
        hy = _Hyper(self.lr, self.beta1, self.beta2, self.eps, self.c_size, self.c_feat_size, self.c_ent,
                    self.c_lap, int(self.num_iters), int(self.record_loss), int(self.use_graph), int(self.use_resident),
                    OPTIMIZERS[self.opt], int(self.edge_results_only), self.momentum, self.alpha, None)
        if self.lr_schedule is not None:
            sch = np.ascontiguousarray(self.lr_schedule, np.float64)
            if sch.shape != (int(self.num_iters),):
                raise ValueError("lr_schedule must hold num_iters learning rates")
            hy._keep = sch                      # the struct only borrows the host array
            hy.lr_schedule = sch.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
            hy.use_graph = 0                    # a schedule is a per-launch argument of the streaming kernels: no hipGraph replay
        return hy


def library_path():
    # GNNX_LIBRARY_PATH: measurement sessions load an A/B build of the same sources (tools/); never set in production
    return os.environ.get("GNNX_LIBRARY_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", _LIB_NAME)


def _load_hip_library():
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built (run `python __graft_entry__.py` / "
            "`hipcc --offload-arch=gfx950`). This engine has no CPU fallback.")
    return ctypes.CDLL(path)


_API = {
    "gnnx_plan_create": (ctypes.c_int, [ctypes.POINTER(_Problem), ctypes.POINTER(_Model), ctypes.POINTER(ctypes.c_void_p)]),
    "gnnx_set_att_weights": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "gnnx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gnnx_total_q": (ctypes.c_int64, [ctypes.c_void_p]),
    "gnnx_total_rows": (ctypes.c_int64, [ctypes.c_void_p]),
    "gnnx_get_layout": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64)]),
    "gnnx_workspace_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "gnnx_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(_Hyper)] + [ctypes.c_void_p] * 8 +
                 [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_run_resume": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(_Hyper), ctypes.POINTER(_Resume)] + [ctypes.c_void_p] * 8 +
                        [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_plan_analyze": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_plan_analyze_features": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_get_route": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]),
    "gnnx_resident_times": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]),
    "gnnx_set_trace": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_pack_csr": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int32] + [ctypes.c_void_p] * 7),
    "gnnx_pack_csr_analyze": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int32] + [ctypes.c_void_p] * 7),
    "gnnx_forward": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_time_kernel": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(_Hyper), ctypes.c_int32, ctypes.c_int32] +
                         [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "gnnx_grad_baseline": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_edge_positions": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_edge_counts_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_edge_layout": (ctypes.c_int, [ctypes.c_void_p] * 6),
    "gnnx_gather_values": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5),
    "gnnx_mt_edge_words": (ctypes.c_int, [ctypes.c_void_p] * 7),
    "gnnx_denoise_edges": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int32] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_auc_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_khop_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "gnnx_khop": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32] +
                  [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_total_raw": (ctypes.c_int64, [ctypes.c_void_p]),
    "gnnx_scatter_masks": (ctypes.c_int, [ctypes.c_void_p] * 4),
    "gnnx_edge_counts": (ctypes.c_int, [ctypes.c_void_p] * 4),
    "gnnx_gather_edges": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_size_t, ctypes.c_void_p]),
    "gnnx_pool_trim": (ctypes.c_int, []),
    "gnnx_sparse_tiny_per_workgroup": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "gnnx_tiny_pack_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "gnnx_set_service_stream": (ctypes.c_int, [ctypes.c_void_p]),
    "gnnx_lane_stream": (ctypes.c_void_p, [ctypes.c_int32]),
    "gnnx_stream_create_cu_mask": (ctypes.c_void_p, [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32]),
    "gnnx_debug_spin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "gnnx_debug_lane_sums": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_xl_create": (ctypes.c_int, [ctypes.POINTER(_Problem), ctypes.POINTER(_Model), ctypes.POINTER(ctypes.c_void_p)]),
    "gnnx_xl_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "gnnx_xl_total_rows": (ctypes.c_int64, [ctypes.c_void_p]),
    "gnnx_xl_rows_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "gnnx_xl_count": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int32] + [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_xl_total_edges": (ctypes.c_int64, [ctypes.c_void_p]),
    "gnnx_xl_entries_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "gnnx_xl_get_layout": (ctypes.c_int, [ctypes.c_void_p] * 4),
    "gnnx_xl_build": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_xl_set_trace": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_xl_mt_edge_words": (ctypes.c_int, [ctypes.c_void_p] * 6),
    "gnnx_xl_set_clocks": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gnnx_set_mt_jump_poly": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64]),
    "gnnx_set_mt_jump_polys": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32]),
    "gnnx_xl_run": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(_Hyper), ctypes.POINTER(_XlState)] + [ctypes.c_void_p] * 5),
    "gnnx_last_error": (ctypes.c_char_p, []),
    "gnnx_version": (ctypes.c_char_p, []),
}


def bind(lib):
    for name, (res, args) in _API.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib_cache = None


def get_library():
    global _lib_cache
    if _lib_cache is None:
        _lib_cache = bind(_load_hip_library())
    return _lib_cache


class GnnxError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        raise GnnxError(lib.gnnx_last_error().decode())


def _fptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


STATE_KEYS = ("conv_first.weight", "conv_first.bias", "conv_block.0.weight", "conv_block.0.bias",
              "conv_last.weight", "conv_last.bias", "pred_model.weight", "pred_model.bias")


def model_arrays(state_dict):
    """fp32 host copies of the 3-layer GCN encoder (models.py:114-132 state_dict names)."""
    extra = [k for k in state_dict if k.startswith("conv_block.") and not k.startswith("conv_block.0.")]
    if extra:
        raise NotImplementedError("the HIP path implements num_gc_layers == 3 (one conv_block), got " + str(extra))
    out = {}
    for k in STATE_KEYS:
        if k not in state_dict:
            raise NotImplementedError(f"state_dict lacks {k}: only bias=True, method='base' encoders are supported")
        v = state_dict[k]
        v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


@dataclass
class Subgraph:
    """One target's inputs (what Explainer.explain builds at explain.py:80-106)."""
    adj: np.ndarray                 # [n, n] symmetric, zero diagonal irrelevant (masked), float
    feat: np.ndarray                # [n, D]
    gt_label: int                   # label used by the prediction loss (explain.py:751)
    target_row: int = 0             # node_idx_new (node mode)
    pred_label: Optional[np.ndarray] = None   # [n] predicted class ids (node mode, Laplacian term)
    mask0: Optional[np.ndarray] = None        # [n, n] initial edge mask (explain.py:645-652)


@dataclass
class JobResult:
    masked_adj: List[np.ndarray]    # per target [n, n] float32: sigma-symmetrised mask * adj of the last forward
    mask: List[np.ndarray]          # per target [n, n] final mask parameter
    feat_mask: np.ndarray           # [T, D] final feature-mask parameter (pre-sigmoid)
    loss: Optional[np.ndarray] = None   # [T, iters, LOSS_TERMS]: pred, size, lap, ent, feat_size, mask density, -, -, class probabilities
    stats: dict = field(default_factory=dict)


@dataclass
class DeviceGraph:
    """The full graph resident on the device in CSR form (uploaded once, reused by every batch)."""
    indptr: torch.Tensor       # int64 [N+1]
    indices: torch.Tensor      # int32 [nnz]
    weights: Optional[torch.Tensor]   # float32 [nnz] or None (binary adjacency)
    feat: torch.Tensor         # float32 [N, D]
    pred_label: Optional[torch.Tensor]   # float32 [N] predicted class ids (node mode)
    num_nodes: int
    binary: bool


@dataclass
class DeviceNeighbors:
    """k-hop neighbour lists of a batch of targets, resident on the device (khop_device)."""
    sizes: np.ndarray          # host int32 [T]
    rows: np.ndarray           # host int32 [T]: node_idx_new of every target (-1: the target is not in its own set)
    nb_flat: torch.Tensor      # device int32 [sum sizes], ascending per target
    nb_off: torch.Tensor       # device int64 [T + 1]: start of every list in nb_flat (prefix sums of the sizes, or a fixed stride: khop_device(one_pass=True))

    def __len__(self):
        return len(self.sizes)

    def subset(self, idx):
        """The lists of the targets idx (host int array) as another DeviceNeighbors over the SAME device list buffer: only the per-target start offsets
        are gathered (the packing kernels read a target's start and take its length from the plan)."""
        idx = np.asarray(idx, np.int64)
        it = torch.from_numpy(idx).to(self.nb_off.device)
        dn = DeviceNeighbors(np.ascontiguousarray(self.sizes[idx]), np.ascontiguousarray(self.rows[idx]), self.nb_flat, self.nb_off.index_select(0, it).contiguous())
        dn._keepalive = getattr(self, "_keepalive", None)
        return dn

    def lists(self):
        """Host copies, one ascending id array per target (tests / inspection)."""
        flat, off = self.nb_flat.cpu().numpy(), self.nb_off.cpu().numpy()
        return [flat[off[t]:off[t] + int(self.sizes[t])].astype(np.int64) for t in range(len(self.sizes))]


@dataclass
class EdgeMasks:
    """The explanation of a batch as edge lists: for target t the upper-triangle edges eoff[t]:eoff[t+1] of its
    sub-graph in row-major order (r < c), with the masked adjacency of the last forward (explain.py:209-211)."""
    n: np.ndarray              # [T] sub-graph sizes
    eoff: np.ndarray           # int64 [T + 1]
    rc: np.ndarray             # int32 [E, 2]
    masked_adj: np.ndarray     # float32 [E]
    feat_mask: np.ndarray      # [T, D] final feature-mask parameter (pre-sigmoid)
    mask_rc: Optional[np.ndarray] = None   # float32 [E, 2]: final mask parameters M[r][c], M[c][r]

    def dense(self, t, dtype=np.float64):
        """The reference's return value for target t: dense symmetric [n, n] (explain.py:209-211)."""
        a, b = int(self.eoff[t]), int(self.eoff[t + 1])
        out = np.zeros((int(self.n[t]), int(self.n[t])), dtype)
        r, c = self.rc[a:b, 0], self.rc[a:b, 1]
        out[r, c] = self.masked_adj[a:b]
        out[c, r] = self.masked_adj[a:b]
        return out


def _h2d(arr, dev):
    """Host array -> device tensor without a host-blocking round trip: a pageable source makes the copy synchronous (the preparing thread of a
    pipelined job then waits for a blit kernel to find a free compute unit); through pinned staging it is only enqueued."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dev.type == _DEVICE_TYPE:
        return t.pin_memory().to(dev, non_blocking=True)
    return t.to(dev)


_ONE_PASS_MAX_INTS = 1 << 24      # padded lists of a one-pass k-hop batch: at most 64 MB


def khop_device(graph, targets, n_hops, lib=None, one_pass=False) -> DeviceNeighbors:
    """k-hop walk sets of `targets` on the device from the resident CSR graph (gnnx_khop): the neighbour lists of
    graph_utils.neighborhoods + Explainer.extract_neighborhood (utils/graph_utils.py:147-158, explain.py:492-501)
    without the host.  Two passes: sizes and node_idx_new (ONE copy back: the host needs them for the plan), then the ascending
    lists, which stay on the device (the emit pass reports nothing back, so nothing waits for it).
    one_pass=True (pipeline.BatchPipeline): ONE launch - every list gets num_nodes slots (nb_off[t] = t * num_nodes, so the offsets do
    not depend on the sizes) and the sizes come back from the same pass; falls back to two passes when the padded lists would exceed 64 MB."""
    lib = lib if lib is not None else get_library()
    dev = graph.feat.device
    T = len(targets)
    tg = _h2d(np.asarray(targets, dtype=np.int32), dev)
    sr_d = torch.empty(2, T, dtype=torch.int32, device=dev)          # sizes | rows
    sb = int(lib.gnnx_khop_scratch_bytes(graph.num_nodes, T))
    scratch = torch.empty(max(sb, 1), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream if dev.type == _DEVICE_TYPE else 0)
    args = (graph.indptr.data_ptr(), graph.indices.data_ptr(), graph.num_nodes, int(n_hops), tg.data_ptr(), T)
    if one_pass and T * graph.num_nodes <= _ONE_PASS_MAX_INTS:
        cache = graph.__dict__.setdefault("_padded_offsets", {})
        nb_off = cache.get(T)
        if nb_off is None:               # the same for every batch of T targets: uploaded once per graph
            nb_off = _h2d(np.arange(T + 1, dtype=np.int64) * graph.num_nodes, dev)
            if dev.type == _DEVICE_TYPE:
                torch.cuda.current_stream(dev).synchronize()     # other threads will use it on THEIR streams: it must have arrived
            cache[T] = nb_off
        nb_flat = torch.empty(T * graph.num_nodes, dtype=torch.int32, device=dev)
        _check(lib, lib.gnnx_khop(*args, sr_d[0].data_ptr(), nb_off.data_ptr(), nb_flat.data_ptr(), sr_d[1].data_ptr(), scratch.data_ptr(), sb, stream))
        sr = sr_d.cpu().numpy()                        # synchronises
        dn = DeviceNeighbors(sr[0].astype(np.int32), sr[1].copy(), nb_flat, nb_off)
        dn._keepalive = (tg, scratch)
        return dn
    _check(lib, lib.gnnx_khop(*args, sr_d[0].data_ptr(), None, None, sr_d[1].data_ptr(), scratch.data_ptr(), sb, stream))
    sr = sr_d.cpu().numpy()                            # synchronises
    sizes = sr[0]
    off = np.zeros(T + 1, np.int64)
    np.cumsum(sizes, out=off[1:])
    nb_off = _h2d(off, dev)
    nb_flat = torch.empty(max(int(off[-1]), 1), dtype=torch.int32, device=dev)
    _check(lib, lib.gnnx_khop(*args, None, nb_off.data_ptr(), nb_flat.data_ptr(), None, scratch.data_ptr(), sb, stream))
    dn = DeviceNeighbors(sizes.astype(np.int32), sr[1].copy(), nb_flat, nb_off)
    dn._keepalive = (tg, scratch)                      # inputs of the emit pass, which may still be running
    return dn


def device_graph(csr, feat, pred=None, device=None):
    """Upload a scipy CSR adjacency + features (+ argmax of the model's predictions) for device-side packing."""
    import scipy.sparse as sp
    csr = sp.csr_matrix(csr)
    csr.sum_duplicates()
    csr.sort_indices()
    if abs(csr - csr.T).nnz != 0:
        raise NotImplementedError("the HIP path requires a symmetric adjacency (all reference datasets are undirected)")
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the gnnx engine runs on MI355X only (no CPU fallback)")
        device = torch.device(_DEVICE_TYPE, torch.cuda.current_device())
    device = torch.device(device)
    binary = bool(np.all(csr.data == 1))
    to = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
    return DeviceGraph(to(csr.indptr, np.int64), to(csr.indices, np.int32), None if binary else to(csr.data, np.float32),
                       to(feat, np.float32), None if pred is None else to(np.argmax(pred, axis=1), np.float32),
                       csr.shape[0], binary)


class MaskOptimJob:
    """A batch of targets resident on one GPU: plan + packed device buffers."""

    @classmethod
    def from_csr(cls, graph: DeviceGraph, neighbors: Sequence[np.ndarray], target_rows, gt_labels, state_dict, lib=None,
                 analyze=True, mask_relu=False, bn=False):
        """Node-mode batch whose sub-graphs are sliced ON THE DEVICE from the CSR graph (gnnx_pack_csr): the host
        only supplies the ascending k-hop neighbour list of every target (explain.py:492-501)."""
        self = cls.__new__(cls)
        self.lib = lib if lib is not None else get_library()
        self.device = graph.feat.device
        self.graph_mode = False
        self.mask_relu = bool(mask_relu)
        self.bn = bool(bn)
        self._init_model(state_dict)
        if graph.feat.shape[1] != self.D:
            raise ValueError("feature width does not match the encoder")
        self.T = len(neighbors)
        if self.T == 0:
            raise ValueError("empty batch")
        on_device = isinstance(neighbors, DeviceNeighbors)
        self.n = np.ascontiguousarray(neighbors.sizes, np.int32) if on_device else np.asarray([len(nb) for nb in neighbors], np.int32)
        if on_device and target_rows is None:
            target_rows = neighbors.rows
        if np.any(np.asarray(target_rows) < 0):
            # gnnx_khop reports -1 when a target is not in its own walk set (an isolated node, or a directed graph): the reference's
            # node_idx_new = sum(row[:node_idx]) would silently explain whichever node sits there (explain.py:496)
            bad = np.nonzero(np.asarray(target_rows) < 0)[0]
            raise IndexError("targets %s are not in their own k-hop walk sets (isolated nodes?)" % bad[:8].tolist())
        self._create_plan(np.asarray(target_rows, np.int32), np.asarray(gt_labels, np.int32))
        self._alloc_device()
        if on_device:        # lists produced by khop_device: nothing crosses PCIe
            nb_flat, nb_off_d = neighbors.nb_flat, neighbors.nb_off
        else:
            nb_off = np.zeros(self.T + 1, np.int64)
            np.cumsum(self.n, out=nb_off[1:])
            nb_flat = torch.from_numpy(np.concatenate(neighbors).astype(np.int32)).to(self.device)
            nb_off_d = torch.from_numpy(nb_off).to(self.device)
        self._enter()
        # packing + routing analysis as one call (the packing kernel counts the rows' entries for the analysis as it places them)
        pack = self.lib.gnnx_pack_csr_analyze if analyze else self.lib.gnnx_pack_csr
        _check(self.lib, pack(
            self.handle, graph.indptr.data_ptr(), graph.indices.data_ptr(),
            graph.weights.data_ptr() if graph.weights is not None else None, graph.feat.data_ptr(), graph.feat.shape[1],
            graph.pred_label.data_ptr(), nb_flat.data_ptr(), nb_off_d.data_ptr(), self.A.data_ptr(), self.X.data_ptr(),
            self.yhat.data_ptr(), self._stream()))
        self._leave()
        self._keepalive = (nb_flat, nb_off_d, graph)
        if analyze:
            self._take_edge_counts()
        return self

    def analyze(self):
        """Let the plan look at the packed adjacency and route every target to the best kernel (gnnx_plan_analyze):
        the sparse on-chip-resident kernel for targets whose edge state fits one compute unit."""
        self._enter()
        _check(self.lib, self.lib.gnnx_plan_analyze_features(self.handle, self.A.data_ptr(), self.X.data_ptr(), self._stream()))
        self._leave()
        self._take_edge_counts()

    def _take_edge_counts(self):
        """The analysis counted the upper-triangle edges as well (the layout of the results): no separate count + copy later."""
        counts = np.zeros(self.T, np.int64)
        _check(self.lib, self.lib.gnnx_edge_counts_host(self.handle, counts.ctypes.data))
        self._edge_counts = counts
        self._eoff = None

    def set_complete_graphs(self):
        """Replace every target's packed adjacency by the complete graph on its n nodes, A = 1 - I (ExplainModule.forward with
        unconstrained=True, explain.py:688-691: the masked adjacency is not multiplied by adj).  Re-run analyze() afterwards."""
        A = np.zeros(self.Q, np.float32)
        for v, n in zip(self._square_views(A), self.n):
            v[:n, :n] = 1.0 - np.eye(int(n), dtype=np.float32)
        self.A.copy_(torch.from_numpy(A))
        self._eoff = None          # the edge layout belongs to the old adjacency
        self._edge_counts = None   # ... and so do the counts of the last analysis

    def adjacency(self):
        """Per-target dense sub-adjacencies as packed on the device (host copies)."""
        A = self.A.cpu().numpy()
        return [v[:n, :n].copy() for v, n in zip(self._square_views(A), self.n)]

    def __init__(self, subgraphs: Sequence[Subgraph], state_dict, graph_mode=False, device=None, lib=None, analyze=True, mask_relu=False, bn=False):
        self.lib = lib if lib is not None else get_library()
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("no HIP device visible: the gnnx engine runs on MI355X only (no CPU fallback)")
            device = torch.device(_DEVICE_TYPE, torch.cuda.current_device())
        self.device = torch.device(device)
        self.graph_mode = bool(graph_mode)
        self.mask_relu = bool(mask_relu)   # mask_act = "ReLU" (explain.py:669-670): dense streaming kernels
        self.bn = bool(bn)                 # --bn (models.py:222-228): dense streaming kernels
        self._init_model(state_dict)
        self.T = len(subgraphs)
        if self.T == 0:
            raise ValueError("empty batch")
        self.n = np.asarray([s.adj.shape[0] for s in subgraphs], np.int32)
        for s in subgraphs:
            a = np.asarray(s.adj)
            if a.shape[0] != a.shape[1] or s.feat.shape != (a.shape[0], self.D):
                raise ValueError("bad sub-graph shapes")
            if not np.array_equal(a, a.T):
                raise NotImplementedError("the HIP path requires a symmetric adjacency (all reference datasets are undirected)")
        rows = np.asarray([0 if graph_mode else s.target_row for s in subgraphs], np.int32)
        labels = np.asarray([s.gt_label for s in subgraphs], np.int32)
        self._create_plan(rows, labels)
        self._alloc_device()
        self._pack(subgraphs)
        if analyze:
            self.analyze()

    def _init_model(self, state_dict):
        self.w = model_arrays(state_dict)
        self.D, self.H = self.w["conv_first.weight"].shape
        self.O = self.w["conv_last.weight"].shape[1]
        self.C = self.w["pred_model.weight"].shape[0]
        if self.w["conv_block.0.weight"].shape != (self.H, self.H) or self.w["conv_last.weight"].shape[0] != self.H:
            raise NotImplementedError("unexpected encoder shapes")
        if self.w["pred_model.weight"].shape[1] != 2 * self.H + self.O:
            raise NotImplementedError("only concat=True prediction heads are supported")
        # method="att" (models.py:36-37, 62-68): the three att_weight matrices, zero padded to [3][32][32] -> every run takes k_att
        self.att = None
        att_keys = [k + ".att_weight" for k in ("conv_first", "conv_block.0", "conv_last")]
        if any(k in state_dict for k in att_keys):
            if not all(k in state_dict for k in att_keys):
                raise NotImplementedError("an encoder with attention weights in some layers only")
            self.att = np.zeros((3, FEAT_STRIDE, FEAT_STRIDE), np.float32)
            for l, (k, d) in enumerate(zip(att_keys, (self.D, self.H, self.H))):
                v = state_dict[k]
                v = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
                if v.shape != (d, d):
                    raise NotImplementedError(f"{k}: expected [{d}, {d}], got {list(v.shape)}")
                self.att[l, :d, :d] = v

    def _create_plan(self, rows, labels):
        prob = _Problem(self.T, self.n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), self.D, self.H, self.O, self.C,
                        int(self.graph_mode), int(getattr(self, "mask_relu", False)), int(getattr(self, "bn", False)))
        mdl = _Model()
        for l, k in enumerate(("conv_first", "conv_block.0", "conv_last")):
            mdl.W[l] = _fptr(self.w[k + ".weight"])
            mdl.b[l] = _fptr(self.w[k + ".bias"])
        mdl.Wp = _fptr(self.w["pred_model.weight"])
        mdl.bp = _fptr(self.w["pred_model.bias"])
        self.handle = ctypes.c_void_p()
        _check(self.lib, self.lib.gnnx_plan_create(ctypes.byref(prob), ctypes.byref(mdl), ctypes.byref(self.handle)))
        if getattr(self, "att", None) is not None:
            _check(self.lib, self.lib.gnnx_set_att_weights(self.handle, _fptr(self.att)))
        self.Q = int(self.lib.gnnx_total_q(self.handle))
        self.R = int(self.lib.gnnx_total_rows(self.handle))
        self.ld = np.zeros(self.T, np.int32)
        self.offQ = np.zeros(self.T, np.int64)
        self.offR = np.zeros(self.T, np.int64)
        _check(self.lib, self.lib.gnnx_get_layout(self.handle, self.ld.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                                  self.offQ.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                                                  self.offR.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        self.ws_bytes = int(self.lib.gnnx_workspace_bytes(self.handle))

    def _alloc_device(self):
        dev = self.device
        f32 = dict(dtype=torch.float32, device=dev)
        self.A = torch.empty(self.Q, **f32)
        self.X = torch.empty(self.R, FEAT_STRIDE, **f32)
        self.yhat = torch.empty(self.R, **f32)
        self.M = torch.empty(self.Q, **f32)
        self.Abar = torch.empty(self.Q, **f32)
        self.fmask = torch.empty(self.T, FEAT_STRIDE, **f32)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.loss = None
        # the jobs of a process run on ONE non-default stream per device (hipGraph capture is illegal on the legacy default
        # stream): a fresh stream per job costs a hardware-queue set-up at its first use (~5 ms, measured in the plan + pack time of
        # every batch), and the library's own side lanes (gnnx_capi.hip: lane_stream) are process-wide for the same reason
        self.stream = _engine_stream(dev) if dev.type == _DEVICE_TYPE else None

    # -- packing -------------------------------------------------------------------------------
    def _square_views(self, flat):
        return [flat[o:o + l * l].reshape(l, l) for o, l in zip(self.offQ, self.ld)]

    def _pack(self, subgraphs):
        A = np.zeros(self.Q, np.float32)
        X = np.zeros((self.R, FEAT_STRIDE), np.float32)
        yhat = np.zeros(self.R, np.float32)
        for s, v, r, n in zip(subgraphs, self._square_views(A), self.offR, self.n):
            v[:n, :n] = s.adj
            X[r:r + n, :self.D] = s.feat
            if not self.graph_mode:
                if s.pred_label is None:
                    raise ValueError("node mode needs pred_label for the Laplacian term (explain.py:780-793)")
                yhat[r:r + n] = s.pred_label
        self.A.copy_(torch.from_numpy(A))
        self.X.copy_(torch.from_numpy(X))
        self.yhat.copy_(torch.from_numpy(yhat))

    def set_masks(self, masks: Sequence[np.ndarray]):
        """Upload the initial edge masks (host-generated so the torch CPU RNG stream matches the reference)."""
        M = np.zeros(self.Q, np.float32)
        for m, v, n in zip(masks, self._square_views(M), self.n):
            v[:n, :n] = m
        self.M.copy_(torch.from_numpy(M), non_blocking=False)
        self._M_on_edges_only = False

    def set_masks_raw(self, raw: torch.Tensor):
        """Upload the initial edge masks as the host generator produced them - ONE contiguous n x n draw per target,
        concatenated in target order (init_edge_masks_raw) - and spread them over the padded layout on the device
        (gnnx_scatter_masks): one H2D copy, no host-side packing."""
        if raw.dtype != torch.float32 or raw.numel() != int(self.lib.gnnx_total_raw(self.handle)):
            raise ValueError("raw mask stream must hold sum(n^2) float32 values")
        self._raw = raw.to(self.device, non_blocking=True)
        if self._raw.data_ptr() == raw.data_ptr():      # host "device" (the emulator): .to() made no copy and the caller may reuse its buffer
            self._raw = raw.clone()
        c = _PIN_CACHE
        if c["buf"] is not None and raw.is_pinned() and raw.untyped_storage().data_ptr() == c["buf"].untyped_storage().data_ptr():
            c["event"] = torch.cuda.Event()           # the shared pinned buffer may be refilled once this copy is done
            c["event"].record(torch.cuda.current_stream(self.device))
        self._enter()
        _check(self.lib, self.lib.gnnx_scatter_masks(self.handle, self._raw.data_ptr(), self.M.data_ptr(), self._stream()))
        self._M_on_edges_only = False
        self._leave()

    def set_masks_on_edges(self, vals: torch.Tensor):
        """Upload the initial masks given on the edges only (init_edge_masks_on_edges: [E, 2] in the order of the edge layout) and place them
        in M.  The other entries of M are NOT initialised: only for jobs whose every target runs on an edge-sparse kernel (route 4-8) and
        whose results leave as edge lists - those kernels read M nowhere else."""
        self._edge_layout()
        E = int(self._eoff[-1])
        if vals.dtype != torch.float32 or vals.numel() != 2 * E:
            raise ValueError("vals must hold 2 E float32 values")
        v = vals.view(-1, 2).to(self.device, non_blocking=True)
        pos = self._epos[:E]
        self.M.index_put_((pos[:, 0],), v[:, 0])
        self.M.index_put_((pos[:, 1],), v[:, 1])
        self._edge_vals0 = v
        self._M_on_edges_only = True      # launch() refuses the kernels that read M off the edges

    def draw_edge_words_device(self, seeds):
        """First half of the seeded initial masks with the engine walked on the DEVICE (gnnx_mt_edge_words): enqueue the walk of every
        target's mt19937 stream (its raw state words land in the Abar array, which nothing reads before the run) and the gather of the two
        raw words of every directed edge entry's Box-Muller pair -> device uint32 [E, 4] (view of an int32 tensor).  The caller copies them
        to the host and finishes with transform_edge_words + set_masks_on_edges."""
        self._edge_layout()
        E = int(self._eoff[-1])
        sd = _h2d(np.ascontiguousarray(np.asarray(seeds).astype(np.int64)), self.device)
        words = torch.empty(max(E, 1), 4, dtype=torch.int32, device=self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_mt_edge_words(self.handle, sd.data_ptr(), self._eoff_d.data_ptr(), self._rc.data_ptr(), self.Abar.data_ptr(),
                                                     words.data_ptr(), self._stream()))
        self._leave()
        self._seeds_keepalive = sd
        return words[:E]

    def set_masks_raw_resident(self):
        """Reset M to the initial masks from the RNG stream uploaded by the last set_masks_raw (a device-only op)."""
        self._enter()
        _check(self.lib, self.lib.gnnx_scatter_masks(self.handle, self._raw.data_ptr(), self.M.data_ptr(), self._stream()))
        self._M_on_edges_only = False
        self._leave()

    def reset_masks(self):
        """Reset M to the initial masks this job was given last - the resident RNG stream (set_masks_raw) or the values on the edges
        (set_masks_on_edges) - without another host draw or upload: what a repeated run of the same batch starts from."""
        if getattr(self, "_M_on_edges_only", False):
            E = int(self._eoff[-1])
            pos, v = self._epos[:E], self._edge_vals0
            self._enter()          # (ordered with the job's stream like every other writer of M: ADVICE r5)
            with torch.cuda.stream(self.stream) if self.stream is not None else _nullcontext():
                self.M.index_put_((pos[:, 0],), v[:, 0])
                self.M.index_put_((pos[:, 1],), v[:, 1])
            self._leave()
        else:
            self.set_masks_raw_resident()

    def _edge_layout(self):
        dev = self.device
        if getattr(self, "_eoff", None) is None:      # the edge structure of the batch is fixed: count once
            counts = getattr(self, "_edge_counts", None)
            cached = counts is not None               # analyze() counted on this adjacency and left the row starts on the device
            if not cached:
                counts_d = torch.empty(self.T, dtype=torch.int64, device=dev)
                self._enter()
                _check(self.lib, self.lib.gnnx_edge_counts(self.handle, self.A.data_ptr(), counts_d.data_ptr(), self._stream()))
                self._leave()
                counts = counts_d.cpu().numpy()
            eoff = np.zeros(self.T + 1, np.int64)
            np.cumsum(counts, out=eoff[1:])
            self._eoff, self._eoff_d = eoff, _h2d(eoff, dev)
            E = max(int(eoff[-1]), 1)
            self._rc = torch.empty(E, 2, dtype=torch.int32, device=dev)
            self._ev = torch.empty(E, dtype=torch.float32, device=dev)
            self._em = torch.empty(E, 2, dtype=torch.float32, device=dev)
            self._epos = torch.empty(E, 2, dtype=torch.int64, device=dev)
            self._enter()        # the (fixed) edge structure, once: rc and the position of every edge in the packed arrays
            if cached:
                _check(self.lib, self.lib.gnnx_edge_layout(self.handle, self.A.data_ptr(), self._eoff_d.data_ptr(), self._rc.data_ptr(),
                                                           self._epos.data_ptr(), self._stream()))
            else:
                _check(self.lib, self.lib.gnnx_edge_positions(self.handle, self.A.data_ptr(), self._eoff_d.data_ptr(), self._rc.data_ptr(),
                                                              self._epos.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self._stream()))
            self._leave()

    def gather_edges_device(self, with_mask=False) -> torch.Tensor:
        """Masked adjacency of the last forward on the upper-triangle edges of every target, as ONE device tensor [E]
        (asynchronous; what a multi-GPU gather ships instead of dense blocks)."""
        self._edge_layout()
        self._enter()
        _check(self.lib, self.lib.gnnx_gather_values(self._epos.data_ptr(), int(self._eoff[-1]), self.Abar.data_ptr(), self.M.data_ptr(),
                                                     self._ev.data_ptr(), self._em.data_ptr() if with_mask else None, self._stream()))
        self._leave()
        return self._ev[:int(self._eoff[-1])]

    def denoise(self, threshold_num=20, vals: Optional[torch.Tensor] = None):
        """io_utils.denoise_graph(masked_adj, node_idx, threshold_num=k, max_component=True) (utils/io_utils.py:193-245;
        explain.py:306-308) for every target of the batch, on the device (gnnx_denoise_edges), applied to the edge values of
        the last run (or to `vals`, a device tensor laid out like gather_edges_device()).
        -> (keep uint8 [E]: edge is in the denoised explanation, threshold [T], stats [T, 3]: nodes, edges, smallest node)."""
        if vals is None:
            vals = self.gather_edges_device()
        self._edge_layout()
        E = max(int(self._eoff[-1]), 1)
        keep = torch.zeros(E, dtype=torch.uint8, device=self.device)
        thr = torch.empty(self.T, dtype=torch.float32, device=self.device)
        stats = torch.empty(self.T, 3, dtype=torch.int32, device=self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_denoise_edges(self.handle, self._eoff_d.data_ptr(), self._rc.data_ptr(), vals.data_ptr(), int(threshold_num),
                                                     keep.data_ptr(), thr.data_ptr(), stats.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                     self._stream()))
        self._leave()
        return keep.cpu().numpy()[:int(self._eoff[-1])].astype(bool), thr.cpu().numpy(), stats.cpu().numpy()

    def auc(self, real, vals: Optional[torch.Tensor] = None):
        """ROC-AUC of the batch's edge scores against 0/1 ground truth `real` [E] (explain.py:325-328) from exact pair counts
        computed on the device (gnnx_auc_counts).  -> (auc, positives, negatives)"""
        self._edge_layout()
        if vals is None:
            vals = self.gather_edges_device()
        E = int(self._eoff[-1])
        if len(real) != E or vals.numel() < E:
            raise ValueError("auc: `real` / `vals` must hold one entry per upper-triangle edge of the batch (%d)" % E)
        real_d = torch.from_numpy(np.ascontiguousarray(real, np.uint8)).to(self.device)
        scratch = torch.empty(max(E, 1), dtype=torch.float32, device=self.device)
        counts = torch.zeros(4, dtype=torch.int64, device=self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_auc_counts(vals.data_ptr(), real_d.data_ptr(), E, scratch.data_ptr(), counts.data_ptr(), self._stream()))
        self._leave()
        P, gt, eq, N = (int(x) for x in counts.cpu().numpy())
        if P == 0 or N == 0:
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")   # sklearn's message
        return (gt + 0.5 * eq) / (float(P) * float(N)), P, N

    def fetch_edges(self, with_mask=False) -> EdgeMasks:
        """The result as edge lists (gnnx_edge_counts / gnnx_gather_edges): only the live entries cross PCIe."""
        self.gather_edges_device(with_mask)
        E = int(self._eoff[-1])
        return EdgeMasks(self.n.copy(), self._eoff, self._rc.cpu().numpy()[:E], self._ev.cpu().numpy()[:E],
                         self.fmask.cpu().numpy()[:, :self.D].copy(), self._em.cpu().numpy()[:E] if with_mask else None)

    def _stream(self):
        return ctypes.c_void_p(self.stream.cuda_stream if self.stream is not None else 0)

    def use_stream(self, stream):
        """Move the job's later work to another stream (the caller orders the two streams, e.g. with an event)."""
        self.stream = stream

    def _enter(self):
        if self.stream is not None:
            cur = torch.cuda.current_stream(self.device)
            if cur != self.stream:
                self.stream.wait_stream(cur)

    def _leave(self):
        if self.stream is not None:
            cur = torch.cuda.current_stream(self.device)
            if cur != self.stream:
                cur.wait_stream(self.stream)

    # -- the hot loop --------------------------------------------------------------------------
    def launch(self, hyper: Hyper, state: Optional[AdamState] = None, keep_state=False, trace=False):
        """Enqueue the whole optimisation on the current stream (asynchronous).  `state`: continue from an optimiser state
        (gnnx_run_resume: self.M holds the mask after state.first_iter steps); `keep_state`: also hand the state after the run
        back (self.state_out, an AdamState whose first_iter counts the steps taken so far); `trace`: record the decision trace of
        every iteration (gnnx_set_trace; fetch_trace())."""
        if trace:
            i32 = dict(dtype=torch.int32, device=self.device)
            self.trace_gates = torch.empty(hyper.num_iters, self.R, 2, **i32)
            self.trace_pool = torch.empty(self.T, hyper.num_iters, 96, **i32) if self.graph_mode else None
            _check(self.lib, self.lib.gnnx_set_trace(self.handle, self.trace_gates.data_ptr(),
                                                     None if self.trace_pool is None else self.trace_pool.data_ptr()))
        if hyper.record_loss and (self.loss is None or self.loss.shape[1] != hyper.num_iters):
            self.loss = torch.empty(self.T, hyper.num_iters, LOSS_TERMS, dtype=torch.float32, device=self.device)
        hy = hyper.c()
        loss_ptr = self.loss.data_ptr() if hyper.record_loss else None
        rs = None
        if state is not None or keep_state:
            st = state if state is not None else AdamState()
            ptr = lambda x: None if x is None else x.data_ptr()
            rs = _Resume(int(st.first_iter), ptr(st.m), ptr(st.v), ptr(st.feat), None, None, None)
            if keep_state:
                f32 = dict(dtype=torch.float32, device=self.device)
                out = AdamState(int(st.first_iter) + int(hyper.num_iters), torch.zeros(self.Q, **f32), torch.zeros(self.Q, **f32),
                                torch.zeros(self.T, 3, FEAT_STRIDE, **f32))
                rs.m_out, rs.v_out, rs.feat_out = out.m.data_ptr(), out.v.data_ptr(), out.feat.data_ptr()
                self.state_out = out
            self._state_keepalive = st
        if getattr(self, "_M_on_edges_only", False) and (not hyper.use_resident or hyper.record_loss or
                                                         not np.isin(self.route(), (4, 5, 6, 7, 8)).all()):
            # set_masks_on_edges left every entry of M off the edges uninitialised: only the edge-sparse kernels may run on it
            raise ValueError("this job's initial masks were given on the edges only (set_masks_on_edges): it needs the edge-sparse resident kernels "
                             "(Hyper.use_resident, no loss logging, every target on routes 4-8)")
        self._enter()
        try:
            _check(self.lib, self.lib.gnnx_run_resume(self.handle, ctypes.byref(hy), ctypes.byref(rs) if rs is not None else None,
                                                      self.A.data_ptr(), self.X.data_ptr(), self.yhat.data_ptr(), self.M.data_ptr(),
                                                      self.Abar.data_ptr(), self.fmask.data_ptr(), loss_ptr, self.ws.data_ptr(),
                                                      self.ws_bytes, self._stream()))
        finally:
            if trace:      # the handle must not keep the trace pointers when the run raised (later launches would fail or write into this buffer)
                self.lib.gnnx_set_trace(self.handle, None, None)
        self._leave()

    def fetch_trace(self):
        """The decision trace of the last launch(trace=True): (gates, pool).  gates[k] = uint32 [iters, n_k, 2] for target k - bit c of
        word (iter, row, l) says that the forward of that iteration found U_{l+1}[row][c] > 0 (the ReLU gates of models.py:241, 251;
        rows outside the layer's row set - beyond two hops / one hop of the target in node mode - read 0); pool = int32
        [T, iters, 3, 32]: the row every graph-mode max-pool picked (models.py:283, 291, 300; -1 = no such column), or None."""
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        g = self.trace_gates.cpu().numpy().view(np.uint32)
        gates = [g[:, o:o + n, :].copy() for o, n in zip(self.offR, self.n)]
        pool = None if self.trace_pool is None else self.trace_pool.cpu().numpy().reshape(self.T, -1, 3, 32)
        return gates, pool

    def set_state_edges(self, first_iter, mask_rc, m_rc, v_rc, feat=None, feat_m=None, feat_v=None) -> AdamState:
        """Load an optimiser state given on the EDGES of the batch (the layout of fetch_edges: [E, 2] = entry (r, c), entry (c, r)
        per upper-triangle edge; feat* [T, D]): the mask entries go into self.M (whose other entries keep what set_masks* put
        there - they never reach an output), the moments into fresh packed arrays.  -> the AdamState for launch(state=...)."""
        self._edge_layout()
        E = int(self._eoff[-1])
        dev = self.device
        pos = self._epos[:E]
        f32 = dict(dtype=torch.float32, device=dev)
        to = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(dev)
        st = AdamState(int(first_iter), torch.zeros(self.Q, **f32), torch.zeros(self.Q, **f32), torch.zeros(self.T, 3, FEAT_STRIDE, **f32))
        for dst, src in ((self.M, mask_rc), (st.m, m_rc), (st.v, v_rc)):
            if src is None or E == 0:
                continue
            src = to(src)
            dst.index_put_((pos[:, 0],), src[:, 0])
            dst.index_put_((pos[:, 1],), src[:, 1])
        for k, src in enumerate((feat, feat_m, feat_v)):
            if src is not None:
                st.feat[:, k, :self.D] = to(src)
        return st

    def fetch_state_edges(self):
        """The optimiser state after a launch(keep_state=True) on the edges of the batch: (mask_rc, m_rc, v_rc [E, 2], feat [T, 3, D])."""
        self._edge_layout()
        E = int(self._eoff[-1])
        pos = self._epos[:E]
        g = lambda a: torch.stack([a[pos[:, 0]], a[pos[:, 1]]], 1).cpu().numpy()
        st = self.state_out
        return g(self.M), g(st.m), g(st.v), st.feat[:, :, :self.D].cpu().numpy()

    def fetch(self, hyper: Hyper) -> JobResult:
        if hyper.edge_results_only:
            raise ValueError("this run wrote its result on the edges only (Hyper.edge_results_only): use fetch_edges()")
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        Abar = self.Abar.cpu().numpy()
        M = self.M.cpu().numpy()
        ma = [v[:n, :n].copy() for v, n in zip(self._square_views(Abar), self.n)]
        mk = [v[:n, :n].copy() for v, n in zip(self._square_views(M), self.n)]
        loss = self.loss.cpu().numpy() if hyper.record_loss else None
        if loss is not None:      # the dense streaming kernels log the density as numerator [6] / denominator [7] (their tiles finish in no order)
            den = loss[:, :, 7]
            part = den != 0
            loss[:, :, 5] = np.where(part & (loss[:, :, 5] == 0), loss[:, :, 6] / np.where(part, den, 1.0), loss[:, :, 5])
        return JobResult(ma, mk, self.fmask.cpu().numpy()[:, :self.D].copy(), loss)

    def run(self, masks, hyper: Hyper) -> JobResult:
        self.set_masks(masks)
        self.launch(hyper)
        return self.fetch(hyper)

    def forward(self, masks, feat_mask=None):
        """One forward under the given masks: (softmax probs [T, C], masked_adj list). ExplainModule.forward."""
        self.set_masks(masks)
        probs = torch.empty(self.T, MAX_CLASSES, dtype=torch.float32, device=self.device)
        fm = None
        if feat_mask is not None:
            f = np.zeros((self.T, FEAT_STRIDE), np.float32)
            f[:, :self.D] = feat_mask
            fm = torch.from_numpy(f).to(self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_forward(self.handle, self.A.data_ptr(), self.X.data_ptr(), self.M.data_ptr(),
                                               fm.data_ptr() if fm is not None else None, self.Abar.data_ptr(),
                                               probs.data_ptr(), self.ws.data_ptr(), self.ws_bytes, self._stream()))
        self._leave()
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        Abar = self.Abar.cpu().numpy()
        ma = [v[:n, :n].copy() for v, n in zip(self._square_views(Abar), self.n)]
        return probs.cpu().numpy()[:, :self.C], ma

    def grad_baseline(self):
        """The reference's gradient baseline (`model="grad"`, explain.py:125-133, 717-738) for every target of the batch:
        sigmoid(|dL/dA| + |dL/dA|^T) * A of the UNMASKED sub-graph, loss = -log softmax(logits[target])[label] with the
        plan's labels (the caller passes the PREDICTED label of each target).  -> list of [n, n] float32."""
        out = torch.empty(self.Q, dtype=torch.float32, device=self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_grad_baseline(self.handle, self.A.data_ptr(), self.X.data_ptr(), out.data_ptr(), self.ws.data_ptr(),
                                                     self.ws_bytes, self._stream()))
        self._leave()
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        o = out.cpu().numpy()
        return [v[:n, :n].copy() for v, n in zip(self._square_views(o), self.n)]

    def time_kernel(self, hyper: Hyper, kind: int, reps: int):
        """(avg ms per launch, algorithmic bytes, algorithmic flops) of one kernel class — bench.py's roofline."""
        ms = ctypes.c_float()
        by = ctypes.c_double()
        fl = ctypes.c_double()
        hy = hyper.c()
        self._enter()
        _check(self.lib, self.lib.gnnx_time_kernel(self.handle, ctypes.byref(hy), kind, reps, self.A.data_ptr(),
                                                   self.X.data_ptr(), self.yhat.data_ptr(), self.M.data_ptr(),
                                                   self.Abar.data_ptr(), self.ws.data_ptr(), self.ws_bytes,
                                                   self._stream(), ctypes.byref(ms), ctypes.byref(by), ctypes.byref(fl)))
        self._leave()
        return ms.value, by.value, fl.value

    def resident_times(self):
        """In-situ device time (ms) of the resident launches of the last launch():
        [dense nb=1, nb=2, nb=3, sparse 1024-thread class, sparse 256, sparse 64, sparse large, sparse 512]."""
        ms = (ctypes.c_float * 8)()
        _check(self.lib, self.lib.gnnx_resident_times(self.handle, ms))
        return [float(x) for x in ms]

    def tiny_pack(self):
        """(targets per compute unit, targets) of the packed single-wave launch of this plan - (0, 0) when it has none (include/gnnx.h)."""
        per_cu, cnt = ctypes.c_int32(), ctypes.c_int32()
        _check(self.lib, self.lib.gnnx_tiny_pack_info(self.handle, ctypes.byref(per_cu), ctypes.byref(cnt), None))
        return int(per_cu.value), int(cnt.value)

    def tiny_packed(self):
        """bool [T]: the targets the packed single-wave launch takes."""
        f = np.zeros(self.T, np.int32)
        _check(self.lib, self.lib.gnnx_tiny_pack_info(self.handle, None, None, f.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return f.astype(bool)

    def route(self):
        """Kernel of every target: 0 dense streaming, 1..3 dense resident (row blocks), 4 / 5 / 6 sparse resident
        (1024- / 256- / 64-thread size class), 7 sparse kernel for larger targets (row arrays in the workspace), 8 sparse
        resident, 512-thread class."""
        r = np.zeros(self.T, np.int32)
        _check(self.lib, self.lib.gnnx_get_route(self.handle, r.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return r

    @property
    def sum_n2(self):
        return float((self.n.astype(np.float64) ** 2).sum())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gnnx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class XLJob:
    """A batch of node-mode targets of ANY size on the XL route (include/gnnx.h: gnnx_xl_*): what Explainer.explain does per target
    (explain.py:80-117 sub-graph slicing + ExplainModule, :137-146 the loop, :208-211 the result) with every sub-graph built ON THE DEVICE as a CSR of
    local ids from the resident full-graph CSR and the k-hop lists - no dense n x n block exists on the host or the device - and masks, optimiser state
    and results as edge lists (engine.EdgeMasks).  Same kernel arithmetic as MaskOptimJob's route 7 (k_sparse_large), for the targets that route
    cannot take: n > 16 383 (9.6 GB per dense array at the BA-House x100k maximum) or more entries within two hops than its LDS holds."""

    def __init__(self, graph: DeviceGraph, neighbors, target_rows, gt_labels, state_dict, lib=None):
        self.lib = lib if lib is not None else get_library()
        self.device = graph.feat.device
        self.graph = graph
        self.graph_mode = False
        MaskOptimJob._init_model(self, state_dict)
        if self.att is not None:
            raise NotImplementedError("method='att' has no XL form")
        if graph.feat.shape[1] != self.D:
            raise ValueError("feature width does not match the encoder")
        on_device = isinstance(neighbors, DeviceNeighbors)
        self.T = len(neighbors)
        if self.T == 0:
            raise ValueError("empty batch")
        self.n = np.ascontiguousarray(neighbors.sizes, np.int32) if on_device else np.asarray([len(nb) for nb in neighbors], np.int32)
        if on_device and target_rows is None:
            target_rows = neighbors.rows
        rows = np.ascontiguousarray(target_rows, np.int32)
        if (rows < 0).any():
            raise IndexError("targets %s are not in their own k-hop walk sets (isolated nodes?)" % np.nonzero(rows < 0)[0][:8].tolist())
        labels = np.ascontiguousarray(gt_labels, np.int32)
        prob = _Problem(self.T, self.n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), rows.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                        labels.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), self.D, self.H, self.O, self.C, 0, 0, 0)
        mdl = _Model()
        for l, k in enumerate(("conv_first", "conv_block.0", "conv_last")):
            mdl.W[l] = _fptr(self.w[k + ".weight"])
            mdl.b[l] = _fptr(self.w[k + ".bias"])
        mdl.Wp = _fptr(self.w["pred_model.weight"])
        mdl.bp = _fptr(self.w["pred_model.bias"])
        self.handle = ctypes.c_void_p()
        _check(self.lib, self.lib.gnnx_xl_create(ctypes.byref(prob), ctypes.byref(mdl), ctypes.byref(self.handle)))
        dev = self.device
        self.stream = _engine_stream(dev) if dev.type == _DEVICE_TYPE else None
        if on_device:
            nb_flat, nb_off_d = neighbors.nb_flat, neighbors.nb_off
        else:
            off = np.zeros(self.T + 1, np.int64)
            np.cumsum(self.n, out=off[1:])
            nb_flat = torch.from_numpy(np.concatenate([np.asarray(nb) for nb in neighbors]).astype(np.int32)).to(dev)
            nb_off_d = torch.from_numpy(off).to(dev)
        self._nb = (nb_flat, nb_off_d)
        self.R = int(self.lib.gnnx_xl_total_rows(self.handle))
        self.ld = np.zeros(self.T, np.int32)
        self.offR = np.zeros(self.T, np.int64)
        self.ws_rows = torch.empty(int(self.lib.gnnx_xl_rows_bytes(self.handle)), dtype=torch.uint8, device=dev)
        g = graph
        gargs = (g.indptr.data_ptr(), g.indices.data_ptr(), g.weights.data_ptr() if g.weights is not None else None, nb_flat.data_ptr(), nb_off_d.data_ptr())
        counts = np.zeros(self.T, np.int64)
        MaskOptimJob._enter(self)
        _check(self.lib, self.lib.gnnx_xl_count(self.handle, *gargs, g.feat.data_ptr(), g.feat.shape[1], g.pred_label.data_ptr() if g.pred_label is not None else None,
                                                self.ws_rows.data_ptr(), self.ws_rows.numel(), counts.ctypes.data, self._stream()))      # synchronises
        self.E = int(self.lib.gnnx_xl_total_edges(self.handle))
        self._eoff = np.zeros(self.T + 1, np.int64)
        _check(self.lib, self.lib.gnnx_xl_get_layout(self.handle, self.ld.ctypes.data, self.offR.ctypes.data, self._eoff.ctypes.data))
        self.ws_entries = torch.empty(int(self.lib.gnnx_xl_entries_bytes(self.handle)), dtype=torch.uint8, device=dev)
        E1 = max(self.E, 1)
        self._rc = torch.empty(E1, 2, dtype=torch.int32, device=dev)
        _check(self.lib, self.lib.gnnx_xl_build(self.handle, *gargs, self.ws_rows.data_ptr(), self.ws_entries.data_ptr(), self.ws_entries.numel(),
                                                self._rc.data_ptr(), self._stream()))
        MaskOptimJob._leave(self)
        f32 = dict(dtype=torch.float32, device=dev)
        self.M_e = torch.empty(E1, 2, **f32)
        self._ev = torch.empty(E1, **f32)
        self.fmask = torch.empty(self.T, FEAT_STRIDE, **f32)
        self._M0 = None

    _stream = MaskOptimJob._stream
    _enter = MaskOptimJob._enter
    _leave = MaskOptimJob._leave
    use_stream = MaskOptimJob.use_stream

    @property
    def sum_n2(self):
        return float((self.n.astype(np.float64) ** 2).sum())

    def edge_ids(self):
        """(eoff [T + 1] host int64, rc [E, 2] DEVICE int32): the upper-triangle edges (r < c, local ids) of every target in row-major order"""
        return self._eoff, self._rc[:self.E]

    def set_masks_on_edges(self, vals: torch.Tensor):
        """The initial masks on the edges: [E, 2] = (M[r][c], M[c][r]) in the order of edge_ids() (engine.init_edge_masks_on_edges /
        transform_edge_words produce exactly that from the seeds)."""
        if vals.dtype != torch.float32 or vals.numel() != 2 * self.E:
            raise ValueError("vals must hold 2 E float32 values")
        self._M0 = vals.reshape(-1, 2).to(self.device, non_blocking=True).clone() if self.E else vals.reshape(0, 2)
        self.reset_masks()

    def set_masks_seeded(self, seeds, threads=None):
        """construct_edge_mask (explain.py:645-652) under the seed protocol - target k's n x n normal_ draw from a generator seeded with seeds[k] - on the
        edges only (the host walks the engine state and lets ATen transform the blocks that hold an edge entry: gnnx_host_draw_edge_masks)."""
        rc = self._rc[:self.E].cpu()
        vals = init_edge_masks_on_edges(self.n, seeds, self._eoff, rc, threads=threads or default_rng_threads(big=True))
        self.set_masks_on_edges(vals)

    def draw_edge_words_device(self, seeds):
        """First half of the seeded masks with the engine walked on the DEVICE (gnnx_xl_mt_edge_words: no n^2 scratch): -> device int32 [E, 4] raw
        engine words; finish with engine.transform_edge_words(self.n, seeds, eoff, rc, words_host) + set_masks_on_edges."""
        enable_mt_jump(self.lib)
        sd = _h2d(np.ascontiguousarray(np.asarray(seeds).astype(np.int64)), self.device)
        words = torch.empty(max(self.E, 1), 4, dtype=torch.int32, device=self.device)
        self._enter()
        _check(self.lib, self.lib.gnnx_xl_mt_edge_words(self.handle, sd.data_ptr(), self.ws_rows.data_ptr(), self.ws_entries.data_ptr(), words.data_ptr(), self._stream()))
        self._leave()
        self._seeds_keepalive = sd
        return words[:self.E]

    def set_masks_seeded_device(self, seeds, threads=None):
        """set_masks_seeded with the engine walk on the device: O(E) host work (ATen's own transform of the picked pairs), bit-identical."""
        if not pair_staging_ok():
            return self.set_masks_seeded(seeds, threads)
        words = self.draw_edge_words_device(seeds).cpu()
        rc = self._rc[:self.E].cpu()
        vals = transform_edge_words(self.n, seeds, self._eoff, rc, words.contiguous(), threads=threads or default_rng_threads(big=True))
        self.set_masks_on_edges(vals)

    def reset_masks(self):
        if self._M0 is None:
            raise ValueError("no initial masks set")
        if self.E:
            self.M_e[:self.E].copy_(self._M0)

    def launch(self, hyper: Hyper, state: Optional["XLState"] = None, keep_state=False, trace=False):
        """Enqueue all iterations of every target (ONE launch, asynchronous).  state / keep_state / trace as MaskOptimJob.launch."""
        if hyper.record_loss:
            raise NotImplementedError("the XL route has no loss logging")
        if trace:
            self.trace_gates = torch.empty(hyper.num_iters, self.R, 2, dtype=torch.int32, device=self.device)
            _check(self.lib, self.lib.gnnx_xl_set_trace(self.handle, self.trace_gates.data_ptr()))
        hy = hyper.c()
        ptr = lambda x: None if x is None else x.data_ptr()
        st = state if state is not None else XLState()
        xs = _XlState(int(st.first_iter), self.M_e.data_ptr(), ptr(st.m), ptr(st.v), ptr(st.feat), None, None, None)
        if keep_state:
            f32 = dict(dtype=torch.float32, device=self.device)
            out = XLState(int(st.first_iter) + int(hyper.num_iters), torch.zeros(max(self.E, 1), 2, **f32), torch.zeros(max(self.E, 1), 2, **f32),
                          torch.zeros(self.T, 3, FEAT_STRIDE, **f32))
            xs.m_out_e, xs.v_out_e, xs.feat_out = out.m.data_ptr(), out.v.data_ptr(), out.feat.data_ptr()
            self.state_out = out
        self._state_keepalive = st
        self._enter()
        try:
            _check(self.lib, self.lib.gnnx_xl_run(self.handle, ctypes.byref(hy), ctypes.byref(xs), self._ev.data_ptr(), self.fmask.data_ptr(),
                                                  self.ws_rows.data_ptr(), self.ws_entries.data_ptr(), self._stream()))
        finally:
            if trace:
                self.lib.gnnx_xl_set_trace(self.handle, None)
        self._leave()

    def gather_edges_device(self, with_mask=False) -> torch.Tensor:
        return self._ev[:self.E]

    def record_clocks(self, on=True):
        """Measurement hook (gnnx_xl_set_clocks): the launches that follow stamp every target's workgroup; target_ms() reads them."""
        self._clk = torch.zeros(self.T, 4, dtype=torch.int64, device=self.device) if on else None
        _check(self.lib, self.lib.gnnx_xl_set_clocks(self.handle, self._clk.data_ptr() if on else None))

    def target_ms(self):
        """[T, 3] milliseconds of the last launch per target: setup, iteration loop, results (far edges + write-back)"""
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        c = self._clk.cpu().numpy().astype(np.float64)
        return np.diff(c, axis=1) / 1e5

    def fetch_edges(self, with_mask=False) -> EdgeMasks:
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        return EdgeMasks(self.n.copy(), self._eoff, self._rc[:self.E].cpu().numpy(), self._ev[:self.E].cpu().numpy(),
                         self.fmask.cpu().numpy()[:, :self.D].copy(), self.M_e[:self.E].cpu().numpy() if with_mask else None)

    def set_state_edges(self, first_iter, mask_rc, m_rc, v_rc, feat=None, feat_m=None, feat_v=None) -> "XLState":
        """An optimiser state given on the edges ([E, 2] arrays in the order of edge_ids(); feat* [T, D]) -> the state for launch(state=...);
        the mask entries go into self.M_e."""
        to = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32)).to(self.device)
        st = XLState(int(first_iter), None, None, torch.zeros(self.T, 3, FEAT_STRIDE, dtype=torch.float32, device=self.device))
        if mask_rc is not None and self.E:
            self.M_e[:self.E].copy_(to(mask_rc))
        if m_rc is not None:
            st.m = to(m_rc).reshape(-1, 2)
        if v_rc is not None:
            st.v = to(v_rc).reshape(-1, 2)
        for k, src in enumerate((feat, feat_m, feat_v)):
            if src is not None:
                st.feat[:, k, :self.D] = to(src)
        return st

    def fetch_state_edges(self):
        """(mask_rc, m_rc, v_rc [E, 2], feat [T, 3, D]) after a launch(keep_state=True)"""
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        st = self.state_out
        return (self.M_e[:self.E].cpu().numpy(), st.m[:self.E].cpu().numpy(), st.v[:self.E].cpu().numpy(), st.feat[:, :, :self.D].cpu().numpy())

    def fetch_trace(self):
        if self.device.type == _DEVICE_TYPE:
            torch.cuda.synchronize(self.device)
        g = self.trace_gates.cpu().numpy().view(np.uint32)
        return [g[:, o:o + n, :].copy() for o, n in zip(self.offR, self.n)], None

    def close(self):
        if getattr(self, "handle", None):
            if self.device.type == _DEVICE_TYPE:
                torch.cuda.synchronize(self.device)      # the plan's tables go back to the pool: nothing may still read them
            self.lib.gnnx_xl_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_MT_JUMP_SET = set()


def enable_mt_jump(lib=None, poly=None, jump=None):
    """Hand the library the jump polynomial of the segmented engine walk (gnnx_set_mt_jump_poly; utils/mt_jump.py) - once per process and library.
    poly / jump: another stride (tests); jump=0 switches back to serial walks."""
    lib = lib if lib is not None else get_library()
    key = id(lib)
    if poly is None and jump is None:
        if key in _MT_JUMP_SET or os.environ.get("GNNX_MT_JUMP", "1") == "0":
            return
        from .utils import mt_jump
        poly, jump = mt_jump.load_jump_poly(), mt_jump.JUMP
    _MT_JUMP_SET.add(key)
    if not jump:
        _check(lib, lib.gnnx_set_mt_jump_poly(None, 0))
        return
    poly = np.ascontiguousarray(poly, np.uint32)
    assert poly.shape == (624,)
    from .utils import mt_jump
    levels = int(os.environ.get("GNNX_MT_JUMP_LEVELS", mt_jump.LEVELS))      # (1: one chain of K - 1 jumps per target, the first form)
    polys = np.ascontiguousarray(mt_jump.jump_polys(poly, levels), np.uint32)      # strides J, 4 J, 16 J: ~0.2 s, once per process
    _check(lib, lib.gnnx_set_mt_jump_polys(polys.ctypes.data, int(jump), levels))


@dataclass
class XLState:
    """Optimiser state of an XLJob batch on the edges: exp_avg / exp_avg_sq [E, 2] (None = zeros), feat [T, 3, 32]."""
    first_iter: int = 0
    m: Optional[torch.Tensor] = None
    v: Optional[torch.Tensor] = None
    feat: Optional[torch.Tensor] = None


def xl_job_from_subgraphs(subgraphs: Sequence[Subgraph], state_dict, device=None, lib=None) -> XLJob:
    """An XLJob whose targets are given as dense Subgraph objects (tests, small callers): the sub-graphs become the blocks of one
    block-diagonal "full graph" and every target's k-hop list is its own block."""
    import scipy.sparse as sp
    blocks, feats, preds, nbs, off = [], [], [], [], 0
    for sgr in subgraphs:
        a = np.asarray(sgr.adj, np.float32)
        blocks.append(sp.csr_matrix(a))
        feats.append(np.asarray(sgr.feat, np.float32))
        preds.append(np.asarray(sgr.pred_label, np.float32))
        nbs.append(np.arange(off, off + a.shape[0], dtype=np.int64))
        off += a.shape[0]
    csr = sp.block_diag(blocks, format="csr")
    feat = np.concatenate(feats)
    if device is None:
        device = torch.device(_DEVICE_TYPE, torch.cuda.current_device())
    g = device_graph(csr, feat, None, device=device)
    g.pred_label = torch.from_numpy(np.concatenate(preds)).to(g.feat.device)
    return XLJob(g, nbs, [s.target_row for s in subgraphs], [s.gt_label for s in subgraphs], state_dict, lib=lib)


_PIN_CACHE = {"buf": None, "event": None}
_RNG_POOL = []


def _pinned_stream_buffer(total):
    """One grow-only pinned host buffer per process for the RNG stream of a batch (allocating 8 MB of pinned memory costs about
    as much as drawing the syn1 masks).  MaskOptimJob.set_masks_raw records an event behind its H2D copy; the buffer is not
    rewritten before that copy has finished."""
    c = _PIN_CACHE
    if c["event"] is not None:
        c["event"].synchronize()
        c["event"] = None
    if c["buf"] is None or c["buf"].numel() < total:
        c["buf"] = torch.empty(int(total * 1.25) + 1024, dtype=torch.float32, pin_memory=True)
    return c["buf"][:total]


def _rng_pool():
    if not _RNG_POOL:
        from concurrent.futures import ThreadPoolExecutor
        _RNG_POOL.append(ThreadPoolExecutor(32))
    return _RNG_POOL[0]


_HOST_LIB_NAME = "libgnnx_host.so"
_host_lib_cache = []


def host_library():
    """libgnnx_host.so (include/gnnx_host.h): ATen's CPU normal_ called from C++ threads, no GIL.  None if it is not built (the
    Python loop below then makes the very same ATen calls, bit for bit, only slower)."""
    if not _host_lib_cache:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", _HOST_LIB_NAME)
        lib = None
        if os.path.exists(path):
            lib = ctypes.CDLL(path)
            lib.gnnx_host_draw_masks.restype = ctypes.c_int
            lib.gnnx_host_draw_masks.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
            lib.gnnx_host_draw_masks_sliced.restype = ctypes.c_int
            lib.gnnx_host_draw_masks_sliced.argtypes = lib.gnnx_host_draw_masks.argtypes + [ctypes.c_int64]
            lib.gnnx_host_draw_edge_masks.restype = ctypes.c_int
            lib.gnnx_host_draw_edge_masks.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_int32, ctypes.c_int64]
            lib.gnnx_host_pair_staging_ok.restype = ctypes.c_int
            lib.gnnx_host_pair_staging_ok.argtypes = []
            lib.gnnx_host_transform_edge_words.restype = ctypes.c_int
            lib.gnnx_host_transform_edge_words.argtypes = [ctypes.c_int32] + [ctypes.c_void_p] * 6 + [ctypes.c_int32]
            lib.gnnx_host_last_error.restype = ctypes.c_char_p
        _host_lib_cache.append(lib)
    return _host_lib_cache[0]


def cpu_quota_cores():
    """cores the cgroup of this process may use (cpu.max = quota / period), or None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        return None


def default_rng_threads(big=False):
    """Host threads for the seeded mask draw: embarrassingly parallel over targets, but beyond ~32 threads the hand-off costs more than it
    saves (tools/probe_rng.py), and never more than half the cores the process may actually use (a container's CPU quota: the GPU box gives 16
    of its host's 256 CPUs - tools/probe_rng_big.py).  GNNX_RNG_THREADS overrides."""
    env = os.environ.get("GNNX_RNG_THREADS")
    if env:
        return max(1, int(env))
    cores = (os.cpu_count() or 2) // 2
    q = cpu_quota_cores()
    if q:      # half the quota: three preparing threads draw at once, and a cgroup that spends its quota inside a 100 ms period is frozen until the
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))      # (torchrun: the ranks of a node share the quota)
        cores = min(cores, max(2, int(q) // ((1 if big else 2) * ranks)))      # next one (profiles/r05_cpu_quota_throttling.txt; 2 x quota until round 5).
        # (big=True - batches of more than 2e7 normals, tens of milliseconds of draw: the whole quota; 75 M normals 36.6 ms on 8 threads, 23 on 32)
    return max(1, min(32, cores))


def init_edge_masks_raw(sizes, generator=None, seeds=None, pin=False, threads=1, out=None, slice_values=None):
    """The initial edge masks of a whole batch as ONE host buffer: target after target the n x n values of the single
    normal_(1, std) draw construct_edge_mask makes (explain.py:645-652), generated in place (a draw into a contiguous
    slice consumes the generator exactly like a draw into a fresh [n, n] tensor).  `seeds`: re-seed a PRIVATE generator
    before every target (the seed protocol of the golden runs) instead of consuming the caller's global stream; the
    targets are then independent and `threads` > 1 draws them on several host threads (normal_ releases the GIL).
    `pin`: the result is a view of ONE process-wide pinned buffer - upload it (set_masks_raw) before the next pinned call;
    `out`: draw into the caller's buffer instead; `slice_values`: slice length for large targets (gnnx_host_draw_masks_sliced; default 2^21)."""
    sizes = [int(n) for n in sizes]
    off = np.zeros(len(sizes) + 1, np.int64)
    np.cumsum(np.asarray(sizes, np.int64) ** 2, out=off[1:])
    if out is not None:        # the caller's own (e.g. pinned) staging buffer: pipeline.BatchPipeline keeps a ring of them
        if out.dtype != torch.float32 or out.numel() != int(off[-1]) or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 buffer of sum(n^2) values")
        buf = out
    else:
        buf = _pinned_stream_buffer(int(off[-1])) if pin else torch.empty(int(off[-1]), dtype=torch.float32)

    def fill(lo, hi, gen):
        for k in range(lo, hi):
            n = sizes[k]
            if seeds is not None:
                gen.manual_seed(int(seeds[k]))
            buf[off[k]:off[k + 1]].normal_(1.0, math.sqrt(2.0) * math.sqrt(2.0 / (n + n)), generator=gen)

    hl = host_library() if seeds is not None else None
    if hl is not None and len(sizes):       # the seed protocol makes targets independent: C++ threads, no GIL (gnnx_host_draw_masks)
        n32 = np.ascontiguousarray(sizes, np.int32)
        sd = np.ascontiguousarray(np.asarray(seeds).astype(np.int64))
        if hl.gnnx_host_draw_masks_sliced(len(sizes), n32.ctypes.data, sd.ctypes.data, off.ctypes.data, buf.data_ptr(), int(max(1, threads)),
                                          int(slice_values) if slice_values else 1 << 21) != 0:
            raise GnnxError(hl.gnnx_host_last_error().decode())
        return buf
    if seeds is None or threads <= 1 or len(sizes) < 2 * threads:
        fill(0, len(sizes), torch.Generator() if seeds is not None else generator)
        return buf
    cuts = [0] + [int(np.searchsorted(off, off[-1] * i / threads)) for i in range(1, threads)] + [len(sizes)]   # equal shares of the floats
    list(_rng_pool().map(lambda i: fill(cuts[i], cuts[i + 1], torch.Generator()), range(threads)))
    return buf


def init_edge_masks_on_edges(sizes, seeds, eoff, rc, threads=1, out=None, slice_values=1 << 17):
    """The initial edge masks of a batch on the EDGES only: out [E, 2] = (M[r][c], M[c][r]) for every upper-triangle edge (r, c) of every
    target (rc [E, 2] int32 local node ids, eoff [T + 1]) - exactly the values init_edge_masks_raw puts at those positions (the same ATen
    draws; gnnx_host_draw_edge_masks), without writing the sum(n^2) values in between.  The edge-sparse kernels read nothing else."""
    hl = host_library()
    if hl is None:
        raise GnnxError("libgnnx_host.so is not built")
    E = int(eoff[-1])
    if out is None:
        out = torch.empty(E, 2, dtype=torch.float32)
    if out.dtype != torch.float32 or out.numel() != 2 * E or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 buffer of 2 E values")
    n32 = np.ascontiguousarray(sizes, np.int32)
    sd = np.ascontiguousarray(np.asarray(seeds).astype(np.int64))
    eo = np.ascontiguousarray(eoff, np.int64)
    rc = rc if isinstance(rc, np.ndarray) else rc.numpy()
    rc = np.ascontiguousarray(rc[:E], np.int32)
    if E and hl.gnnx_host_draw_edge_masks(len(n32), n32.ctypes.data, sd.ctypes.data, eo.ctypes.data, rc.ctypes.data, out.data_ptr(), int(max(1, threads)),
                                          int(slice_values)) != 0:
        raise GnnxError(hl.gnnx_host_last_error().decode())
    return out


def pair_staging_ok():
    """True when the host's normal_ has the property the device-side walk rests on (every lane of its 16-value transform computes the same
    function of its own pair of uniforms: checked once per process against ATen's one-call draw by libgnnx_host.so)."""
    hl = host_library()
    return hl is not None and bool(hl.gnnx_host_pair_staging_ok())


def transform_edge_words(sizes, seeds, eoff, rc, words, threads=1, out=None):
    """Second half of the device-side draw: words [E, 4] (HOST int32 / uint32: MaskOptimJob.draw_edge_words_device) -> out [E, 2] =
    (M[r][c], M[c][r]), the values ATen's normal_ gives those raw engine words (gnnx_host_transform_edge_words) - bit-identical to
    init_edge_masks_on_edges, without a pass over the n^2 draws of any target."""
    hl = host_library()
    if hl is None:
        raise GnnxError("libgnnx_host.so is not built")
    E = int(eoff[-1])
    if out is None:
        out = torch.empty(E, 2, dtype=torch.float32)
    if out.dtype != torch.float32 or out.numel() != 2 * E or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 buffer of 2 E values")
    if words.numel() != 4 * E or not words.is_contiguous() or words.element_size() != 4:
        raise ValueError("words must be a contiguous 32-bit buffer of 4 E values")
    n32 = np.ascontiguousarray(sizes, np.int32)
    sd = np.ascontiguousarray(np.asarray(seeds).astype(np.int64))
    eo = np.ascontiguousarray(eoff, np.int64)
    rc = rc if isinstance(rc, np.ndarray) else rc.numpy()
    rc = np.ascontiguousarray(rc[:E], np.int32)
    if E and hl.gnnx_host_transform_edge_words(len(n32), n32.ctypes.data, sd.ctypes.data, eo.ctypes.data, rc.ctypes.data, words.data_ptr(),
                                               out.data_ptr(), int(max(1, threads))) != 0:
        raise GnnxError(hl.gnnx_host_last_error().decode())
    return out


def init_edge_mask(n, generator=None):
    """construct_edge_mask(init_strategy='normal') (explain.py:645-652): ONE normal_(1, std) draw of n*n
    values from the torch CPU generator, std = gain('relu') * sqrt(2 / (n + n))."""
    std = math.sqrt(2.0) * math.sqrt(2.0 / (n + n))
    m = torch.empty(n, n, dtype=torch.float32)
    m.normal_(1.0, std, generator=generator)
    return m.numpy()
