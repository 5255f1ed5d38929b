"""Multi-GPU target sharding (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Targets are independent: the only communication is distributing the target list and
collecting the finished masks — there is NO collective inside the 300-iteration loop.  Cost model: `target_cost`
(GPU time a target adds to a saturated batch, by kernel class); shards are balanced by
longest-processing-time-first, which matters because the size distribution is heavy-tailed (BA-House x100k: 74 %
of the targets have n <= 32 and cost 2.5 us each, the 757 with n > 512 cost ~45 us each).

The reference has no distributed code at all; this replaces the sequential
`[self.explain(i) for i in node_indices]` (explainer/explain.py:296-299) across devices.
"""
from typing import Callable, Dict, List, Sequence

import numpy as np


# Size classes of the cost model = the kernel classes of the plan (gnnx_plan_analyze; csrc/gnnx_sparse.hpp): one-wave targets
# (n <= 32), the 256-thread class (n <= 128), the 512- / 1024-thread classes (n <= 512), and the large-target kernel beyond
# (k_sparse_large / its XL form: any size since round 6 - no target streams dense blocks any more).
CLASS_EDGES = (32, 128, 512)
# GPU microseconds one target adds to a SATURATED batch of its class on one MI355X, 300 iterations.  Beyond 512 nodes the cost grows with
# the target: one workgroup (= one compute unit) per target for LARGE_MS(n) milliseconds - measured with the kernel's per-target device
# clocks on BA-House x100k (profiles/r06_probe_xl_ba100k_first.json): 6.8 ms at n = 700, 14 at 4 400, 27 at 11 500, 52 at 18 800, then faster than
# linearly (the sub-graphs of the hub's neighbourhood: 120 ms at 30 000, 480 at 48 000).  The table's fourth entry is the cost at n = 900.
# The defaults were measured on the BA-House x100k target set; `calibrate_cost_table` re-measures them on the workload and machine at
# hand, so the shards do not depend on one dataset's fit.
DEFAULT_COST_TABLE = np.asarray([2.5, 11.0, 17.0, 29.0], np.float64)


def large_ms(n):
    """compute-unit milliseconds of the large-target kernel for a sub-graph of n nodes (300 iterations; the shape of the fit above)"""
    n = np.asarray(n, np.float64)
    return (5.5 + 2.2e-3 * np.minimum(n, 2.0e4)) * np.maximum(1.0, n / 2.0e4) ** 2.5


def size_class(sizes) -> np.ndarray:
    return np.searchsorted(np.asarray(CLASS_EDGES), np.asarray(sizes), side="left")


def target_cost(sizes, table=None) -> np.ndarray:
    """GPU time (microseconds) one target of n nodes adds to a batch that fills an MI355X, 300 iterations: the quantity the shards
    must balance.  The edge-sparse kernels cost per workgroup slot, not per n^2 (every iteration is the same latency chain whatever
    n), so the model is one constant per kernel class (`table`, default DEFAULT_COST_TABLE; the 512-thread class scaled linearly with
    n around the class mean it was measured at, the large-target kernel by its measured shape large_ms)."""
    n = np.asarray(sizes, np.float64)
    t = DEFAULT_COST_TABLE if table is None else np.asarray(table, np.float64)
    c = size_class(n)
    return np.choose(np.minimum(c, 3), [t[0], t[1], t[2] * (0.6 + 0.4 * n / 320.0), t[3] * large_ms(n) / large_ms(900.0)])


def calibrate_cost_table(sizes, run_batch, per_class=1024, min_members=64) -> np.ndarray:
    """Measure the per-class constants of `target_cost` on this workload and machine.  `run_batch(indices) -> milliseconds` runs the
    targets `indices` (positions in `sizes`) as one batched job for the full iteration count.  A class with fewer than `min_members`
    targets cannot saturate the GPU and keeps its default.  Deterministic choice of the sample (the first `per_class` members)."""
    sizes = np.asarray(sizes)
    c = size_class(sizes)
    table = DEFAULT_COST_TABLE.copy()
    for k in range(4):
        idx = np.nonzero(c == k)[0][:per_class]
        if len(idx) < min_members:
            continue
        ms = float(run_batch(idx))
        unit = target_cost(sizes[idx], np.where(np.arange(4) == k, 1.0, DEFAULT_COST_TABLE)).sum()     # the class constant's multiplier
        table[k] = ms * 1e3 / unit
    return table


def lpt_shards(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first partition of item indices into world_size shards (deterministic)."""
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += float(costs[i])
    return [sorted(s) for s in shards]


def sparse_pack(masked_adj: np.ndarray):
    """Masks are exactly zero off the sub-graph's edges: ship only the non-zero entries."""
    r, c = np.nonzero(masked_adj)
    return masked_adj.shape[0], r.astype(np.int32), c.astype(np.int32), masked_adj[r, c]


def sparse_unpack(packed):
    n, r, c, v = packed
    out = np.zeros((n, n), v.dtype)
    out[r, c] = v
    return out


def _gather_variable(t, group, device):
    """all_gather of 1-D tensors of different lengths: sizes first, then the data padded to the longest (two collectives on the
    group's device - RCCL over xGMI with the nccl backend; no pickling through the host as all_gather_object does)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n, group=group)
    ns = [int(x.item()) for x in ns]
    m = max(max(ns), 1)
    buf = torch.zeros(m, dtype=t.dtype, device=device)
    buf[:t.numel()] = t.to(device)
    outs = [torch.empty(m, dtype=t.dtype, device=device) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [o[:k].cpu() for o, k in zip(outs, ns)]


def run_sharded(targets: Sequence, costs: Sequence[float], compute: Callable[[List], List[np.ndarray]],
                group=None, gather_to_all: bool = True) -> Dict:
    """Every rank calls this with the SAME targets/costs.  `compute(list_of_targets)` runs this rank's shard
    (one batched GPU job) and returns one masked adjacency per target.  Returns {target: masked_adj} on every
    rank.  The masks travel as their non-zero entries in two padded tensor all-gathers (indices, values)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(zip(targets, compute(list(targets))))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = lpt_shards(costs, world)[rank]
    results = compute([targets[i] for i in mine]) if mine else []
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    packed = [sparse_pack(np.asarray(m)) for m in results]
    dtype = packed[0][3].dtype if packed else np.float64
    # int32 stream: [k, then per target (index, n, nnz)], then all rows, then all columns; value stream: all values (as float64 bits)
    head = np.asarray([len(mine)] + [x for i, pk in zip(mine, packed) for x in (i, pk[0], len(pk[1]))], np.int32)
    ints = np.concatenate([head] + [pk[1] for pk in packed] + [pk[2] for pk in packed]).astype(np.int32)
    vals = np.concatenate([pk[3].astype(np.float64) for pk in packed]) if packed else np.zeros(0, np.float64)
    all_ints = _gather_variable(torch.from_numpy(ints), group, device)
    all_vals = _gather_variable(torch.from_numpy(vals), group, device)
    out = {}
    for it, vt in zip(all_ints, all_vals):
        it, vt = it.numpy(), vt.numpy()
        k = int(it[0])
        meta = it[1:1 + 3 * k].reshape(k, 3)
        nnz = meta[:, 2].astype(np.int64)
        off = np.concatenate([[0], np.cumsum(nnz)])
        rows, cols = it[1 + 3 * k:1 + 3 * k + off[-1]], it[1 + 3 * k + off[-1]:1 + 3 * k + 2 * off[-1]]
        for j in range(k):
            a, b = off[j], off[j + 1]
            out[targets[int(meta[j, 0])]] = sparse_unpack((int(meta[j, 1]), rows[a:b], cols[a:b], vt[a:b].astype(dtype)))
    return out
