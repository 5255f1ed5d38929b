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


def target_cost(sizes) -> np.ndarray:
    """GPU time (microseconds) one target of n nodes adds to a batch that fills an MI355X, 300 iterations: the quantity the
    shards must balance.  Measured by size class on the BA-House x100k target set (tools/probe_classes.py, round 2): the
    edge-sparse kernels cost per workgroup slot, not per n^2 - 2.2-3.1 us for the one-wave class (n <= 32, six or seven
    targets per CU), ~11 us for the 256-thread class (n <= 128, two per CU), 14-20 us for the 512-thread class (one per CU,
    ~10-13 us per iteration whatever n), ~45 us for k_sparse_large (n ~ 900 on average; its iteration grows with the
    entries of the rows within two hops).  Targets beyond its range stream dense n x n blocks: ~28 n^2 bytes per iteration
    at ~4 TB/s."""
    n = np.asarray(sizes, np.float64)
    cost = np.where(n <= 32, 2.5, np.where(n <= 128, 11.0, np.where(n <= 512, 10.0 + 0.02 * n, 25.0 + 0.025 * n)))
    return np.where(n > 16383, 300 * 28.0 * n * n / 4e6, cost)


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


def run_sharded(targets: Sequence, costs: Sequence[float], compute: Callable[[List], List[np.ndarray]],
                group=None, gather_to_all: bool = True) -> Dict:
    """Every rank calls this with the SAME targets/costs.  `compute(list_of_targets)` runs this rank's shard
    (one batched GPU job) and returns one masked adjacency per target.  Returns {target: masked_adj} on every
    rank (or on rank 0 only when gather_to_all is False)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(zip(targets, compute(list(targets))))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = lpt_shards(costs, world)[rank]
    results = compute([targets[i] for i in mine]) if mine else []
    payload = [(i, sparse_pack(np.asarray(m))) for i, m in zip(mine, results)]
    if gather_to_all:
        gathered = [None] * world
        dist.all_gather_object(gathered, payload, group=group)
    else:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(payload, gathered, dst=0, group=group)
        if rank != 0:
            return {}
    out = {}
    for part in gathered:
        for i, packed in part:
            out[targets[i]] = sparse_unpack(packed)
    return out
