"""MI355X-native GNNExplainer mask-optimisation engine (drop-in for the reference's
explainer/explain.py hot path).  See DESIGN.md."""
__version__ = "0.1.0"

import os as _os

# HIP binds every stream to one of GPU_MAX_HW_QUEUES hardware queues (default 4) when it is created, and packets of streams that
# share a queue execute in submission order - a barrier packet of one stream (waiting for a 4 ms optimisation launch) stalls the
# k-hop / packing kernels of the next batch queued behind it on another stream (measured: 4.1 ms instead of 0.3 ms for the k-hop
# pass of pipeline.BatchPipeline).  The engine uses up to eight streams per device (three launch lanes, prepare / optimise / fetch,
# the per-device engine stream, the caller's), so ask for eight queues - effective when set before the first HIP call.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (the runtime's upper limit)
# Host <-> device copies above GPU_FORCE_BLIT_COPY_SIZE KB (default 16) go to the SDMA engines, whose latency (tens of us per copy
# plus the signal round trip) is what a plan's table upload (a few hundred KB) and a batch's result download wait for: with the
# threshold at 1 MB those copies are blit kernels like the small ones - plan + pack + route of a syn1 batch 3.0-3.4 -> 1.6-1.9 ms
# inside the pipeline, end to end 112-141 k -> 139-164 k nodes/s (one session, alternating runs); 0 (everything on SDMA) gave
# 92-105 k.  The 8 MB of initial masks per batch stay on SDMA.  Effective when set before the first HIP call.
_os.environ.setdefault("GPU_FORCE_BLIT_COPY_SIZE", "1024")

# A container's CPU quota (cgroup cpu.max; the GPU boxes: 16 cores of a 256-CPU host).  torch sizes its intra-op pool from the CPU COUNT: one
# parallel CPU op wakes 128 OpenMP threads that spin after the region, the cgroup spends its 1.6 core-seconds within a few milliseconds and the
# kernel freezes EVERY thread of the process until the next 100 ms period - 20-batch timed regions of 169 k beside 250 k nodes/s
# (profiles/r05_cpu_quota_throttling.txt: nr_throttled rises in every run).  Before torch is imported: size the pool to the quota.
def _cpu_quota_cores():
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        return None


_q = _cpu_quota_cores()
if _q and _q < (_os.cpu_count() or 1):
    _ranks = max(1, int(_os.environ.get("LOCAL_WORLD_SIZE", "1")))      # (torchrun: the ranks of a node share the quota)
    for _v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        _os.environ.setdefault(_v, str(max(1, int(_q) // (2 * _ranks))))
    _os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

# One NUMA node.  The GPU boxes are two-socket hosts (2 x 64 cores, 256 CPUs) and the container's threads may run on any CPU: the pipeline's
# handful of threads (prepare workers, launch / fetch thread, RNG pool, HIP's own) migrate between the sockets, their pinned staging buffers
# and the runtime's queues sit on one.  Measured on syn1 (profiles/r05_cpu_affinity_numa.txt; the driver's command, five alternating runs):
# free 233-246 k nodes/s, prepare stage 2.65 ms, single repetitions down to 142 k; confined to EITHER node 254-264 k, prepare 1.9-2.2 ms.
# Called at import (before torch / HIP create their threads, which inherit it) when the process is in a quota-limited container and its
# affinity spans several nodes; GNNX_CPU_AFFINITY=0 leaves the affinity alone.  The ranks of a node spread over its NUMA nodes.
def confine_to_one_numa_node(node_root="/sys/devices/system/node", get_affinity=None, set_affinity=None):
    """(node_root / get_affinity / set_affinity: injected by tests/test_host_api.py)"""
    import glob as _glob
    if _os.environ.get("GNNX_CPU_AFFINITY", "1") == "0" or not hasattr(_os, "sched_setaffinity"):
        return None
    try:
        cur = set(get_affinity() if get_affinity else _os.sched_getaffinity(0))
        nodes = []
        for path in sorted(_glob.glob(_os.path.join(node_root, "node[0-9]*", "cpulist"))):
            cpus = set()
            for part in open(path).read().strip().split(","):
                if part:
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
            if cpus & cur:
                nodes.append(cpus & cur)
        if len(nodes) < 2:
            return None                      # one node, or already confined
        rank, world = int(_os.environ.get("LOCAL_RANK", "0")), max(1, int(_os.environ.get("LOCAL_WORLD_SIZE", "1")))
        pick = nodes[min(len(nodes) - 1, rank * len(nodes) // world)]
        (set_affinity or (lambda cpus: _os.sched_setaffinity(0, cpus)))(pick)
        return sorted(pick)
    except Exception:
        return None


NUMA_CPUS = confine_to_one_numa_node() if (_q and _q < (_os.cpu_count() or 1)) else None
