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
