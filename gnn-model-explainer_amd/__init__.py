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
