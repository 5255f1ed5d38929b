#!/usr/bin/env python
"""Measurement tool (GPU box): which host-side operations of the prepare stage wait for a long kernel running on ANOTHER stream?
A syn1 optimisation (3.9 ms, one launch on its lane stream) runs on stream L; meanwhile, on stream P, time: a pageable H2D copy, a
pinned H2D copy, a tiny kernel + event sync, .cpu() of a small tensor, a pinned D2H + event sync, gnnx_khop (sizes pass)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob
from gnn_model_explainer_amd.utils.graph_utils import KHopIndex
dev = torch.device("cuda", 0)
ck = helpers.load_ckpt("syn1")
idx = KHopIndex((ck["num_nodes"], ck["edges"]), 3)
graph = engine.device_graph(idx.csr, ck["feat"], ck["pred"])
targets = np.arange(300, 700)
L, P = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
with torch.cuda.stream(L):
    dn = engine.khop_device(graph, targets, 3)
    job = MaskOptimJob.from_csr(graph, dn, None, ck["label"][targets], ck["sd"])
    job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + targets))
    job.launch(Hyper(num_iters=300))
torch.cuda.synchronize()
small = np.arange(400, dtype=np.int32)
pin = torch.from_numpy(small).pin_memory()
dsmall = torch.zeros(400, dtype=torch.int32, device=dev)
pin_out = torch.empty(400, dtype=torch.int32).pin_memory()
lib = engine.get_library()

def ops():
    return {
        "pageable H2D (.to)": lambda: torch.from_numpy(small).to(dev),
        "pinned H2D non_blocking + event": lambda: (dsmall.copy_(pin, non_blocking=True), torch.cuda.current_stream().synchronize()),
        "tiny kernel + stream sync": lambda: (dsmall.add_(1), torch.cuda.current_stream().synchronize()),
        ".cpu() of 400 ints": lambda: dsmall.cpu(),
        "pinned D2H non_blocking + stream sync": lambda: (pin_out.copy_(dsmall, non_blocking=True), torch.cuda.current_stream().synchronize()),
        "torch.empty on device": lambda: torch.empty(1 << 20, dtype=torch.float32, device=dev),
        "khop_device (2 passes, 2 syncs)": lambda: engine.khop_device(graph, targets, 3),
    }

for busy in (False, True):
    print("--- long kernel running on stream L:" , busy)
    for name, fn in ops().items():
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            if busy:
                with torch.cuda.stream(L):
                    job.set_masks_raw_resident()
                    job.launch(Hyper(num_iters=300))
                time.sleep(0.0005)
            with torch.cuda.stream(P):
                t0 = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - t0) * 1e3)
        print(f"  {name:42s} {min(ts):7.3f} ms (min of 4; others {[round(x, 3) for x in ts]})")
