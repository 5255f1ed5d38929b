#!/usr/bin/env python
"""Measurement tool (GPU box): the dense streaming kernels on the BA-House x100k sample, per kernel and per K-slice policy.

    GNNX_SPARSE_RESIDENT=0 python tools/probe_conv.py [targets=1024] [wide:ku,wide:ku,...]

For every GNNX_CONV_WIDE:GNNX_CONV_KU pair in the list (1:0 = one row block per workgroup over whole rows, the round-2 form) it builds the plan of the same batch,
times k_mask / k_conv<FWD1> / k_conv<FWD2> / k_node_head / k_conv<BWD1> with gnnx_time_kernel (HIP events around `reps`
launches on the engine's stream) and prints the algorithmic rates: k_conv reads Abar once (4 n^2 B), k_mask moves 28 n^2 B,
one iteration 40 n^2 B (node mode)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GNNX_SPARSE_RESIDENT", "0")
import torch
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob

ntargets = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kus = [tuple(int(v) for v in x.split(":")) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1:0", "1:1024"])]   # wide:ku pairs
only_streaming = "--all" not in sys.argv
wl = bench.Workload("ba100k", ntargets)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
hy = Hyper(num_iters=300, edge_results_only=True)
dn = engine.khop_device(graph, wl.targets, 3)
names = ["k_mask", "k_conv<FWD1>", "k_conv<FWD2>", "k_node_head", "k_conv<BWD1>"]
out = {}
for ku in kus:
    os.environ["GNNX_CONV_WIDE"] = str(ku[0])
    os.environ["GNNX_CONV_KU"] = str(ku[1])
    job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"])
    raw = engine.init_edge_masks_raw(dn.sizes, seeds=1000 + wl.targets, pin=True)
    job.set_masks_raw(raw)
    torch.cuda.synchronize()
    route = job.route()
    n2 = float((job.n[route == 0].astype(np.float64) ** 2).sum())
    per = [job.time_kernel(hy, k, 20) for k in range(5)]
    per = [job.time_kernel(hy, k, 20) for k in range(5)]     # second pass: clocks and caches settled
    it_us = sum(x[0] for x in per) * 1e3
    row = {nm: round(x[0] * 1e3, 1) for nm, x in zip(names, per)}
    row["iteration_us"] = round(it_us, 1)
    row["conv_TBps"] = [round(4 * n2 / (per[k][0] * 1e-3) / 1e12, 2) for k in (1, 2, 4)]
    row["mask_TBps"] = round(28 * n2 / (per[0][0] * 1e-3) / 1e12, 2)
    row["iteration_frac_of_8TBps"] = round(40 * n2 / (it_us * 1e-6) / 8e12, 3)
    # a whole run for the wall clock (hipGraph replay of 300 iterations; the resident launches of the small targets beside it)
    job.set_masks_raw_resident(); job.launch(hy); torch.cuda.synchronize()
    t0 = time.perf_counter()
    job.set_masks_raw_resident(); job.launch(hy); torch.cuda.synchronize()
    row["run_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    out[ku] = row
    print(f"WIDE:KU={ku}: streaming targets {int((route == 0).sum())}, sum n^2 {n2:.3e}", json.dumps(row), flush=True)
    job.close()
