#!/usr/bin/env python
"""Measurement tool (GPU box): where the ~2.4 ms of host time of one prepared syn1 batch go (pipeline.BatchPipeline._prepare), step by
step, with the GPU idle and with an optimisation running on another stream."""
import os, sys, time, ctypes
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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
lib = engine.get_library()
P = torch.cuda.Stream(dev, priority=-1)
L = torch.cuda.Stream(dev)
with torch.cuda.stream(L):
    dn0 = engine.khop_device(graph, targets, 3)
    busy = MaskOptimJob.from_csr(graph, dn0, None, ck["label"][targets], ck["sd"])
    busy.set_masks_raw(engine.init_edge_masks_raw(dn0.sizes, seeds=1000 + targets))
    busy.launch(Hyper(num_iters=300))
torch.cuda.synchronize()
lib.gnnx_set_service_stream(P.cuda_stream)

def one(keep_busy):
    t = {}
    def lap(name, t0):
        t[name] = t.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    if keep_busy:
        with torch.cuda.stream(L):
            for _ in range(3):
                busy.set_masks_raw_resident(); busy.launch(Hyper(num_iters=300))
    with torch.cuda.stream(P):
        t0 = time.perf_counter(); dn = engine.khop_device(graph, targets, 3); lap("khop_device (2 passes)", t0)
        job = MaskOptimJob.__new__(MaskOptimJob)
        job.lib, job.device, job.graph_mode, job.mask_relu, job.bn = lib, dev, False, False, False
        t0 = time.perf_counter(); job._init_model(ck["sd"]); lap("model arrays", t0)
        job.T = len(targets); job.n = np.ascontiguousarray(dn.sizes, np.int32)
        t0 = time.perf_counter(); job._create_plan(np.asarray(dn.rows, np.int32), np.asarray(ck["label"][targets], np.int32)); lap("gnnx_plan_create + layout", t0)
        t0 = time.perf_counter(); job._alloc_device(); lap("torch.empty x 7", t0)
        t0 = time.perf_counter()
        job._enter()
        engine._check(lib, lib.gnnx_pack_csr(job.handle, graph.indptr.data_ptr(), graph.indices.data_ptr(), None, graph.feat.data_ptr(), graph.feat.shape[1],
                                             graph.pred_label.data_ptr(), dn.nb_flat.data_ptr(), dn.nb_off.data_ptr(), job.A.data_ptr(), job.X.data_ptr(), job.yhat.data_ptr(), job._stream()))
        job._leave(); lap("gnnx_pack_csr (enqueue)", t0)
        t0 = time.perf_counter(); job.analyze(); lap("gnnx_plan_analyze (sync)", t0)
        t0 = time.perf_counter(); job._edge_layout(); lap("edge layout (counts sync + positions)", t0)
        t0 = time.perf_counter(); rc = job._rc.cpu(); lap("rc D2H", t0)
    torch.cuda.synchronize()
    job.close()
    return t

for keep_busy in (False, True):
    acc = {}
    for rep in range(12):
        t = one(keep_busy)
        if rep >= 2:
            for k, v in t.items():
                acc.setdefault(k, []).append(v)
    print("--- GPU", "busy (3 optimisations queued on another stream)" if keep_busy else "idle")
    tot = 0
    for k, v in acc.items():
        print(f"  {k:42s} {np.median(v):7.3f} ms"); tot += np.median(v)
    print(f"  {'total':42s} {tot:7.3f} ms")
