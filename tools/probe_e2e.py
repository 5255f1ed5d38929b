#!/usr/bin/env python
"""Measurement tool (GPU box): where one batch's host-side set-up time goes (cold and warm)."""
import os, sys, time, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import MaskOptimJob, _check
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "syn1", 2048)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
lib = engine.get_library()
T = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
for rep in range(3):
    t0 = T(); dn = engine.khop_device(graph, wl.targets, 3); t1 = T()
    self = MaskOptimJob.__new__(MaskOptimJob)
    self.lib, self.device, self.graph_mode = lib, graph.feat.device, False
    self._init_model(wl.ck["sd"]); t2 = T()
    self.T = len(dn); self.n = np.ascontiguousarray(dn.sizes, np.int32)
    self._create_plan(np.asarray(dn.rows, np.int32), np.asarray(wl.label[wl.targets], np.int32)); t3 = T()
    self._alloc_device(); t4 = T()
    self._enter()
    _check(lib, lib.gnnx_pack_csr(self.handle, graph.indptr.data_ptr(), graph.indices.data_ptr(), None, graph.feat.data_ptr(), graph.feat.shape[1],
                                  graph.pred_label.data_ptr(), dn.nb_flat.data_ptr(), dn.nb_off.data_ptr(), self.A.data_ptr(), self.X.data_ptr(),
                                  self.yhat.data_ptr(), self._stream()))
    self._leave(); t5 = T()
    self.analyze(); t6 = T()
    raw = engine.init_edge_masks_raw(dn.sizes, seeds=1000 + wl.targets, pin=True); t7 = T()
    self.set_masks_raw(raw); t8 = T()
    self.launch(engine.Hyper(num_iters=300)); t9 = T()
    em = self.fetch_edges(); t10 = T()
    names = ["khop", "model_arrays", "plan_create", "alloc_device", "pack_csr", "analyze", "host_rng", "h2d+scatter", "run", "fetch_edges"]
    ts = [t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10]
    print("rep", rep, " ".join("%s=%.2f" % (n, (b - a) * 1e3) for n, a, b in zip(names, ts[:-1], ts[1:])), "total=%.2f ms" % ((t10 - t0) * 1e3), flush=True)
    self.close()
