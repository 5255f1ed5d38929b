#!/usr/bin/env python
"""Measurement tool (GPU box): upper bound of what taking the B operand of k_conv off the L1-miss path could buy.  A TEMPORARY
copy of the sources reads B rows (k & 7) instead of k - they hit the CU's L1 - so the results are WRONG and only the launch times
mean something: k_conv with A (Abar) as the only stream of L1 misses."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
src = open(os.path.join(CSRC, "gnnx_kernels.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
assert "b[u] = on ? Bsrc[(size_t)k * FS] : 0.0f;" in src
src = src.replace("b[u] = on ? Bsrc[(size_t)k * FS] : 0.0f;", "b[u] = on ? Bsrc[(size_t)(k & 7) * FS] : 0.0f;")
capi = capi.replace('#include "../../include/gnnx.h"', '#include "../../../include/gnnx.h"')
tmp = os.path.join(ROOT, "tools", "_build", "conv_nob")
os.makedirs(tmp, exist_ok=True)
so = os.path.join(tmp, "libprobe.so")
if "--build" in sys.argv:
    for f in os.listdir(CSRC):
        if f.endswith(".hpp"):
            open(os.path.join(tmp, f), "w").write(open(os.path.join(CSRC, f)).read())
    open(os.path.join(tmp, "gnnx_kernels.hpp"), "w").write(src)
    open(os.path.join(tmp, "capi_probe.hip"), "w").write(capi)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "capi_probe.hip", "-o", "libprobe.so"], cwd=tmp)
    print("built", so)
    sys.exit(0)
os.environ.setdefault("GNNX_SPARSE_RESIDENT", "0")
import torch
import bench
from gnn_model_explainer_amd import engine
from gnn_model_explainer_amd.engine import Hyper, MaskOptimJob
wl = bench.Workload("ba100k", 1024)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
dn = engine.khop_device(graph, wl.targets, 3)
hy = Hyper(num_iters=300, edge_results_only=True)
for name, lib in (("product", None), ("B rows from L1 (wrong results)", engine.bind(ctypes.CDLL(so)))):
    for ku in ("0", "2048", "1024"):
        os.environ["GNNX_CONV_KU"] = ku
        job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"], lib=lib)
        job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + wl.targets, pin=True))
        torch.cuda.synchronize()
        for k in (1, 2, 4):
            job.time_kernel(hy, k, 10)
        print(name, "KU", ku, {k: round(job.time_kernel(hy, k, 20)[0] * 1e3, 1) for k in (1, 2, 4)}, flush=True)
        job.close()
