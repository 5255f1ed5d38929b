#!/usr/bin/env python
"""Measurement tool (GPU box): per-workgroup timeline of ONE k_conv<FWD2> launch on the streaming set of the BA-House x100k
sample (wall_clock64 stamps injected into a TEMPORARY copy of the sources): start, end of the K loop, end of the epilogue per
workgroup, next to the target's ld.  Answers: is the launch bound by the longest workgroups (chains of dependent HBM round
trips) or by throughput; how long do the fixed parts (weights, reduction, epilogue) take."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
src = open(os.path.join(CSRC, "gnnx_kernels.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
NP = 4 * 16384
src = src.replace("namespace gnnx {\n", "namespace gnnx {\n__device__ unsigned long long g_probe[%d];\n" % NP, 1)
a = src.index("__global__ __launch_bounds__(256) void k_conv(")
body0 = src.index("    conv_stage_weights<MODE>(p, tl, sh, iter);", a)
src = src[:body0] + "    if (MODE == FWD2 && tid == 0 && blockIdx.x < 16384) { g_probe[4 * blockIdx.x] = wall_clock64(); g_probe[4 * blockIdx.x + 3] = ld; }\n" + src[body0:]
k1 = src.index("    // the wave tiles meet in LDS", a)
src = src[:k1] + "    if (MODE == FWD2 && tid == 0 && blockIdx.x < 16384) g_probe[4 * blockIdx.x + 1] = wall_clock64();\n" + src[k1:]
k2 = src.index("// The row blocks whose K range was cut", a)
end = src.rindex("}\n", a, k2)
src = src[:end] + "    if (MODE == FWD2 && tid == 0 && blockIdx.x < 16384) g_probe[4 * blockIdx.x + 2] = wall_clock64();\n" + src[end:]
capi = capi.replace('#include "../../include/gnnx.h"', '#include "../../../include/gnnx.h"')
capi += '\nextern "C" int gnnx_probe_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnnx::g_probe), sizeof(unsigned long long) * n); }\n'
tmp = os.path.join(ROOT, "tools", "_build", "conv")
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
lib = engine.bind(ctypes.CDLL(so))
wl = bench.Workload("ba100k", 1024)
graph = engine.device_graph(wl.idx.csr, wl.feat, wl.pred, lib=lib) if "lib" in engine.device_graph.__code__.co_varnames else engine.device_graph(wl.idx.csr, wl.feat, wl.pred)
dn = engine.khop_device(graph, wl.targets, 3)
job = MaskOptimJob.from_csr(graph, dn, None, wl.label[wl.targets], wl.ck["sd"], lib=lib)
job.set_masks_raw(engine.init_edge_masks_raw(dn.sizes, seeds=1000 + wl.targets, pin=True))
hy = Hyper(num_iters=300, edge_results_only=True)
for k in (0, 1, 2):
    job.time_kernel(hy, k, 3)
ms = job.time_kernel(hy, 2, 1)[0]
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * NP)()
lib.gnnx_probe_read(buf, NP)
a = np.frombuffer(buf, dtype=np.uint64).astype(np.int64).reshape(-1, 4)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
st, kl, en, ld = (a[:, 0] - t0) / 100.0, (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - a[:, 1]) / 100.0, a[:, 3]
print(f"k_conv<FWD2>: {len(a)} workgroups, launch {ms * 1e3:.1f} us by events, last workgroup ends at {(a[:, 2].max() - t0) / 100.0:.1f} us")
print("  ld    workgroups   start us (min / median / max)     K loop us (median / max)   epilogue us (median / max)")
for v in np.unique(ld)[::-1][:40:1]:
    m = ld == v
    if m.sum() < 1:
        continue
    print(f"  {v:5d} {int(m.sum()):6d}        {st[m].min():7.1f} {np.median(st[m]):7.1f} {st[m].max():7.1f}          {np.median(kl[m]):7.1f} {kl[m].max():7.1f}           {np.median(en[m]):6.1f} {en[m].max():6.1f}")
edges = np.linspace(0, (a[:, 2].max() - t0) / 100.0, 21)
busy = [(int(((a[:, 0] - t0) / 100.0 <= e).sum() - ((a[:, 2] - t0) / 100.0 <= e).sum())) for e in edges]
print("workgroups in flight at 20 points of the launch:", busy)
