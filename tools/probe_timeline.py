#!/usr/bin/env python
"""Measurement tool (GPU box): device-side timeline of one k_conv<FWD2> launch on the syn1 workload.
Builds a TEMPORARY copy of the product sources with wall_clock64() stamps injected at phase boundaries
(anchored on comments), loads it instead of libgnnx_hip.so and prints per-phase times of the slowest workgroup
and the mean over workgroups.  The product sources are not modified."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
hdr = open(os.path.join(CSRC, "gnnx_kernels.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
NP = 8
hdr = hdr.replace("namespace gnnx {\n", "namespace gnnx {\n__device__ unsigned long long g_probe[4096 * %d];\n"
                  "#define PROBE(k) do { if (MODE == FWD2 && threadIdx.x == 0 && blockIdx.x < 4096) g_probe[blockIdx.x * %d + (k)] = wall_clock64(); } while (0)\n" % (NP, NP), 1)
def inject(src, anchor, stamp, after=True):
    assert anchor in src, anchor
    return src.replace(anchor, (anchor + stamp) if after else (stamp + anchor), 1)
i0 = hdr.index("__global__ __launch_bounds__(256) void k_conv(")
body = hdr[i0:]
body = inject(body, "    __shared__ ConvShared sh;\n", "    PROBE(0);\n")
body = inject(body, "    conv_epilogue_operands<MODE>(p, tl, tm, irow, cg, pre_a, pre_b, pre_c, pre_ar, pre_rn);\n", "    PROBE(1);\n")
body = inject(body, "    // split-K reduction through LDS\n", "    PROBE(2);\n", after=False)
body = inject(body, "    float z4[4];\n#pragma unroll\n    for (int j = 0; j < 4; ++j) {\n        float s = 0.0f;", "    PROBE(3);\n", after=False)
body = inject(body, "    conv_epilogue<MODE>(p, tl, tm, sh, z4, irow, tl.rb, pre_a, pre_b, pre_c, pre_ar, pre_rn);\n", "    PROBE(4);\n")
hdr = hdr[:i0] + body
capi = capi.replace('#include "gnnx_kernels.hpp"', '#include "gnnx_kernels_probe.hpp"')
capi = capi.replace('#include "../../include/gnnx.h"', '#include "%s"' % os.path.join(ROOT, "include", "gnnx.h"))
capi += '\nextern "C" int gnnx_probe_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnnx::g_probe), sizeof(unsigned long long) * n); }\n'
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "gnnx_kernels_probe.hpp"), "w").write(hdr)
open(os.path.join(tmp, "capi_probe.hip"), "w").write(capi)
so = os.path.join(tmp, "libprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(tmp, "capi_probe.hip"), "-o", so])
import bench
from gnn_model_explainer_amd import engine
lib = engine.bind(ctypes.CDLL(so))
wl = bench.Workload("syn1"); wl.prepare()
ck, subs = wl.ck, [wl.dense_subgraph(k) for k in range(len(wl.targets))]
job = engine.MaskOptimJob(subs, ck["sd"], lib=lib)
hy = engine.Hyper(num_iters=3)
job.run([s.mask0 for s in subs], hy)
ms, _, _ = job.time_kernel(hy, 2, 1)      # warm + 1 timed launch of FWD2
buf = (ctypes.c_ulonglong * (4096 * NP))()
lib.gnnx_probe_read(buf, 4096 * NP)
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, NP)[:748, :5].astype(np.int64)
t0 = a[:, 0].min()
d = (a - t0) * 10.0 / 1e3      # wall_clock64: 100 MHz -> us
print("k_conv<FWD2> event-timed launch: %.2f us" % (ms * 1e3))
print("per-WG stamps (us since first WG start): start, operands issued, K loop done, reduce start, end")
print("mean   ", np.round(d.mean(0), 2))
print("max    ", np.round(d.max(0), 2))
slow = d[:, 4].argmax(); print("slowest WG", slow, np.round(d[slow], 2), "n_tile", subs and "")
print("WG duration: mean %.2f  p50 %.2f  max %.2f us; last WG start %.2f us" % ((d[:,4]-d[:,0]).mean(), np.median(d[:,4]-d[:,0]), (d[:,4]-d[:,0]).max(), d[:,0].max()))
