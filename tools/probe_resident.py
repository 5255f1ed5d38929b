#!/usr/bin/env python
"""Measurement tool (GPU box): device-side phase timeline of one iteration of k_resident<NB> (block 0; syn4 -> NB = 1, syn5 has 2-block targets first), via
wall_clock64() stamps injected into a TEMPORARY copy of the sources (anchored on comments)."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
res = open(os.path.join(CSRC, "gnnx_resident.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
NP = 16
res = res.replace("namespace gnnx {\n", "namespace gnnx {\n__device__ unsigned long long g_probe[%d];\n"
                  "#define PROBE(k) do { if (iter == 5 && threadIdx.x == 0 && blockIdx.x == 0) g_probe[(k)] = wall_clock64(); } while (0)\n" % NP, 1)
anchors = [
    ("        // ---- layer 1: Zraw = Abar . X ; U1 ----\n", 0),
    ("        // ---- layer 2: U2 ----\n", 1),
    ("        // ---- row t of Abar . relu(U2) (the only row of layer 3 the reference reads, explain.py:713) ----\n", 2),
    ("        // ---- layer 3 (row t only), head, dE, dZ3[t] ----\n", 3),
    ("        // ---- dZ2 (rank-1: dX2[r] = Abar[r][t] dZ3[t] + dE2 on row t) and g3 ----\n", 4),
    ("        // ---- dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 ; feature-mask gradient ----\n", 5),
    ("        // ---- per tile pair: G = dL/dAbar (+ transpose) on MFMA (K = D + H split over the waves; layer 3 is the\n", 6),
    ("        if (tid < p.D) {  // feature mask\n", 7),
]
for a, k in anchors:
    assert a in res, a
    res = res.replace(a, "        PROBE(%d);\n" % k + a, 1)
res = res.replace("        if (iter + 1 < p.num_iters) publish_abar();  // the returned mask", "        PROBE(8);\n        if (iter + 1 < p.num_iters) publish_abar();\n        PROBE(9);  // the returned mask", 1)
capi = capi.replace('#include "gnnx_resident.hpp"', '#include "gnnx_resident_probe.hpp"')
capi = capi.replace('#include "gnnx_kernels.hpp"', '#include "%s"' % os.path.join(CSRC, "gnnx_kernels.hpp"))
capi = capi.replace('#include "../../include/gnnx.h"', '#include "%s"' % os.path.join(ROOT, "include", "gnnx.h"))
res = res.replace('#include "gnnx_kernels.hpp"', '#include "%s"' % os.path.join(CSRC, "gnnx_kernels.hpp"))
capi += '\nextern "C" int gnnx_probe_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnnx::g_probe), sizeof(unsigned long long) * n); }\n'
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "gnnx_resident_probe.hpp"), "w").write(res)
open(os.path.join(tmp, "capi_probe.hip"), "w").write(capi)
so = os.path.join(tmp, "libprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(tmp, "capi_probe.hip"), "-o", so])
import bench
from gnn_model_explainer_amd import engine
lib = engine.bind(ctypes.CDLL(so))
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "syn4"); wl.prepare()
ck, subs = wl.ck, [wl.dense_subgraph(k) for k in range(len(wl.targets))]
job = engine.MaskOptimJob(subs, ck["sd"], lib=lib)
job.run([s.mask0 for s in subs], engine.Hyper(num_iters=20))
buf = (ctypes.c_ulonglong * NP)()
lib.gnnx_probe_read(buf, NP)
a = np.frombuffer(buf, dtype=np.uint64)[:10].astype(np.int64)
d = np.diff(a) * 10.0 / 1e3
names = ["layer1 (contract+epilogue)", "layer2", "row t of layer 3 (colsum)", "layer3 row t + head + dZ3", "dZ2 + g3",
         "BWD1 + df colsum", "G tiles MFMA + gradient + Adam", "feature mask", "publish Abar"]
for nme, v in zip(names, d):
    print("%-32s %6.2f us" % (nme, v))
print("iteration total                  %6.2f us" % ((a[9] - a[0]) * 10.0 / 1e3))
