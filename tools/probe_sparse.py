#!/usr/bin/env python
"""Measurement tool (GPU box): device-side phase timeline of one iteration of k_sparse_resident for the largest syn1
target (block 0 of the launch), via wall_clock64() stamps injected into a TEMPORARY copy of the sources."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
src = open(os.path.join(CSRC, "gnnx_sparse.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
NP = 32
src = src.replace("namespace gnnx {\n", "namespace gnnx {\n__device__ unsigned long long g_probe[%d];\n"
                  "#define PROBE(k) do { if (iter == 5 && threadIdx.x == 0 && blockIdx.x == 0) g_probe[(k)] = wall_clock64(); } while (0)\n" % NP, 1)
anchors = [l for l in src.split("\n") if l.strip().startswith("// ========") and "graph mode" not in l]
names = []
for k, a in enumerate(anchors):
    src = src.replace(a + "\n", "        PROBE(%d);\n" % k + a + "\n", 1)
    names.append(a.strip(" /="))
k = len(anchors)
# finer stamps inside layer 2 (wave 0 = the hub rows): after the gather, after the split-row combine, after MFMA + epilogue
SUB = 20
l2 = "            sparse_gather<true, HQ>(sAb, scol, sU1, sH, H, re0, re1, h, acc);\n            sparse_combine<HQ>(acc, lane, first, nsplit, wsplit);\n"
assert l2 in src
src = src.replace(l2, l2.replace(";\n            sparse_combine", ";\n            PROBE(%d);\n            sparse_combine" % SUB) + "            PROBE(%d);\n" % (SUB + 1), 1)
l2b = "            sparse_forward_rowlocal<HQ>(acc, sW2, sh.bias[1], H, H, li, h, first, sU2 + r * sH, sRn2 + r);\n"
assert l2b in src
src = src.replace(l2b, l2b + "            PROBE(%d);\n" % (SUB + 2), 1)
src = src.replace("        if (iter + 1 < p.num_iters) publish_abar();  // the returned mask", "        PROBE(%d);\n        if (iter + 1 < p.num_iters) publish_abar();\n        PROBE(%d);  // the returned mask" % (k, k + 1), 1)
names += ["publish Abar"]
for f in ("gnnx_kernels.hpp", "gnnx_resident.hpp"):
    capi = capi.replace('#include "%s"' % f, '#include "%s"' % os.path.join(CSRC, f))
    src = src.replace('#include "%s"' % f, '#include "%s"' % os.path.join(CSRC, f))
capi = capi.replace('#include "gnnx_sparse.hpp"', '#include "gnnx_sparse_probe.hpp"')
capi = capi.replace('#include "../../include/gnnx.h"', '#include "%s"' % os.path.join(ROOT, "include", "gnnx.h"))
capi += '\nextern "C" int gnnx_probe_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnnx::g_probe), sizeof(unsigned long long) * n); }\n'
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "gnnx_sparse_probe.hpp"), "w").write(src)
open(os.path.join(tmp, "capi_probe.hip"), "w").write(capi)
so = os.path.join(tmp, "libprobe.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(tmp, "capi_probe.hip"), "-o", so])
import bench
from gnn_model_explainer_amd import engine
lib = engine.bind(ctypes.CDLL(so))
wl = bench.Workload("syn1"); wl.prepare()
order = np.argsort([-len(x) for x in wl.nbs])
sel = [int(order[int(sys.argv[1]) if len(sys.argv) > 1 else 0])]
subs = [wl.dense_subgraph(k) for k in sel]
print("target n =", subs[0].adj.shape[0], "undirected edges =", int((subs[0].adj != 0).sum() // 2))
job = engine.MaskOptimJob(subs, wl.ck["sd"], lib=lib)
job.run([s.mask0 for s in subs], engine.Hyper(num_iters=20))
buf = (ctypes.c_ulonglong * NP)()
lib.gnnx_probe_read(buf, NP)
a = np.frombuffer(buf, dtype=np.uint64)[:len(names) + 1].astype(np.int64)
d = np.diff(a) * 10.0 / 1e3
for nme, v in zip(names, d):
    print("%-100s %7.2f us" % (nme[:100], v))
print("iteration total %7.2f us" % ((a[len(names)] - a[0]) * 10.0 / 1e3))
b = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
print("layer 2, wave 0: gather %.2f us, combine %.2f us, MFMA + epilogue %.2f us, wait at barrier %.2f us" % (
    (b[20] - b[1]) / 100.0, (b[21] - b[20]) / 100.0, (b[22] - b[21]) / 100.0, (b[2] - b[22]) / 100.0))
