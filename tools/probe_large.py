#!/usr/bin/env python
"""Measurement tool (GPU box): device-side phase timeline of one iteration of k_sparse_large for one BA-House x100k target
(default: the largest of the 2048-target sample), via wall_clock64() stamps injected into a TEMPORARY copy of the sources."""
import ctypes, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
src = open(os.path.join(CSRC, "gnnx_sparse_large.hpp")).read()
capi = open(os.path.join(CSRC, "gnnx_capi.hip")).read()
NP = 32
src = src.replace("namespace gnnx {\n", "namespace gnnx {\n__device__ unsigned long long g_probe[%d];\n"
                  "#define PROBE(k) do { if (iter == 5 && threadIdx.x == 0 && blockIdx.x == 0) g_probe[(k)] = wall_clock64(); } while (0)\n" % NP, 1)
src = src.replace("#define PROBE(k)", "#define PROBE0(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_probe[(k)] = wall_clock64(); } while (0)\n#define PROBE(k)", 1)
setup_marks = [("    // ---------------- setup 1: hop levels", 16), ("    // ---------------- setup 2: the rows of A", 17),
               ("    // column ids of the active entries", 18), ("    // ---------------- setup 3: row slots", 19),
               ("    // slot records -> workspace", 20), ("    // ---------------- setup 4: the undirected edges", 21),
               ("    // ---------------- row arrays (columns beyond", 22), ("    for (int iter = 0; iter < p.num_iters; ++iter) {", 23),
               ("    // ---------------- results: dense Abar block", 24), ("    // ---------------- far edges: the whole trajectory", 25)]
for mark, idx in setup_marks:
    assert mark in src, mark
    src = src.replace(mark, "    PROBE0(%d);\n" % idx + mark, 1)
src = src.replace("    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? sh.fcur[tid] : 0.0f;\n}", "    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? sh.fcur[tid] : 0.0f;\n    PROBE0(26);\n}", 1)
anchors = [l for l in src.split("\n") if l.strip().startswith("// ========")]
names = []
for k, a in enumerate(anchors):
    src = src.replace(a + "\n", "        PROBE(%d);\n" % k + a + "\n", 1)
    names.append(a.strip(" /="))
k = len(anchors)
# finer stamps inside layer 1 (wave 0's view): after the slot record, the gather, the combine, the row-local part
fine = [("            const bool first = SA.first;\n            const int r = first ? SA.row : 0;\n            float acc[DQ];", 27),
        ("            sparse_combine<DQ>(acc, SA.rem, SA.wsplit);\n            // Zraw for the feature-mask gradient", 28),
        ("            sparse_forward_rowlocal_global<DQ>(acc, sW1, sh.bias[0]", 29)]
for mark, idx in fine:
    assert mark in src, mark
    src = src.replace(mark, "            PROBE(%d);\n" % idx + mark, 1)
end_anchor = "        if (tid < D) {  // feature mask\n"
assert end_anchor in src
src = src.replace(end_anchor, "        PROBE(%d);\n" % k + end_anchor, 1)
capi = capi.replace('#include "../../include/gnnx.h"', '#include "../../../include/gnnx.h"')
capi += '\nextern "C" int gnnx_probe_read(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnnx::g_probe), sizeof(unsigned long long) * n); }\n'
# `--build`: cross-compile here (no GPU needed) into tools/_build_large/ - the .so travels with the gpurun snapshot
tmp = os.path.join(ROOT, "tools", "_build", "large")
os.makedirs(tmp, exist_ok=True)
so = os.path.join(tmp, "libprobe.so")
srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".hip"))]
if "--build" in sys.argv or not os.path.exists(so) or any(os.path.getmtime(f) > os.path.getmtime(so) for f in srcs):
    for f in os.listdir(CSRC):   # unpatched copies next to the patched source
        if f.endswith(".hpp"):
            open(os.path.join(tmp, f), "w").write(open(os.path.join(CSRC, f)).read())
    open(os.path.join(tmp, "gnnx_sparse_large.hpp"), "w").write(src)
    open(os.path.join(tmp, "capi_probe.hip"), "w").write(capi)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "capi_probe.hip", "-o", "libprobe.so"], cwd=tmp)
if "--build" in sys.argv:
    print("built", so)
    sys.exit(0)
import bench, helpers
from gnn_model_explainer_amd import engine
lib = engine.bind(ctypes.CDLL(so))
NTARGETS = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
wl = bench.Workload("ba100k", NTARGETS)
nbs = wl.idx.neighbors_batch(wl.targets)
route_all = None
order = np.argsort([-len(x) for x in nbs])
kk = int(order[int(sys.argv[1]) if len(sys.argv) > 1 else 0])
tt, nb = int(wl.targets[kk]), nbs[kk]
subs = [wl.dense_subgraph(tt, nb, int(np.searchsorted(nb, tt)), helpers.seeded_mask0(tt, len(nb)).numpy())]
print("target n =", subs[0].adj.shape[0], "undirected edges =", int((subs[0].adj != 0).sum() // 2))
job = engine.MaskOptimJob(subs, wl.ck["sd"], lib=lib)
print("route", job.route())
job.run([s.mask0 for s in subs], engine.Hyper(num_iters=20))
buf = (ctypes.c_ulonglong * NP)()
lib.gnnx_probe_read(buf, NP)
a = np.frombuffer(buf, dtype=np.uint64)[:len(names) + 1].astype(np.int64)
d = np.diff(a) * 10.0 / 1e3
for nme, v in zip(names, d):
    print("%-100s %7.2f us" % (nme[:100], v))
print("iteration (without the feature-mask tail) %7.2f us" % ((a[len(names)] - a[0]) * 10.0 / 1e3))
b = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
print("layer 1, wave 0: slot record %.2f us, gather %.2f, combine + Zraw %.2f, row-local + U1 store + barrier %.2f" % (
    (b[27] - b[0]) / 100.0, (b[28] - b[27]) / 100.0, (b[29] - b[28]) / 100.0, (b[1] - b[29]) / 100.0))
lab = ["hop levels", "rows of A, compact entry ranges", "active column ids", "slot tables (one thread per set)",
       "slot records", "edges (near first) + state planes", "row arrays / model / first Abar", "20 iterations", "dense Abar + near scatter",
       "far edges (20 iterations in registers)"]
for i, nme in enumerate(lab):
    print("%-50s %9.1f us" % (nme, (b[17 + i] - b[16 + i]) / 100.0))
