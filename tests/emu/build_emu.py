"""TEST INFRASTRUCTURE ONLY: compile the product's HIP sources against the CPU emulator header.

Builds tests/emu/_build/libgnnx_emu.so from the SAME gnnx_capi.hip / gnnx_kernels.hpp that hipcc
turns into libgnnx_hip.so, so CPU-only CI exercises the real host orchestration and kernel code.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "gnn-model-explainer_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libgnnx_emu.so")


def build(force=False):
    srcs = [os.path.join(CSRC, "gnnx_capi.hip"), os.path.join(CSRC, "gnnx_kernels.hpp"),
            os.path.join(HERE, "emu_runtime.cpp"), os.path.join(HERE, "include", "hip", "hip_runtime.h"),
            os.path.join(ROOT, "include", "gnnx.h"), os.path.join(CSRC, "gnnx_resident.hpp"),
            os.path.join(CSRC, "gnnx_sparse.hpp"), os.path.join(CSRC, "gnnx_sparse_large.hpp"),
            os.path.join(CSRC, "gnnx_graph.hpp"), os.path.join(CSRC, "gnnx_att.hpp"), os.path.join(CSRC, "gnnx_xl.hpp")]
    fresh = lambda: os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    import fcntl
    with open(OUT + ".lock", "w") as lock:      # pytest-xdist workers: one of them builds, the others wait and find the library fresh
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        cxx = "/opt/rocm/lib/llvm/bin/clang++"
        if not os.path.exists(cxx):
            cxx = "clang++"
        tmp = OUT + ".tmp%d" % os.getpid()
        cmd = [cxx, "-O2", "-g", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=on", "-Wno-psabi", "-I", os.path.join(HERE, "include"),
               "-x", "c++", srcs[0], srcs[2], "-o", tmp]
        subprocess.check_call(cmd)
        os.replace(tmp, OUT)      # never a half-written library under the final name
    return OUT


if __name__ == "__main__":
    print(build(force=True))
