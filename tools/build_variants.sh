#!/bin/bash
# A/B builds of the kernel library (same sources, one -D setting each) under tools/_build/ab/ for the measurement sessions:
#   tools/build_variants.sh name1 "-DGNNX_X=0" name2 "-DGNNX_Y=0 -DGNNX_Z=1" ...      (builds run in parallel)
# (round 4 measured the resident kernel's forms this way - sessions n ... r2; the macros those sessions switched are gone from the sources:
#  the winners are plain code, three stay constexpr switches at the top of gnnx_sparse.hpp; GNNX_IEEE_MATH still is a -D)
# A session selects one with GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_<name>.so (engine.library_path).
cd "$(dirname "$0")/.."
mkdir -p tools/_build/ab
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC $f gnn-model-explainer_amd/csrc/gnnx_capi.hip -o tools/_build/ab/libgnnx_hip_$n.so > tools/_build/ab/build_$n.log 2>&1; echo "built $n ($f): rc $?" ) &
done
wait
