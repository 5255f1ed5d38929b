#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6xone}; mkdir -p $O
GNNX_LIBRARY_PATH=$PWD/tools/_build/libgnnx_hip_prev.so timeout 600 python tools/r6_xone_check.py $O/prev.npz 2>&1 | tail -3
timeout 600 python tools/r6_xone_check.py $O/new.npz 2>&1 | tail -3
python tools/r6_xone_check.py --compare $O/prev.npz $O/new.npz
rm -f $O/prev.npz $O/new.npz
