#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
for k in 45 70 100; do timeout 300 python tools/probe_sparse.py $k > $O/probe_sparse_$k.log 2>&1; tail -13 $O/probe_sparse_$k.log | grep -v amdgpu; done
