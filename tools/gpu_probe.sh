#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1
tail -12 $O/probe_sparse_0.log; tail -4 $O/probe_sparse_150.log
