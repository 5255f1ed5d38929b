#!/bin/bash
# phase timelines of k_sparse_large: largest target of the 2048-target sample, largest of the 16384-target set, a mid-size one
O=gpurun_out/$1; mkdir -p $O
timeout 300 python tools/probe_large.py 0 > $O/probe_large_0.log 2>&1; tail -24 $O/probe_large_0.log
timeout 600 python tools/probe_large.py 0 16384 > $O/probe_large_xl.log 2>&1; tail -24 $O/probe_large_xl.log
timeout 300 python tools/probe_large.py 60 > $O/probe_large_60.log 2>&1; tail -24 $O/probe_large_60.log
