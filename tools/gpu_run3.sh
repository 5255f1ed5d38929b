#!/bin/bash
O=gpurun_out/r2c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python tools/probe_e2e.py syn1 > $O/probe_e2e_syn1.log 2>&1
timeout 300 python tools/probe_e2e.py ba100k > $O/probe_e2e_ba100k.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
ls -R $O/prof | head -20 > $O/prof_ls.txt
rm -rf $O/prof
cat $O/probe_e2e_syn1.log $O/probe_e2e_ba100k.log | grep rep
