#!/bin/bash
# rocprofv3 kernel statistics of the 16 384-target BA-House x100k bench line
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_kernel_stats_ba100k_16384targets.csv
rm -rf $O/prof
head -8 $O/r02_kernel_stats_ba100k_16384targets.csv | cut -c1-200
