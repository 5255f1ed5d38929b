#!/bin/bash
# rocprofv3 kernel statistics of the BA-House x100k 2048-target bench (plan + optimisation kernels)
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --workload ba100k --targets ${2:-2048} --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.json 2> $GRAFT_REPO_ROOT/$O/bench.err
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; head -25 $O/kernel_stats.csv | cut -c1-220
