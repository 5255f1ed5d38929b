#!/bin/bash
# round 5, session an: rocprofv3 kernel stats of config 5 (16 384-target set, loop only) and of the k_att batch, for profiles/
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5an}; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ba -- python $GRAFT_REPO_ROOT/bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --no-cpu-baseline --loop-only > $GRAFT_REPO_ROOT/$O/r05_bench_ba100k_16384targets_loop_only_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_att -- python $GRAFT_REPO_ROOT/tools/probe_att.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_ba -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_kernel_stats_ba100k_16384targets_loop_only.csv; rm -rf $O/prof_ba
find $O/prof_att -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_kernel_stats_method_att_syn1.csv; rm -rf $O/prof_att
head -6 $O/r05_kernel_stats_ba100k_16384targets_loop_only.csv | cut -c1-220; head -4 $O/r05_kernel_stats_method_att_syn1.csv | cut -c1-220
