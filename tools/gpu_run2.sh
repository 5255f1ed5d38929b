#!/bin/bash
O=gpurun_out/r2b; mkdir -p $O
export TMPDIR=/tmp
export GNNX_DUMP_OUTLIERS=$PWD/$O/outliers.json
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-parity-gate > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
rm -rf $O/prof
tail -15 $O/pytest_gpu.log; tail -3 $O/bench_syn1.err
