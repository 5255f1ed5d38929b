#!/bin/bash
# GPU-box session: parity suite, bench, phase timeline, kernel trace.  Outputs under gpurun_out/r2a/.
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O/prof -name "*.csv" ! -name "*kernel_stats.csv" -size +1M -delete
tail -5 $O/pytest_gpu.log; tail -3 $O/bench_syn1.err; cat $O/probe_sparse_0.log | tail -15
