#!/bin/bash
# full GPU session: every -m gpu test, bench with CPU baseline, smoke
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
export GNNX_DUMP_OUTLIERS=$PWD/$O/outliers.json
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
grep -E "passed|failed|FAILED|rc=" $O/pytest_gpu.log | tail -12; tail -2 $O/bench_syn1.err; tail -2 $O/smoke.log
