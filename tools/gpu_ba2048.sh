#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_emu_kernels.py tests/test_gpu_full_configs.py -m gpu -q -x --timeout=300 -k "large or config5 or degenerate or routing" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --workload ba100k --targets 2048 --no-cpu-baseline > $O/bench_ba100k_2048.json 2> $O/bench_ba100k_2048.err; echo "rc=$?" >> $O/bench_ba100k_2048.err
tail -3 $O/pytest_gpu.log
python -c "
import json;d=json.loads(open('$O/bench_ba100k_2048.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step']); print(json.dumps(d['roofline']['launches']))"
timeout 300 python tools/probe_large.py > $O/probe_large.log 2>&1; tail -12 $O/probe_large.log
