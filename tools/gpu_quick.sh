#!/bin/bash
# quick kernel-iteration session: emulator-twin cases on the GPU, bench without CPU baseline, phase timeline
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_emu_kernels.py -m gpu -q -x --timeout=300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1.json 2> $O/bench_syn1.err; echo "bench rc=$?" >> $O/bench_syn1.err
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1
tail -3 $O/pytest_gpu.log; tail -2 $O/bench_syn1.err; python -c "
import json;d=json.loads(open('$O/bench_syn1.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step'],'e2e',d['pcie_inclusive']['value']); p=d.get('parity'); print(p['rule'])"
tail -11 $O/probe_sparse_0.log; tail -3 $O/probe_sparse_150.log
