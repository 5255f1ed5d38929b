#!/bin/bash
# k_sparse_large session: its emulator twins + the BA-House x100k goldens on hardware, then the 2048- and 16384-target benches
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_emu_kernels.py tests/test_gpu_full_configs.py -m gpu -q -x --timeout=600 -k "large or config5 or degenerate or routing" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for T in 2048 16384; do
timeout 1200 python bench.py --steps 3 --warmup 1 --workload ba100k --targets $T --no-cpu-baseline > $O/bench_ba100k_$T.json 2> $O/bench_ba100k_$T.err; echo "rc=$?" >> $O/bench_ba100k_$T.err
tail -2 $O/bench_ba100k_$T.err
python -c "
import json;d=json.loads(open('$O/bench_ba100k_$T.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step']); print(d['config']['routing_rank0']); print(json.dumps(d['roofline']['launches'])[:900]); print(d['pcie_inclusive']['warm_batch'])"
done
