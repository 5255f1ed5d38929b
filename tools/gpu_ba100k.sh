#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python bench.py --steps 2 --warmup 1 --workload ba100k --targets 16384 --no-cpu-baseline > $O/bench_ba100k_16384.json 2> $O/bench_ba100k_16384.err; echo "rc=$?" >> $O/bench_ba100k_16384.err
tail -4 $O/bench_ba100k_16384.err
python -c "
import json;d=json.loads(open('$O/bench_ba100k_16384.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step']); print(d['config']['routing_rank0']); print(json.dumps(d['roofline']['launches'])[:1500]); print(d['pcie_inclusive']['warm_batch'])"
