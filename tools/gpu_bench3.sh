#!/bin/bash
# the three headline bench lines (syn1 default with CPU baselines, BA-House x100k 2048 / 16384 targets)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_syn1_default.json
timeout 600 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ba100k_2048targets.json
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ba100k_16384targets.json
for f in $O/r02_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), round(d['pcie_inclusive']['value']), d['pcie_inclusive']['warm_batch'], d.get('parity',{}).get('rule','')[:60])"; done
