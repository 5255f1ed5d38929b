#!/bin/bash
# A/B (round 3): IEEE math (-DGNNX_IEEE_MATH) vs hardware approximations (default) - windowed parity counts and the syn1 loop time
O=gpurun_out/$1; mkdir -p $O
for v in precise approx; do
  if [ $v = approx ]; then export GNNX_LIBRARY_PATH=$PWD/tools/_build/libgnnx_approx.so; else unset GNNX_LIBRARY_PATH; fi
  GNNX_DUMP_WINDOWS=$O/rows_$v python -m pytest tests/test_windowed_parity.py -m gpu -q -s -k "windows" > $O/windowed_$v.log 2>&1
  echo "== $v"; grep -o "^[a-z0-9]*: [0-9]* windows the two CPU implementations agree on, [0-9]* within 1e-5 (worst [0-9.e-]*); [0-9]* sub-windows they disagree on: [0-9]* within 1e-5, worst [0-9.e-]*" $O/windowed_$v.log
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-gate > $O/bench_$v.json 2> $O/bench_$v.err
  python -c "import json,sys; d=json.load(open('$O/bench_$v.json')); print('syn1 ms_per_step', d['ms_per_step'], 'parity', d.get('parity',{}).get('rule'))"
done
