#!/bin/bash
# round 4, session h: pipeline depth / workers with nine repetitions each (the run-to-run spread is 7-18 %)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for cfg in "3 3" "4 3" "4 2" "5 3" "3 3" "4 3"; do set -- $cfg; GNNX_PIPE_DEPTH=$1 GNNX_PIPE_WORKERS=$2 timeout 600 python bench.py --steps 20 --warmup 5 --reps 9 --no-parity-gate --no-cpu-baseline > $O/bench_d$1_w$2.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_d$1_w$2.json').read().strip().splitlines()[-1]);r=d['end_to_end_stage_ms']['repetitions'];print('depth $1 workers $2 median',round(d['value']),'values',[round(v/1e3,1) for v in r['values']])"; done
