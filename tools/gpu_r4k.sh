#!/bin/bash
# round 4, session k: host threads of the seeded draw in the syn1 pipeline (the container has a 16-core CPU quota)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for th in 32 16 8 32 16; do GNNX_RNG_THREADS=$th timeout 600 python bench.py --reps 9 --no-parity-gate --no-cpu-baseline > $O/bench_rng$th.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_rng$th.json').read().strip().splitlines()[-1]);e=d['end_to_end_stage_ms'];print('rng threads $th median',round(d['value']),'rng ms',round(e['host_rng_ms'],2),'prepare',round(e['prepare_ms'],2),[round(v/1e3,1) for v in e['repetitions']['values']])"; done
