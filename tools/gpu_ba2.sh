#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2; do
for T in 2048 16384; do
timeout 900 python bench.py --workload ba100k --targets $T --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_ba_${T}_$rep.json
python -c "
import json;d=json.load(open('$O/b_ba_${T}_$rep.json'));print('ba100k $T rep $rep', round(d['value']), round(d['ms_per_step'],3), {k[:28]:round(v['ms_total'],2) for k,v in d['roofline']['launches'].items()})"
done
done
