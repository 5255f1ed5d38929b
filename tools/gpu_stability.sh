#!/bin/bash
# every workload's bench line twice (fresh processes): does the overlap of the resident launches hold from run to run?
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -2
for rep in 1 2; do
for W in syn1 syn4 syn5; do
timeout 300 python bench.py --workload $W --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/b_${W}_$rep.json
python -c "
import json;d=json.load(open('$O/b_${W}_$rep.json'));print('$W rep $rep', round(d['value']), round(d['ms_per_step'],3), {k[:28]:round(v['ms_total'],2) for k,v in d['roofline']['launches'].items()}, d.get('parity',{}).get('rule','')[:40])"
done
for T in 2048 16384; do
timeout 900 python bench.py --workload ba100k --targets $T --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/b_ba_${T}_$rep.json
python -c "
import json;d=json.load(open('$O/b_ba_${T}_$rep.json'));print('ba100k $T rep $rep', round(d['value']), round(d['ms_per_step'],3), {k[:28]:round(v['ms_total'],2) for k,v in d['roofline']['launches'].items()})"
done
done
timeout 600 python tools/config4_mutag_like.py 2>/dev/null | tail -1 | cut -c1-200
