#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for K in 0 1; do
GNNX_KEEP_256=$K timeout 600 python bench.py --steps 5 --warmup 2 --workload ba100k --targets 2048 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_keep$K.json
python -c "
import json;d=json.load(open('$O/bench_keep$K.json'));print('keep256=$K value',round(d['value']),'ms',round(d['ms_per_step'],3)); print(json.dumps(d['roofline']['launches'])[:600])"
done
timeout 600 python -m pytest tests/test_explainer_api.py -m gpu -q -x -k surface 2>&1 | tail -2
