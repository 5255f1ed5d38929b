#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/b_$rep.json
python -c "
import json;d=json.load(open('$O/b_$rep.json'));print(round(d['value']), round(d['pcie_inclusive']['value']), {k:round(v,2) for k,v in d['pcie_inclusive']['warm_batch'].items()})"
done
