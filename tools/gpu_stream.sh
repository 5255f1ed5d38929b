#!/bin/bash
# the dense streaming kernels on the BA-House x100k sample (sparse routing switched off): per-kernel launch times and roofline
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
GNNX_SPARSE_RESIDENT=0 timeout 1200 python bench.py --steps 2 --warmup 1 --workload ba100k --targets ${2:-1024} --no-cpu-baseline > $O/bench_stream.json 2> $O/bench_stream.err; echo "rc=$?" >> $O/bench_stream.err
tail -2 $O/bench_stream.err
python -c "
import json;d=json.loads(open('$O/bench_stream.json').read().strip().splitlines()[-1]);print('value',d['value'],'ms',d['ms_per_step']); print(d['config']['routing_rank0']); print(json.dumps(d['roofline'])[:1800])"
