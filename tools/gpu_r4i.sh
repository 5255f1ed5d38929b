#!/bin/bash
# round 4, session i: the round's last commit - the whole GPU suite, smoke, the default bench line
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err | cut -c1-200; python -c "
import json;d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]);print('value',round(d['value']),'ms',round(d['ms_per_step'],3),'steps',d['steps'],'warmup',d['warmup'],'roofline',round(d['roofline']['frac'],4),'cpu',d['cpu_baseline']['value'], d['cpu_baseline'].get('cgroup_cpu_quota_cores'), d['end_to_end_stage_ms']['repetitions'])"
