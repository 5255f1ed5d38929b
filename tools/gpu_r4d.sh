#!/bin/bash
# round 4, session d: full-horizon decision tests with the calm-target rule, config-4 outcome sets, shard balance; the host draw of the
# 16 384-target set by threads / slice length / destination; prepare workers of the pipeline
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_decision_parity.py tests/test_gpu_full_configs.py tests/test_gpu_scaling.py -m gpu -q --timeout=900 -s -k "full_horizon or config4_64 or lpt_shards" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -h "300 epochs from the seeds\|three numbers\|config4 (64 graphs)\|FAILED\|passed\|failed\| Error\|measured" $O/pytest_gpu.log | cut -c1-1200 | head -30
timeout 900 python tools/probe_rng_big.py > $O/probe_rng_big.log 2>&1; cat $O/probe_rng_big.log | tail -32
for w in 2 3 4; do GNNX_PIPE_WORKERS=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1_workers$w.json 2> $O/bench_syn1_workers$w.err; python -c "
import json;d=json.loads(open('$O/bench_syn1_workers$w.json').read().strip().splitlines()[-1]);print('workers $w value',round(d['value']),'ms',round(d['ms_per_step'],3), json.dumps(d['end_to_end_stage_ms']['repetitions']))"; done
GNNX_PIPE_DEPTH=4 timeout 600 python bench.py --steps 20 --warmup 5 --no-parity-gate --no-cpu-baseline > $O/bench_syn1_depth4.json 2> $O/bench_syn1_depth4.err; python -c "
import json;d=json.loads(open('$O/bench_syn1_depth4.json').read().strip().splitlines()[-1]);print('depth 4 value',round(d['value']),'ms',round(d['ms_per_step'],3), json.dumps(d['end_to_end_stage_ms']['repetitions']))"
