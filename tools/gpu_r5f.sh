#!/bin/bash
# round 5, session f: the decision-parity suite on the SHIPPED form (the logging / tracing instantiation of the algebraic constant-feature form), sleeping
# waits in the large-batch pipeline (host core-seconds per step), config 5 end to end with and without process-wide blocking sync
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5f}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|: other id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming\|largest target n\|cost table" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r05_parity_lines.txt
cat $O/pytest_gpu_tail.txt
B="python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline"
timeout 500 $B 2>$O/bench_ba100k.err | tail -1 > $O/r05_bench_ba100k_16384targets.json
GNNX_BLOCKING_SYNC=1 timeout 500 $B 2>$O/bench_ba100k_bs.err | tail -1 > $O/r05_bench_ba100k_16384targets_blocking_sync.json
timeout 300 python bench.py --no-cpu-baseline 2>$O/bench_default.err | tail -1 > $O/r05_bench_syn1_nocpu.json
for f in $O/r05_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r['bound'], round(r['frac'],4), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e['prepare_ms'],1), 'rng', round(e['host_rng_ms'],1), 'walk', round(e.get('device_walk_ms',0),1), 'xform', round(e.get('host_transform_ms',0),1), 'hostcpu_s', round(e['host_bound_projection']['host_core_seconds_per_step'],3), 'knee', round(e['host_bound_projection']['knee_n_gpus'],1), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],1))" 2>&1 | tail -1; done
grep -c "" $O/r05_parity_lines.txt; grep "every decision identical" $O/r05_parity_lines.txt | cut -c1-330
