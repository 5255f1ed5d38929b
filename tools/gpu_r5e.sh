#!/bin/bash
# round 5, session e: the seeded draw with the engine walked on the DEVICE (gnnx_mt_edge_words + gnnx_host_transform_edge_words) - GPU suite with the pair
# workgroups and the new draw, config 5 end to end with the host walk and with the device walk on the same box, the headline line
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5e}; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|: other id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming\|largest target n\|cost table" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r05_parity_lines.txt
cat $O/pytest_gpu_tail.txt
B="python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline"
GNNX_PIPE_DEVICE_WALK=0 timeout 500 $B 2>$O/bench_ba100k_hostwalk.err | tail -1 > $O/r05_bench_ba100k_16384targets_host_walk.json
timeout 500 $B 2>$O/bench_ba100k.err | tail -1 > $O/r05_bench_ba100k_16384targets.json
timeout 300 python bench.py 2>$O/bench_default.err | tail -1 > $O/r05_bench_syn1_default.json
for f in $O/r05_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r['bound'], round(r['frac'],4), [round(v/1000) for v in e['repetitions']['values']], 'prepare', round(e['prepare_ms'],1), 'rng', round(e['host_rng_ms'],1), 'walk', round(e.get('device_walk_ms',0),1), 'xform', round(e.get('host_transform_ms',0),1), 'hostcpu_s', round(e['host_bound_projection']['host_core_seconds_per_step'],3), 'knee', round(e['host_bound_projection']['knee_n_gpus'],1), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],1))" 2>&1 | tail -1; done
tail -2 $O/bench_ba100k.err
