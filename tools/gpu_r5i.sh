#!/bin/bash
# round 5, session i: the sharded bench path at HEAD (4 gloo ranks on one GPU: device engine walk per shard, blocking waits, the all-gather hook), the
# generic-width instantiations' speed, syn4 / syn5 / config 4 lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5i}; mkdir -p $O
export TMPDIR=/tmp
GNNX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 4 --steps 2 --warmup 1 --targets 16384 > $O/r05_bench_sharded_4ranks_one_gpu_gloo.json 2> $O/bench_dist4.err; echo "rc=$?" >> $O/bench_dist4.err
tail -4 $O/bench_dist4.err | cut -c1-300
python -c "
import json; d=json.loads(open('$O/r05_bench_sharded_4ranks_one_gpu_gloo.json').read().strip().splitlines()[-1]); e=d['end_to_end_stage_ms']; print('sharded x4 on one GPU:', round(d['value']), 'ms/step', round(d['ms_per_step'],1), 'loop', round(d['loop_only']['ms_per_step'],1), 'hostcpu_s', round(e['host_bound_projection']['host_core_seconds_per_step'],3), 'walk', round(e.get('device_walk_ms',0),1), 'xform', round(e.get('host_transform_ms',0),1), 'single', d.get('single_gpu_same_workload',{}).get('value'))"
timeout 200 python tools/probe_generic_widths.py 2>/dev/null | grep -v amdgpu > $O/r05_generic_widths_syn1.txt; cat $O/r05_generic_widths_syn1.txt
timeout 200 python bench.py --workload syn5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn5.json
timeout 200 python bench.py --workload syn4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn4.json
timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_config4.json
for f in $O/r05_bench_syn*.json $O/r05_bench_config4.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r.get('bound'), round(r['frac'],3), d.get('parity',{}).get('rule','')[:70])" 2>&1 | tail -1; done
