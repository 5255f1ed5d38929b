#!/bin/bash
# round 5, session w: the new defaults (room: mixed kernel 224 registers + 5 KB of LDS free, service kernels 32 registers; torch's CPU pool and
# the RNG threads sized to the container's quota) - GPU suite, then the driver's command against the pre-room kernel build, with cpu.stat
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5w}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log | tee $O/pytest_gpu_tail.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
stat() { grep -h "nr_throttled\|throttled_usec\|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
run() { # tag env...
  tag=$1; shift
  s0=$(stat)
  env "$@" timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_$tag.json
  echo "$tag | before: $s0 | after: $(stat)" >> $O/throttle.txt
}
for i in 1 2 3; do
  run new_$i A=1
  run new_w2_$i GNNX_PIPE_WORKERS=2
  run preroom_$i GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_poolonly.so
done
B="python bench.py --no-cpu-baseline --reps 5 --steps 300 --warmup 10"
for i in 1 2; do
  run new_k300_$i A=1
  run new_w2_k300_$i GNNX_PIPE_WORKERS=2
  run preroom_k300_$i GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_poolonly.so
done
cat $O/throttle.txt
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'rng', round(e.get('host_rng_ms',0),2), 'host core-s/step', round(e['host_bound_projection']['host_core_seconds_per_step'],4), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
