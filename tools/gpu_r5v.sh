#!/bin/bash
# round 5, session v: what makes single 20-batch repetitions slow (169 k beside 250 k)?  The container's CPU quota (16 cores of 256): the kernel
# throttles a cgroup that used its quota within a 100 ms period until the next one.  cpu.stat's nr_throttled / throttled_usec around every run,
# for the default host settings, fewer RNG threads, two prepare workers, sleeping waits
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5v}; mkdir -p $O
export TMPDIR=/tmp
cat /sys/fs/cgroup/cpu.max > $O/cpu_max.txt 2>/dev/null; nproc >> $O/cpu_max.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
stat() { grep -h "nr_throttled\|throttled_usec\|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
run() { # tag env...
  tag=$1; shift
  s0=$(stat)
  env "$@" GNNX_LIBRARY_PATH=tools/_build/ab/libgnnx_hip_r224s.so timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_$tag.json
  echo "$tag | before: $s0 | after: $(stat)" >> $O/throttle.txt
}
for i in 1 2; do
  run default_$i A=1
  run rng8_$i GNNX_RNG_THREADS=8
  run rng4_$i GNNX_RNG_THREADS=4
  run w2_$i GNNX_PIPE_WORKERS=2
  run w2_rng8_$i GNNX_PIPE_WORKERS=2 GNNX_RNG_THREADS=8
  run blocking_$i GNNX_BLOCKING_SYNC=1
  run w2_rng8_blocking_$i GNNX_PIPE_WORKERS=2 GNNX_RNG_THREADS=8 GNNX_BLOCKING_SYNC=1
done
cat $O/cpu_max.txt; cat $O/throttle.txt
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'rng', round(e.get('host_rng_ms',0),2), 'host core-s/step', round(e['host_bound_projection']['host_core_seconds_per_step'],4))" 2>&1 | tail -1; done
