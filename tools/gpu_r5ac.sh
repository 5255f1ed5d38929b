#!/bin/bash
# round 5, session ac: run-to-run and repetition-to-repetition spread of the driver's 20-batch regions: CPU affinity (the container may run on
# any of the host's 256 CPUs with a 16-core quota), five alternations
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ac}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
nproc > $O/host.txt; taskset -p $$ >> $O/host.txt; lscpu | grep -i "numa\|socket\|thread" >> $O/host.txt
for i in 1 2 3 4 5; do
  timeout 300 $B 2>/dev/null | tail -1 > $O/bench_syn1_free_$i.json
  timeout 300 taskset -c 0-15 $B 2>/dev/null | tail -1 > $O/bench_syn1_pin16_$i.json
  timeout 300 taskset -c 0-31 $B 2>/dev/null | tail -1 > $O/bench_syn1_pin32_$i.json
done
cat $O/host.txt
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'h2d', round(e.get('h2d_scatter_enqueue_ms',0),2), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
