#!/bin/bash
# round 5, session ad: which NUMA node is the GPU on, and does pinning to the WHOLE node do what pinning to 16 / 32 of its CPUs does
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5ad}; mkdir -p $O
export TMPDIR=/tmp
python - > $O/numa.txt 2>&1 <<'P'
import torch, glob, os
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if k.startswith('pci')})
for d in glob.glob('/sys/class/drm/card*/device'):
    try: print(d, os.path.realpath(d), open(d + '/numa_node').read().strip(), open(d + '/vendor').read().strip())
    except Exception as e: print(d, e)
for n in glob.glob('/sys/devices/system/node/node*/cpulist'): print(n, open(n).read().strip())
P
cat $O/numa.txt
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
for i in 1 2 3; do
  timeout 300 taskset -c 0-63,128-191 $B 2>/dev/null | tail -1 > $O/bench_syn1_node0_$i.json
  timeout 300 taskset -c 64-127,192-255 $B 2>/dev/null | tail -1 > $O/bench_syn1_node1_$i.json
  timeout 300 taskset -c 0-31 $B 2>/dev/null | tail -1 > $O/bench_syn1_pin32_$i.json
  timeout 300 taskset -c 64-95 $B 2>/dev/null | tail -1 > $O/bench_syn1_pin32node1_$i.json
done
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); e=d['end_to_end_stage_ms']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), [round(v/1000) for v in e['repetitions']['values']], 'spread', round(e['repetitions']['spread_pct'],1), 'prepare', round(e.get('prepare_ms',0),2), 'h2d', round(e.get('h2d_scatter_enqueue_ms',0),2), 'one batch', round(d['pcie_inclusive']['batch_total_ms'],2))" 2>&1 | tail -1; done
