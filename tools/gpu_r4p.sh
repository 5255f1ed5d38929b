#!/bin/bash
# round 4, session p: fast head (h1 vs h0), gather unroll 4 (u4), unconditional g3 loads (g = the shipped default), timelines, the GPU suite on g
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
show() { python -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', 'loop ms', round(d['ms_per_step'],4), 'kernel ms', round(r.get('avg_launch_us', 0)/1e3,4), 'frac', round(r['frac'],4), d.get('parity',{}).get('rule','')[:60])" 2>&1 | tail -1; }
for rep in 1 2; do
for v in h0 h1 u4 g; do
  GNNX_LIBRARY_PATH=$GRAFT_REPO_ROOT/tools/_build/ab/libgnnx_hip_$v.so timeout 200 python bench.py --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${v}_$rep.json 2> $O/loop_${v}_$rep.err; show $O/loop_${v}_$rep.json "syn1 $v run $rep"
done; done
timeout 200 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/timeline_n310_g.txt
timeout 200 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/timeline_onewave_g.txt
tail -11 $O/timeline_n310_g.txt | cut -c1-200; tail -4 $O/timeline_onewave_g.txt | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r04_parity_lines.txt
cat $O/pytest_gpu_tail.txt
for w in syn5 syn4; do timeout 200 python bench.py --workload $w --loop-only --steps 20 --warmup 5 --no-cpu-baseline --reps 1 > $O/loop_${w}_g.json 2> /dev/null; show $O/loop_${w}_g.json "$w g"; done
timeout 300 python bench.py --workload config4 --loop-only --steps 5 --warmup 2 --no-cpu-baseline --reps 1 > $O/loop_config4_g.json 2> /dev/null; show $O/loop_config4_g.json "config4 g"
