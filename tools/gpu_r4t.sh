#!/bin/bash
# round 4, session t (the last GPU seconds): the last round of owned edges dealt from the top thread down - GPU suite, headline line, rocprofv3 stats, timelines
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4t}; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming\|largest target n\|cost table" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r04_parity_lines.txt
timeout 60 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
timeout 120 python bench.py 2>$O/bench_default.err | tail -1 > $O/r04_bench_syn1_default.json
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1_loop -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-only > $GRAFT_REPO_ROOT/$O/r04_bench_syn1_loop_only_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof_syn1_loop -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_kernel_stats_syn1_loop_only.csv; rm -rf $O/prof_syn1_loop
timeout 60 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/r04_timeline_sparse_resident_syn1_n310.txt
timeout 60 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/r04_timeline_sparse_resident_syn1_one_wave.txt
for f in $O/r04_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), d['roofline']['kernel'][:34], round(d['roofline']['frac'],4), d.get('parity',{}).get('rule','')[:80])" 2>/dev/null; done
head -2 $O/r04_kernel_stats_syn1_loop_only.csv | cut -c1-200
tail -4 $O/r04_timeline_sparse_resident_syn1_n310.txt | cut -c1-220
