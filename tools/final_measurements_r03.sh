#!/bin/bash
# Round-3 measurement session on the GPU box: every artefact under profiles/r03_* comes from this script (tag = $1).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final_r03}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -s 2>&1 > $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -h "well-conditioned\|beyond 1e-5 (id\|config4 \[full\|AUC \|\[full\]\|\[early\]" $O/pytest_gpu.log | grep -v "^E " | cut -c1-900 > $O/r03_parity_lines.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r03_bench_syn1_default.json
timeout 300 python bench.py --workload syn4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_syn4.json
timeout 300 python bench.py --workload syn5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_syn5.json
timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r03_bench_config4.json
timeout 600 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_ba100k_2048targets.json
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r03_bench_ba100k_16384targets.json
GNNX_SPARSE_RESIDENT=0 timeout 900 python bench.py --workload ba100k --targets 1024 --steps 2 --warmup 1 --no-cpu-baseline --loop-only 2>/dev/null | tail -1 > $O/r03_bench_ba100k_1024targets_dense_streaming.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r03_bench_syn1_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof_syn1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r03_kernel_stats_syn1.csv
rm -rf $O/prof_syn1
timeout 300 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/r03_timeline_sparse_resident_syn1_n310.txt
timeout 300 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/r03_timeline_sparse_resident_syn1_one_wave.txt
# the dense streaming pair (k_conv / k_mask) on the BA-House x100k streaming set, the access-pattern micro-benchmark, method=att
GNNX_SPARSE_RESIDENT=0 timeout 300 python tools/probe_conv.py 1024 1:0,1:1024,2:0 2>/dev/null | grep WIDE > $O/r03_conv_streaming_set.txt
GNNX_SPARSE_RESIDENT=0 timeout 300 python tools/probe_conv_timeline.py 2>/dev/null | grep -v amdgpu > $O/r03_timeline_k_conv_ba100k.txt
(cd tools/micro && for a in "4992 3" "1056 64" "4992 1"; do timeout 60 ./stream_pattern $a; done) > $O/r03_micro_stream_pattern.txt 2>&1
timeout 300 python tools/probe_att.py 2>/dev/null | grep -v Warning > $O/r03_method_att_syn1_400targets.txt
bash tools/gpu_pmc_stream.sh ${1:-final_r03}/pmc_stream > /dev/null 2>&1
bash tools/gpu_pmc.sh ${1:-final_r03}/pmc_syn1 syn1 > /dev/null 2>&1
bash tools/gpu_pmc.sh ${1:-final_r03}/pmc_ba100k ba100k > /dev/null 2>&1
cat $O/pytest_gpu_tail.txt
for f in $O/r03_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), d['roofline']['kernel'][:34], round(d['roofline']['frac'],4), d.get('parity',{}).get('rule','')[:80])"; done
head -4 $O/r03_kernel_stats_syn1.csv | cut -c1-200
