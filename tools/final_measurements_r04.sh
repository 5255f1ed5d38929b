#!/bin/bash
# Round-4 measurement session on the GPU box: every artefact under profiles/r04_* comes from this script (tag = $1).
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final_r04}; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s 2>&1 > $O/pytest_gpu.log; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r04_parity_lines.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r04_bench_syn1_default.json
timeout 300 python bench.py --workload syn4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_syn4.json
timeout 300 python bench.py --workload syn5 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_syn5.json
timeout 600 python bench.py --workload config4 --steps 5 --warmup 2 2>/dev/null | tail -1 > $O/r04_bench_config4.json
timeout 600 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_ba100k_2048targets.json
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r04_bench_ba100k_16384targets.json
GNNX_SPARSE_RESIDENT=0 timeout 900 python bench.py --workload ba100k --targets 1024 --steps 2 --warmup 1 --no-cpu-baseline --loop-only 2>/dev/null | tail -1 > $O/r04_bench_ba100k_1024targets_dense_streaming.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r04_bench_syn1_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1_loop -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-only > $GRAFT_REPO_ROOT/$O/r04_bench_syn1_loop_only_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
find $O/prof_syn1_loop -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_kernel_stats_syn1_loop_only.csv
rm -rf $O/prof_syn1_loop
find $O/prof_syn1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r04_kernel_stats_syn1.csv
rm -rf $O/prof_syn1
for form in 0 1 2; do GNNX_XCONST=$form timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate --loop-only 2>/dev/null | tail -1 > $O/r04_bench_syn1_loop_only_form$form.json; done
timeout 300 python tools/probe_logging.py > $O/r04_loss_logging_explain_node.txt 2>&1
GNNX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 4 --steps 2 --warmup 1 --reps 1 --targets 4096 --no-single-gpu-leg > $O/r04_bench_sharded_4ranks_one_gpu_gloo.json 2> $O/bench_sharded.err
timeout 300 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/r04_timeline_sparse_resident_syn1_n310.txt
timeout 300 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/r04_timeline_sparse_resident_syn1_one_wave.txt
# (the dense streaming pair was not touched in round 4: its probes / micro-benchmark / PMC passes of round 3 stand - profiles/r03_*)
timeout 300 python tools/probe_att.py 2>/dev/null | grep -v Warning > $O/r04_method_att_syn1_400targets.txt
bash tools/gpu_pmc.sh ${1:-final_r04}/pmc_syn1 syn1 > /dev/null 2>&1
cat $O/pytest_gpu_tail.txt
for f in $O/r04_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), d['roofline']['kernel'][:34], round(d['roofline']['frac'],4), d.get('parity',{}).get('rule','')[:80])"; done
head -4 $O/r04_kernel_stats_syn1.csv | cut -c1-200
