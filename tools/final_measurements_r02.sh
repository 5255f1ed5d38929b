#!/bin/bash
# Round-2 measurement session on the GPU box: every artefact under profiles/r02_* comes from this script.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r02; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout=600 2>&1 | tail -2 > $O/pytest_gpu_tail.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/r02_bench_syn1_default.json
timeout 600 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ba100k_2048targets.json
timeout 900 python bench.py --workload ba100k --targets 16384 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ba100k_16384targets.json
GNNX_SPARSE_RESIDENT=0 timeout 900 python bench.py --workload ba100k --targets 1024 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_ba100k_1024targets_dense_streaming.json
timeout 300 python bench.py --workload syn4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_syn4.json
timeout 300 python bench.py --workload syn5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r02_bench_syn5.json
timeout 600 python tools/config4_mutag_like.py 2>/dev/null | tail -1 > $O/r02_config4_mutag_like.json
GNNX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
    bench.py --gpus 2 --steps 2 --warmup 1 --targets 4096 2>/dev/null | tail -1 > $O/r02_bench_sharded_2ranks_one_gpu_gloo.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ba100k -- python $GRAFT_REPO_ROOT/bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_syn1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_kernel_stats_syn1.csv
find $O/prof_ba100k -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r02_kernel_stats_ba100k_2048targets.csv
rm -rf $O/prof_syn1 $O/prof_ba100k
timeout 300 python tools/probe_sparse.py 0 2>/dev/null | grep -v amdgpu > $O/r02_timeline_sparse_resident_syn1_n310.txt
timeout 300 python tools/probe_sparse.py 150 2>/dev/null | grep -v amdgpu > $O/r02_timeline_sparse_resident_syn1_one_wave.txt
timeout 300 python tools/probe_large.py 0 2>/dev/null | grep -v amdgpu > $O/r02_timeline_sparse_large_ba100k_n2460.txt
timeout 600 python tools/probe_large.py 0 16384 2>/dev/null | grep -v amdgpu > $O/r02_timeline_sparse_large_ba100k_n5600.txt
timeout 900 python tools/probe_classes.py 16384 2>/dev/null | grep -v amdgpu > $O/r02_size_classes_ba100k_16384targets.txt
bash tools/gpu_pmc.sh final_r02/pmc_syn1 syn1 > /dev/null 2>&1
bash tools/gpu_pmc.sh final_r02/pmc_ba100k ba100k > /dev/null 2>&1
cat $O/pytest_gpu_tail.txt
for f in $O/r02_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), d['roofline']['kernel'][:40], d.get('parity',{}).get('rule','')[:90])"; done
cat $O/r02_config4_mutag_like.json | cut -c1-400
