#!/bin/bash
# Round-5 closing measurement session: every artefact under profiles/r05_* that DESIGN.md / README.md quote, at the round's last kernel commit.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final_r05}; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
timeout 900 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|: other id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming\|largest target n\|cost table" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r05_parity_lines.txt
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1 >> $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
timeout 120 tools/micro/chain_latency > $O/r05_chain_latency.txt 2>&1
timeout 60 tools/micro/permlane_swap > $O/r05_permlane_swap.txt 2>&1
python tools/chain_latency_json.py $O/r05_chain_latency.txt profiles/r05_chain_latency.json $O/r05_permlane_swap.txt > /dev/null   # (the bench runs below read it)
stat() { grep -h "nr_throttled\|throttled_usec\|nr_periods" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; }
echo "before the driver's command: $(stat)" > $O/r05_cpu_throttle_around_driver_command.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_default.err | tail -1 > $O/r05_bench_syn1_default.json
echo "after: $(stat)" >> $O/r05_cpu_throttle_around_driver_command.txt
for i in 2 3; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn1_default_run$i.json; done
timeout 400 python bench.py --steps 300 --warmup 10 --reps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn1_steady_state_300_batches.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1_loop -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-only > $GRAFT_REPO_ROOT/$O/r05_bench_syn1_loop_only_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_syn1 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r05_bench_syn1_under_rocprof.json 2>/dev/null
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-gate --loop-only"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_insts -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_lds -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_syn1_loop -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_kernel_stats_syn1_loop_only.csv; rm -rf $O/prof_syn1_loop
find $O/prof_syn1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_kernel_stats_syn1.csv; rm -rf $O/prof_syn1
python tools/pmc_summary.py $O/r05_pmc_summary_syn1_loop_only.json $O/r05_pmc_per_kernel_syn1_loop_only.csv $O/pmc_insts $O/pmc_lds $O/pmc_fetch $O/pmc_write > /dev/null
rm -rf $O/pmc_insts $O/pmc_lds $O/pmc_fetch $O/pmc_write
timeout 60 python tools/probe_sparse_waves.py 0 2>/dev/null | grep -v amdgpu > $O/r05_timeline_per_wave_syn1_n310.txt
timeout 60 python tools/probe_sparse_waves.py 150 2>/dev/null | grep -v amdgpu > $O/r05_timeline_per_wave_syn1_one_wave.txt
timeout 200 python bench.py --workload syn5 --steps 300 --warmup 10 --reps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn5.json
timeout 200 python bench.py --workload syn4 --steps 300 --warmup 10 --reps 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_syn4.json
timeout 300 python bench.py --workload config4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_config4.json
timeout 300 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_2048targets.json
timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>$O/bench_ba100k.err | tail -1 > $O/r05_bench_ba100k_16384targets.json
GNNX_BLOCKING_SYNC=1 timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_16384targets_blocking_sync.json
GNNX_PIPE_DEVICE_WALK=0 timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_16384targets_host_walk.json
GNNX_SPARSE_RESIDENT=0 timeout 600 python bench.py --steps 2 --warmup 1 --workload ba100k --targets 1024 --no-cpu-baseline 2>/dev/null | tail -1 > $O/r05_bench_ba100k_1024targets_dense_streaming.json
timeout 120 python tools/probe_att.py 2>/dev/null | grep -v Warning > $O/r05_method_att_syn1_400targets.txt
timeout 120 python tools/probe_logging.py 2>&1 | grep -v amdgpu > $O/r05_loss_logging_explain_node.txt
timeout 200 python tools/probe_generic_widths.py 2>/dev/null | grep -v amdgpu > $O/r05_generic_widths_syn1.txt
GNNX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 4 --steps 2 --warmup 1 --targets 16384 2>/dev/null | tail -1 > $O/r05_bench_sharded_4ranks_one_gpu_gloo.json
for f in $O/r05_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r['kernel'][:30], r.get('bound'), round(r['frac'],4), d.get('parity',{}).get('rule','')[:70])" 2>/dev/null; done
head -3 $O/r05_kernel_stats_syn1_loop_only.csv | cut -c1-200
cat $O/r05_generic_widths_syn1.txt $O/r05_method_att_syn1_400targets.txt $O/r05_loss_logging_explain_node.txt | cut -c1-300
