#!/bin/bash
# round 5, session a: HEAD on the GPU before any kernel change - the latencies the chain bound is priced with, the GPU suite, the headline line with the
# executed-work roofline, config 5 END TO END with the block-granular edge draw (never measured on the GPU box before), the instruction-count PMC pass
# the issue bound reads, the dense streaming baseline
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5a}; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/host.txt; cat /sys/fs/cgroup/cpu.max >> $O/host.txt 2>/dev/null
timeout 120 tools/micro/chain_latency > $O/r05_chain_latency.txt 2>&1
timeout 600 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log > $O/pytest_gpu_tail.txt
grep -n "FAILED\|^ERROR" $O/pytest_gpu.log | head -20 >> $O/pytest_gpu_tail.txt
grep -h "every decision identical\|300 epochs from the seeds\|same decisions, beyond\|: tie   id\|ba100k (\|well-conditioned\|beyond 1e-5 (id\|config4 \[full\|config4 (64\|AUC \|\[full\]\|\[early\]\|k_sparse_large vs streaming\|largest target n\|cost table" $O/pytest_gpu.log | grep -v "^E " | cut -c1-1300 > $O/r05_parity_lines.txt
cat $O/pytest_gpu_tail.txt
timeout 300 python bench.py 2>$O/bench_default.err | tail -1 > $O/r05_bench_syn1_default.json
timeout 500 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 2 --no-cpu-baseline 2>$O/bench_ba100k.err | tail -1 > $O/r05_bench_ba100k_16384targets.json
cd /tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-gate --loop-only"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_insts -- $B > /dev/null 2>$GRAFT_REPO_ROOT/$O/pmc_insts.err
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_lds -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/r05_pmc_summary_syn1_loop_only.json $O/r05_pmc_per_kernel_syn1_loop_only.csv $O/pmc_insts $O/pmc_lds > /dev/null
rm -rf $O/pmc_insts $O/pmc_lds
GNNX_SPARSE_RESIDENT=0 timeout 600 python bench.py --steps 2 --warmup 1 --workload ba100k --targets 1024 --no-cpu-baseline 2>$O/bench_stream.err | tail -1 > $O/r05_bench_ba100k_1024targets_dense_streaming.json
for f in $O/r05_bench_*.json; do python -c "
import json,sys; d=json.load(open('$f')); r=d['roofline']; print('$f'.split('/')[-1], round(d['value']), round(d['ms_per_step'],3), 'loop', round(d.get('loop_only',{}).get('ms_per_step',0),3), r['kernel'][:34], r['bound'], round(r['frac'],4), json.dumps(r.get('model',{}).get('frac_by_bound')), json.dumps(d.get('end_to_end_stage_ms',{}).get('repetitions',{}).get('values')), d.get('parity',{}).get('rule','')[:80])" 2>&1 | tail -3; done
tail -3 $O/bench_ba100k.err; tail -2 $O/bench_stream.err
cat $O/r05_chain_latency.txt | head -40
