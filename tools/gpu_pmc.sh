#!/bin/bash
# counter passes of the bench command (syn1): HBM traffic, SQ / MFMA, LDS.  One counter group per run, no other trace domains.
O=gpurun_out/$1; mkdir -p $O
W=${2:-syn1}
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-gate --workload $W --targets 2048"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_lds -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-gate --workload $W --targets 2048 > $GRAFT_REPO_ROOT/$O/bench_under_trace.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_summary_$W.json $O/pmc_per_kernel_$W.csv $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats_$W.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds $O/stats
head -5 $O/kernel_stats_$W.csv | cut -c1-160
