#!/bin/bash
# counter passes of the dense streaming kernels (sparse routing off) on the BA-House x100k 1024-target sample
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GNNX_SPARSE_RESIDENT=0
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --iters 20 --no-cpu-baseline --no-parity-gate --no-graph --workload ba100k --targets 1024"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_sq -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_lds -- $B > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_summary_stream.json $O/pmc_per_kernel_stream.csv $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
python -c "
import json;d=json.load(open('$O/pmc_summary_stream.json'));c=d['counters_mean_per_launch']
for k in c:
    if 'k_conv' in k or 'k_mask' in k or 'node_head' in k:
        v=c[k]; print(k, 'launches', d['launches'][k], 'HBM MB', round(d['hbm_bytes_per_launch'][k]/1e6,1), 'wait', round(v['SQ_WAIT_ANY']/max(1,v['SQ_WAVE_CYCLES']),3), 'issue', round(v['SQ_ACTIVE_INST_ANY']/max(1,v['SQ_WAVE_CYCLES']),3), 'mfma/busy', round(v['SQ_VALU_MFMA_BUSY_CYCLES']/max(1,v['SQ_BUSY_CYCLES']),3), 'waves', round(v['SQ_WAVES']), 'gui', round(v['GRBM_GUI_ACTIVE']))"
