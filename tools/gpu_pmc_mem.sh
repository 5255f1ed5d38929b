#!/bin/bash
# memory-path counters of the dense streaming kernels (sparse routing off) on the BA-House x100k 1024-target sample:
# where do k_conv / k_mask wait - L1 miss queue, L2 tags, the fabric (EA) or DRAM credits?
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GNNX_SPARSE_RESIDENT=0
B="python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --iters 12 --no-cpu-baseline --no-parity-gate --no-graph --loop-only --workload ba100k --targets 1024"
P() { timeout 400 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$1 -- $B > /dev/null 2>&1; echo "pass $1 rc=$?"; }
P m1 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P m2 "TCC_EA_RDREQ_sum TCC_EA_RDREQ_LEVEL_sum TCC_EA_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
P m3 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr"
P m4 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_summary_mem.json $O/pmc_per_kernel_mem.csv $O/m1 $O/m2 $O/m3 $O/m4
rm -rf $O/m1 $O/m2 $O/m3 $O/m4
python -c "
import json;d=json.load(open('$O/pmc_summary_mem.json'));c=d['counters_mean_per_launch']
for k in c:
    if 'k_conv' in k or 'k_mask' in k:
        print(k, d['launches'][k]); print('   ', {a: round(b, 1) for a, b in c[k].items()})"
