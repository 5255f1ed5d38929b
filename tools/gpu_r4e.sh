#!/bin/bash
# round 4, session e: k_att at scale, config-4 full horizon + outcome sets, (ba100k windows when the fixture is there)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_options.py tests/test_decision_parity.py tests/test_gpu_full_configs.py -m gpu -q --timeout=900 -s -k "att_kernel_at_scale or full_horizon_decisions_config4 or config4_64 or ba100k_route_stratified" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -h "method=att at scale\|300 epochs from the seeds\|three numbers\|config4 (64 graphs)\|ba100k\|FAILED\|passed\|failed\| Error" $O/pytest_gpu.log | cut -c1-1500 | head -30
timeout 300 python tools/probe_logging.py 2>&1 | grep -v amdgpu | tail -3
