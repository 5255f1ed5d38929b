#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_full_configs.py -m gpu -q -x --timeout=600 -s 2>&1 | grep -E "syn[145] \[|config4 \[|passed|failed" | cut -c1-200
