#!/bin/bash
# round 5, session aa: k_att with hop pruning + chunked rows - its GPU tests (live-reference fixtures, autograd checks), then the 400-target timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5aa}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_options.py -m gpu -q -x -s -k "att" > $O/pytest_att.log 2>&1; tail -3 $O/pytest_att.log; grep -h "method=att" $O/pytest_att.log | cut -c1-600
timeout 300 python tools/probe_att.py 2>/dev/null | grep -v amdgpu | tee $O/r05_method_att_syn1_400targets.txt
