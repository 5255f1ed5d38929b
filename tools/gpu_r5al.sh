#!/bin/bash
# round 5, session al: mask density + class probabilities in the kernels' logging forms (the reference's per-epoch print line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5al}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "logging or explainer_api or print_training or trace or pair or mixed or large" > $O/pytest_sub.log 2>&1; tail -2 $O/pytest_sub.log
timeout 120 python tools/probe_logging.py 2>&1 | grep -v amdgpu | tee $O/r05_loss_logging_explain_node.txt | cut -c1-300
