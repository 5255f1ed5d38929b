#!/bin/bash
# Measurement tool (GPU box): A/B of the resident-kernel size limit (GNNX_RESIDENT_MAX_BLOCKS) on the bench workloads.
cd "${GRAFT_REPO_ROOT:-.}"
for nb in 1 2 3; do
  for wl in syn1 syn5; do
    GNNX_RESIDENT_MAX_BLOCKS=$nb timeout 300 python bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max_blocks',$nb,'$wl',d['value'],d['unit'],d['ms_per_step'],'ms/step')"
  done
done
