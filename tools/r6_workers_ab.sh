#!/bin/bash
# the driver's command (20 steps / 5 warm-up) by prepare workers and in-pipeline launch timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6wk}; mkdir -p $O
for cfg in "2 1" "3 1" "4 1" "2 0" "2 1"; do
  set -- $cfg
  GNNX_PIPE_WORKERS=$1 GNNX_PIPE_LAUNCH_MS=$2 timeout 300 python bench.py --no-cpu-baseline > $O/w$1_l$2.json 2> $O/w$1_l$2.err
  python - <<PY
import json
r = json.loads(open("$O/w$1_l$2.json").read().strip().split("\n")[-1])
print("workers=$1 launch_ms=$2 value %.1f k  ms/step %.3f  reps %s" % (r["value"] / 1e3, r["ms_per_step"], [round(x / 1e3) for x in r["end_to_end_stage_ms"]["repetitions"]["values"]]))
PY
done
