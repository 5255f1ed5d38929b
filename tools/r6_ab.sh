#!/bin/bash
# round 6 A/B: the 16 384-motif-target set end to end for several XL thresholds / prepare-worker counts
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6ab}; mkdir -p $O
for cfg in "2048 2" "4096 2" "2048 4" "16383 4"; do
  set -- $cfg
  GNNX_XL_MIN_N=$1 GNNX_PIPE_WORKERS=$2 timeout 900 python bench.py --workload ba100k --targets 16384 --steps 4 --warmup 1 --no-cpu-baseline --no-parity-gate > $O/ba100k_16384_xlmin$1_w$2.json 2> $O/ba100k_16384_xlmin$1_w$2.err
  python - <<PY
import json
r = json.load(open("$O/ba100k_16384_xlmin$1_w$2.json"))
print("xl_min_n=$1 workers=$2 value", round(r["value"]), "ms/step", round(r["ms_per_step"], 1))
print("   stages", {k: round(v, 1) for k, v in r.get("end_to_end_stage_ms", {}).items() if isinstance(v, float)})
PY
done
