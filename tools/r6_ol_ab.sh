#!/bin/bash
# A/B of GNNX_OPAQUE_LANE (the lane's own indices declared modified at the top of the iteration): library variants built into tools/_build/
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6ol}; mkdir -p $O
for v in ${VARIANTS:-0 1 2 3}; do
  lib=""; [ "$v" != "0" ] && lib="$PWD/tools/_build/libgnnx_hip_ol$v.so"
  for wl in ${WLS:-syn1 syn5 ba100k}; do
    extra="--steps 100 --warmup 10"; [ "$wl" = "ba100k" ] && extra="--steps 6 --warmup 2"
    GNNX_LIBRARY_PATH=$lib timeout 900 python bench.py --workload $wl $extra --no-cpu-baseline --no-parity-gate > $O/${wl}_ol$v.json 2> $O/${wl}_ol$v.err
    python - <<PY
import json
try:
    r = json.loads(open("$O/${wl}_ol$v.json").read().strip().split("\n")[-1])
    print("ol=$v $wl value %.1f k  ms/step %.3f  loop_only %.3f ms  launches %s  reps %s" % (r["value"] / 1e3, r["ms_per_step"], r["loop_only"]["ms_per_step"], {k[:28]: round(v["ms_total"], 3) for k, v in r["roofline"]["launches"].items()},
          [round(x / 1e3) for x in r.get("end_to_end_stage_ms", {}).get("repetitions", {}).get("values", [])]))
except Exception as e:
    print("ol=$v $wl FAILED", e); print(open("$O/${wl}_ol$v.err").read()[-800:])
PY
  done
done
