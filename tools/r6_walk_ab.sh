#!/bin/bash
# the driver's command: host walk (default for one process) vs device walk of the edge draw, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6walk}; mkdir -p $O
for i in 1 2 3; do
  for dw in 0 1; do
    GNNX_PIPE_DEVICE_WALK=$dw timeout 300 python bench.py --no-cpu-baseline > $O/dw${dw}_$i.json 2> $O/dw${dw}_$i.err
    python - <<PY
import json
r = json.loads(open("$O/dw${dw}_$i.json").read().strip().split("\n")[-1]); e = r["end_to_end_stage_ms"]
print("device_walk=$dw run $i value %.1f k  reps %s  prepare %.2f rng %.2f core-s %.4f" % (r["value"] / 1e3, [round(x / 1e3) for x in e["repetitions"]["values"]], e.get("prepare_ms", 0), e.get("host_rng_ms", 0), e["host_bound_projection"]["host_core_seconds_per_step"]))
PY
  done
done
