#!/bin/bash
# the driver's command with the edge-only draw for small batches too (GNNX_PIPE_EDGE_MIN = 0) against the default (2e7), alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6em}; mkdir -p $O
for i in 1 2 3; do
  for em in 2e7 0; do
    GNNX_PIPE_EDGE_MIN=$em timeout 300 python bench.py --no-cpu-baseline > $O/em${em}_$i.json 2> $O/em${em}_$i.err
    python - <<PY
import json
r = json.loads(open("$O/em${em}_$i.json").read().strip().split("\n")[-1]); e = r["end_to_end_stage_ms"]
print("edge_min=$em run $i value %.1f k  reps %s  prepare %.2f rng %.2f wait %.2f h2d %.2f host core-s/step %.4f" % (r["value"] / 1e3, [round(x / 1e3) for x in e["repetitions"]["values"]],
      e.get("prepare_ms", 0), e.get("host_rng_ms", 0), e.get("wait_for_rng_ms", 0), e.get("h2d_scatter_enqueue_ms", 0), e["host_bound_projection"]["host_core_seconds_per_step"]))
PY
  done
done
