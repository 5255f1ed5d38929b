#!/bin/bash
# the driver's command (20 steps / 5 warm-up) by prepare workers, alternating (edge draw for every batch: the round's closing default)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6wk}; mkdir -p $O
for i in 1 2 3; do
  for w in 2 3; do
    GNNX_PIPE_WORKERS=$w timeout 300 python bench.py --no-cpu-baseline > $O/w${w}_$i.json 2> $O/w${w}_$i.err
    python - <<PY
import json
r = json.loads(open("$O/w${w}_$i.json").read().strip().split("\n")[-1]); e = r["end_to_end_stage_ms"]
print("workers=$w run $i value %.1f k  reps %s  prepare %.2f" % (r["value"] / 1e3, [round(x / 1e3) for x in e["repetitions"]["values"]], e.get("prepare_ms", 0)))
PY
  done
done
