#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6ab}; mkdir -p $O
for cfg in "512 2" "512 4"; do
  set -- $cfg
  GNNX_XL_MIN_N=$1 GNNX_PIPE_WORKERS=$2 timeout 900 python bench.py --workload ba100k-all --steps 2 --warmup 1 --no-cpu-baseline > $O/all_xlmin$1_w$2.json 2> $O/all_xlmin$1_w$2.err
  python - <<PY
import json
r = json.load(open("$O/all_xlmin$1_w$2.json"))
print("xl_min_n=$1 workers=$2 value", round(r["value"]), "ms/step", round(r["ms_per_step"]), "every node: loop", round(r["every_node"]["loop_s"], 2), "e2e pipelined", round(r["every_node"]["end_to_end_pipelined_s"], 2), "alone", round(r["every_node"]["end_to_end_batches_alone_s"], 2))
print("   pipelined ms per stratum", [round(s.get("pipelined_ms_per_batch", 0)) for s in r["strata"]], "loop", [round(s.get("loop_ms", 0)) for s in r["strata"]], "alone", [round(s.get("one_batch_alone_ms", 0)) for s in r["strata"]])
print("   stage ms of the last XL strata:", r["strata"][-2].get("stage_ms"), r["strata"][-1].get("stage_ms"))
PY
done
