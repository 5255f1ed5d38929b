#!/bin/bash
# round 5, session l: pipeline workers x depth on the headline with pair workgroups, three alternating rounds, 7 repetitions of 50 batches each
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5l}; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --reps 7"
for i in 1 2 3; do
  for cfg in "3 4" "4 6" "5 6" "4 8" "6 8" "3 6"; do
    set -- $cfg
    GNNX_PIPE_WORKERS=$1 GNNX_PIPE_DEPTH=$2 timeout 200 $B 2>/dev/null | tail -1 > $O/bench_syn1_w$1_d$2_run$i.json
  done
done
python - <<PY
import json, glob, collections, statistics
rows = collections.defaultdict(list)
for f in sorted(glob.glob("$O/bench_syn1_w*_run*.json")):
    d = json.load(open(f)); key = f.split("/")[-1].split("_run")[0].replace("bench_syn1_", "")
    rows[key].append((d["value"], d["end_to_end_stage_ms"]["repetitions"]["spread_pct"], d["end_to_end_stage_ms"]["prepare_ms"]))
for k, v in rows.items():
    print(k, "medians", [round(x[0] / 1000, 1) for x in v], "mean", round(statistics.mean(x[0] for x in v) / 1000, 1), "spread %", [round(x[1], 1) for x in v], "prepare ms", [round(x[2], 2) for x in v])
PY
