#!/bin/bash
# packed single-wave launch in the SATURATED regime: the motif-only 16 384-target set of BA-House x100k (74 % single-wave targets) and the all-node sample
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6sat}; mkdir -p $O
for pack in ${PACKS:-0 12 16}; do
  GNNX_TINY_PACK=$pack timeout 900 python bench.py --workload ba100k --steps 6 --warmup 2 --no-cpu-baseline --no-parity-gate > $O/ba100k_pack$pack.json 2> $O/ba100k_pack$pack.err
  python - <<PY
import json
try:
    r = json.loads(open("$O/ba100k_pack$pack.json").read().strip().split("\n")[-1])
    print("pack=$pack ba100k-16384 value %.1f k  ms/step %.2f  loop_only %.2f ms  launches %s" % (r["value"] / 1e3, r["ms_per_step"], r["loop_only"]["ms_per_step"], {k[:44]: round(v["ms_total"], 2) for k, v in r["roofline"]["launches"].items()}))
except Exception as e:
    print("pack=$pack FAILED", e); print(open("$O/ba100k_pack$pack.err").read()[-1500:])
PY
done
