#!/bin/bash
# A/B of the packed single-wave launch (GNNX_TINY_PACK = 0 / 12 / 16): the driver's command (syn1), syn4 / syn5 steady state, the (0, 32] stratum of the all-node sample
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6tiny}; mkdir -p $O
for pack in ${PACKS:-0 12 16}; do
  for wl in ${WLS:-syn1 syn4 syn5}; do
    GNNX_TINY_PACK=$pack timeout 600 python bench.py --workload $wl --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-parity-gate > $O/${wl}_pack$pack.json 2> $O/${wl}_pack$pack.err
    python - <<PY
import json
try:
    r = json.loads(open("$O/${wl}_pack$pack.json").read().strip().split("\n")[-1])
    print("pack=$pack $wl value %.1f k  ms/step %.3f  loop_only %.3f ms  launches %s" % (r["value"] / 1e3, r["ms_per_step"], r["loop_only"]["ms_per_step"], {k[:40]: round(v["ms_total"], 3) for k, v in r["roofline"]["launches"].items()}))
    print("    parity:", (r.get("parity") or {}).get("rule", "")[:200])
except Exception as e:
    print("pack=$pack $wl FAILED", e); print(open("$O/${wl}_pack$pack.err").read()[-1500:])
PY
  done
done
