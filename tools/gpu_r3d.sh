#!/bin/bash
# round 3: pipeline tests + the end-to-end bench line
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_pipeline.py -m gpu -q -x > $O/pipeline_tests.log 2>&1; tail -15 $O/pipeline_tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_syn1.json 2> $O/bench_syn1.err; tail -5 $O/bench_syn1.err
python - <<PY
import json
d=json.load(open("$O/bench_syn1.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "loop_only", d["loop_only"]["value"], d["loop_only"]["ms_per_step"])
print("stages", d.get("end_to_end_stage_ms"))
print("roofline", {k: d["roofline"][k] for k in ("bound","achieved","peak","frac","avg_launch_us")})
print("cpu_baseline", d.get("cpu_baseline",{}).get("value"))
print("parity", d.get("parity",{}).get("rule"))
PY
