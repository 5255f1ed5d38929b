#!/bin/bash
# round 3: kernel iteration work - emulator-twin tests on hardware, the bench line, phase timelines of the largest / a small syn1 target
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_emu_kernels.py tests/test_gpu_parity.py -m gpu -q -x > $O/kernel_tests.log 2>&1; tail -3 $O/kernel_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_syn1.json 2> $O/bench_syn1.err
python - <<PY
import json
d=json.load(open("$O/bench_syn1.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "loop_only", d["loop_only"]["value"], d["loop_only"]["ms_per_step"])
print("parity", d.get("parity",{}).get("rule"))
PY
timeout 300 python tools/probe_sparse.py 0 > $O/probe_sparse_0.log 2>&1; grep -v amdgpu $O/probe_sparse_0.log | tail -14
timeout 300 python tools/probe_sparse.py 150 > $O/probe_sparse_150.log 2>&1; grep -v amdgpu $O/probe_sparse_150.log | tail -14
