#!/bin/bash
# round 3: the sharded path - shard balance test on one GPU, and bench.py --gpus 8 with eight gloo ranks sharing the one GPU of the box
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_scaling.py -m gpu -q -s > $O/scaling_test.log 2>&1; grep -h "cost table\|passed\|failed" $O/scaling_test.log | cut -c1-400
export GNNX_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --steps 3 --warmup 1 --no-single-gpu-leg > $O/bench_sharded_8ranks_one_gpu_gloo.json 2> $O/bench_sharded_8ranks.err
tail -3 $O/bench_sharded_8ranks.err | cut -c1-300
python - <<PY
import json
d=json.load(open("$O/bench_sharded_8ranks_one_gpu_gloo.json"))
c=d["config"]
print("value", d["value"], "ms_per_step", d["ms_per_step"]); print("sum_n2_per_rank", c.get("sum_n2_per_rank")); print("modelled", c.get("modelled_gpu_us_per_rank")); print("gathered", c.get("gathered_edge_entries_per_rank")); print("cost table", c.get("cost_table_us"))
PY
