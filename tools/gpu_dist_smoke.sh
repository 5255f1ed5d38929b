#!/bin/bash
# the sharded bench path on a 1-GPU box: two ranks share the GPU, collectives over gloo (the driver's real runs use RCCL)
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
GNNX_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --targets 2048 > $O/bench_dist2.json 2> $O/bench_dist2.err; echo "rc=$?" >> $O/bench_dist2.err
timeout 600 python bench.py --steps 3 --warmup 1 --workload ba100k --targets 2048 --no-cpu-baseline > $O/bench_ba100k_1gpu.json 2> $O/bench_ba100k_1gpu.err; echo "rc=$?" >> $O/bench_ba100k_1gpu.err
tail -5 $O/bench_dist2.err; cut -c1-1500 $O/bench_dist2.json; tail -3 $O/bench_ba100k_1gpu.err; cut -c1-600 $O/bench_ba100k_1gpu.json
