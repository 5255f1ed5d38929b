cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 100 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 python bench.py 2>/dev/null | tail -1 > gpurun_out/final/r01_bench_sparse_syn1_default.json
timeout 500 python bench.py --workload ba100k --targets 2048 --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/final/r01_bench_sparse_ba100k_2048targets.json
timeout 300 python bench.py --workload syn4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/r01_bench_sparse_syn4.json
timeout 300 python bench.py --workload syn5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final/r01_bench_sparse_syn5.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_syn1 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_ba100k -- python $GRAFT_REPO_ROOT/bench.py --workload ba100k --targets 2048 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_fetch -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/pmc_write -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for f in gpurun_out/final/*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f', round(d['value']), d['ms_per_step'], d['roofline']['kernel'][:30], round(d['roofline']['frac'],3), d['config']['routing'])"; done
find gpurun_out/final -name "*.csv" | head -20
