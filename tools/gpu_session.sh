#!/bin/bash
# ONE parametrised GPU session script (round 6; replaces the ~75 one-off gpu_r3*.sh ... gpu_r5*.sh of the earlier rounds - their measurements
# live on in profiles/ and DESIGN_HISTORY.md):
#     gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <session-name> <step> [<step> ...]'
# Every step writes under gpurun_out/<session-name>/ and prints a short tail; steps are independent.  A step may carry arguments after a colon,
# e.g. bench:--workload=ba100k-all or pytest:tests/test_xl_route.py.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
NAME=${1:-session}; shift
O=$ROOT/gpurun_out/$NAME; mkdir -p "$O"
export TMPDIR=/tmp
stats_csv() { find "$1" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$2"; rm -rf "$1"; }
for step in "$@"; do
  s=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}; arg=${arg//,/ }
  echo "=== $step"
  case $s in
    build)        timeout 900 python __graft_entry__.py 2>&1 | tail -2 ;;
    smoke)        timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4 ;;
    pytest)       timeout 3000 python -m pytest ${arg:-tests} -m gpu -x -q 2>&1 | tail -15 | tee "$O/pytest_$(echo "$arg" | tr '/ :' '___').txt" ;;
    pytest_all)   timeout 5000 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee "$O/pytest_all.txt" ;;
    sample)       GNNX_WRITE_SAMPLE="$O/ba100k_all_sample.npy" timeout 1500 python bench.py --workload ba100k-all --steps 1 --warmup 1 --no-cpu-baseline > "$O/bench_sample.json" 2> "$O/bench_sample.err"; ls -la "$O" ;;
    ties)         GNNX_WRITE_TIES=1 timeout 3000 python -m pytest tests/test_decision_parity.py -m gpu -q 2>&1 | tail -8 | tee "$O/ties_pytest.txt"; cp tests/golden/*_ties.json "$O/" ;;
    probe_xl)     timeout 1500 python tools/probe_xl.py $arg --out "$O/probe_xl.json" 2>&1 | tail -12 ;;
    bench)        tag=$(echo "$arg" | tr -c 'a-zA-Z0-9' '_'); timeout 1500 python bench.py $arg > "$O/bench_$tag.json" 2> "$O/bench_$tag.err"; tail -c 1500 "$O/bench_$tag.json"; tail -3 "$O/bench_$tag.err" ;;
    rocprof)      tag=$(echo "$arg" | tr -c 'a-zA-Z0-9' '_'); (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$tag" -- python "$ROOT/bench.py" $arg > "$O/bench_under_rocprof_$tag.json" 2>/dev/null); stats_csv "$O/prof_$tag" "$O/kernel_stats_$tag.csv"; head -8 "$O/kernel_stats_$tag.csv" | cut -c1-200 ;;
    rocprofpy)    tag=$(echo "$arg" | tr -c 'a-zA-Z0-9' '_'); (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$tag" -- python $ROOT/$arg > "$O/rocprofpy_$tag.log" 2>&1); stats_csv "$O/prof_$tag" "$O/kernel_stats_$tag.csv"; head -12 "$O/kernel_stats_$tag.csv" | cut -c1-70,200-330 ;;
    pmc)          # three counter passes of the loop of a workload (arg = bench.py flags), summarised per kernel: profiles/rNN_pmc_summary_<workload>_loop_only.json
                  tag=$(echo "$arg" | tr -c 'a-zA-Z0-9' '_'); B="python $ROOT/bench.py $arg --steps 3 --warmup 1 --no-cpu-baseline --no-parity-gate --loop-only"
                  (cd /tmp && timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/pmc_a_$tag" -- $B > /dev/null 2> "$O/pmc_a_$tag.err"
                   timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_b_$tag" -- $B > /dev/null 2> "$O/pmc_b_$tag.err"
                   timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_c_$tag" -- $B > /dev/null 2> "$O/pmc_c_$tag.err"
                   timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_d_$tag" -- $B > /dev/null 2> "$O/pmc_d_$tag.err")
                  python tools/pmc_summary.py "$O/pmc_summary_$tag.json" "$O/pmc_per_kernel_$tag.csv" "$O/pmc_a_$tag" "$O/pmc_b_$tag" "$O/pmc_c_$tag" "$O/pmc_d_$tag" | cut -c1-400
                  rm -rf "$O/pmc_a_$tag" "$O/pmc_b_$tag" "$O/pmc_c_$tag" "$O/pmc_d_$tag" "$O"/pmc_?_$tag.err ;;
    dist)         # the sharded bench with N ranks SHARING this box's one GPU (gloo instead of RCCL for the gather): a functional run of the --gpus N path, not a scaling number.  arg = N[,bench flags]
                  n=${arg%% *}; rest=""; [[ "$arg" == *" "* ]] && rest=${arg#* }
                  GNNX_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n $rest > "$O/bench_sharded_${n}ranks_one_gpu_gloo.json" 2> "$O/bench_sharded_${n}ranks.err"
                  tail -c 1800 "$O/bench_sharded_${n}ranks_one_gpu_gloo.json"; grep -v "amdgpu.ids" "$O/bench_sharded_${n}ranks.err" | tail -6 ;;
    py)           timeout 1500 python $arg 2>&1 | tail -30 ;;
    sh)           timeout 1500 bash $arg 2>&1 | tail -30 ;;
    *)            echo "unknown step $s" ;;
  esac
done
