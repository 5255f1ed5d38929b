#!/bin/bash
# round 4, session g: the edge-only host draw in the pipeline (16 384- and 2048-target BA-House x100k sets), its thread count; pipeline tests
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_pipeline.py tests/test_host_api.py -m gpu -q --timeout=600 > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
for th in 96 32 64 128; do GNNX_PIPE_EDGE_THREADS=$th timeout 900 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --reps 3 --no-parity-gate --no-cpu-baseline > $O/bench_ba100k_16384_edge_threads$th.json 2> $O/bench_ba100k_16384_edge_threads$th.err; python -c "
import json;d=json.loads(open('$O/bench_ba100k_16384_edge_threads$th.json').read().strip().splitlines()[-1]);e=d['end_to_end_stage_ms'];print('edge draw threads $th value',round(d['value']),'ms',round(d['ms_per_step'],1),'loop',round(d['loop_only']['ms_per_step'],1),'rng',round(e['host_rng_ms'],1),'prepare',round(e['prepare_ms'],1),'plan',round(e['plan_pack_route_layout_ms'],1),'khop',round(e['khop_ms'],1), e['repetitions']['values'])"; done
GNNX_PIPE_EDGE_DRAW=0 timeout 900 python bench.py --workload ba100k --targets 16384 --steps 3 --warmup 1 --reps 3 --no-parity-gate --no-cpu-baseline > $O/bench_ba100k_16384_full_draw.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_ba100k_16384_full_draw.json').read().strip().splitlines()[-1]);e=d['end_to_end_stage_ms'];print('full draw value',round(d['value']),'ms',round(d['ms_per_step'],1),'rng',round(e['host_rng_ms'],1), e['repetitions']['values'])"
timeout 600 python bench.py --workload ba100k --targets 2048 --steps 5 --warmup 2 --no-parity-gate --no-cpu-baseline > $O/bench_ba100k_2048.json 2> /dev/null; python -c "
import json;d=json.loads(open('$O/bench_ba100k_2048.json').read().strip().splitlines()[-1]);e=d['end_to_end_stage_ms'];print('2048 targets value',round(d['value']),'ms',round(d['ms_per_step'],1),'rng',round(e['host_rng_ms'],1), e['repetitions']['values'], 'edges only' , e.get('host_rng_edges_only'))"
