#!/bin/bash
# copy the PMC summary of this session's counter passes where bench.py looks for it (profiles/rNN_pmc_summary_<workload>_loop_only.json)
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/$1/pmc_summary_*syn1*.json | head -1)
c=$(ls gpurun_out/$1/pmc_per_kernel_*syn1*.csv | head -1)
cp "$f" profiles/r06_pmc_summary_syn1_loop_only.json && cp "$c" profiles/r06_pmc_per_kernel_syn1_loop_only.csv
cp profiles/r06_pmc_summary_syn1_loop_only.json profiles/r06_pmc_per_kernel_syn1_loop_only.csv gpurun_out/$1/
python - <<'PY'
import json
d = json.load(open("profiles/r06_pmc_summary_syn1_loop_only.json"))
k = "k_sparse_resident_mixed"
print("hbm bytes per launch", d["hbm_bytes_per_launch"].get(k), "launch ms", d["avg_launch_ms"].get(k), d["normalised"].get(k))
PY
