#!/bin/bash
# address-translation counters of k_mask for plans built one after the other in one process (its launch time differs between them)
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GNNX_SPARSE_RESIDENT=0
timeout 500 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p -- python $GRAFT_REPO_ROOT/tools/probe_conv.py 1024 1:0,1:1024,2:0 > $GRAFT_REPO_ROOT/$O/probe.txt 2>&1
cd $GRAFT_REPO_ROOT
grep WIDE $O/probe.txt | cut -c1-120
python - <<PY
import csv, glob, collections
f = glob.glob("$O/p/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_mask" in r["Kernel_Name"]]
per = collections.OrderedDict()
for r in rows:
    per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = per.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(per)
print(len(ids), "k_mask dispatches")
step = max(1, len(ids) // 24)
for k in ids[::step]:
    c = per[k]
    print(k, {n: round(v) for n, v in c.items()})
PY
rm -rf $O/p
