#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (counter_collection CSVs) into per-kernel means per launch.

    python tools/pmc_summary.py OUT.json [OUT.csv] DIR [DIR ...]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE
counts half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) - the correction that guide prescribes."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def base(name):
    m = re.search(r"(k_\w+)", name)
    return m.group(1) if m else name.split("(")[0]


def main():
    args = sys.argv[1:]
    out_json = args.pop(0)
    out_csv = args.pop(0) if args and args[0].endswith(".csv") else None
    acc = defaultdict(lambda: defaultdict(list))       # counter -> kernel -> per-dispatch sums
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(float)
            for row in csv.DictReader(open(f)):
                per[(row["Counter_Name"], row["Kernel_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
            for (c, k, _), v in per.items():
                acc[c][k].append(v)
    rows, summary = [], {"counters_mean_per_launch": {}, "hbm_bytes_per_launch": {}, "launches": {}}
    for c in sorted(acc):
        for k in sorted(acc[c]):
            v = acc[c][k]
            rows.append((c, k, len(v), sum(v) / len(v)))
            summary["counters_mean_per_launch"].setdefault(base(k), {})[c] = sum(v) / len(v)
            summary["launches"][base(k)] = len(v)
    for k, cs in summary["counters_mean_per_launch"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            summary["hbm_bytes_per_launch"][k] = (2.0 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024.0
    # normalised utilisations: SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over the SIMDs that issue
    # MFMA, GRBM_GUI_ACTIVE the shader cycles of the launch; 256 CUs x 4 SIMDs can be busy at once.  SQ_WAIT_ANY / SQ_WAVE_CYCLES
    # are both in quad-cycles, so their ratio needs no unit.
    NUM_SIMDS = 256 * 4
    summary["normalised"] = {}
    for k, cs in summary["counters_mean_per_launch"].items():
        nrm = {}
        if cs.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            nrm["mfma_busy_frac_of_all_simds"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] * NUM_SIMDS)
        if cs.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in cs:
            nrm["waves_waiting_frac"] = cs["SQ_WAIT_ANY"] / cs["SQ_WAVE_CYCLES"]
        if cs.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_ANY" in cs:
            nrm["waves_issuing_frac"] = cs["SQ_ACTIVE_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
        if cs.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in cs:
            nrm["lds_bank_conflict_frac_of_lds_active"] = cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_LDS_IDX_ACTIVE"]
        if nrm:
            summary["normalised"][k] = nrm
    # launch durations of the counter passes themselves (their --kernel-trace): what the counters of a launch are divided by
    dur = defaultdict(list)
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                try:
                    dur[base(row["Kernel_Name"])].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6)
                except (KeyError, ValueError):
                    pass
    summary["avg_launch_ms"] = {k: sum(v) / len(v) for k, v in dur.items() if v}
    summary["note"] = ("rocprofv3 --pmc, separate passes per counter group; HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB "
                       "(gfx950: FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section)")
    json.dump(summary, open(out_json, "w"), indent=1)
    if out_csv:
        with open(out_csv, "w") as f:
            f.write("counter,kernel,launches,mean_counter_value_per_launch\n")
            for c, k, n, m in rows:
                f.write("%s,%s,%d,%r\n" % (c, k.replace(",", ";")[:110], n, m))
    print(json.dumps(summary["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
