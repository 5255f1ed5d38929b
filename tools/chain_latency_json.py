#!/usr/bin/env python
"""tools/micro/chain_latency output (txt) -> the latency table bench.py's chain bound reads (profiles/r05_chain_latency.json).

    python tools/chain_latency_json.py profiles/r05_chain_latency.txt profiles/r05_chain_latency.json [profiles/r05_permlane_swap.txt]

unloaded = one wave alone (64 threads x 1 workgroup): what a dependent operation costs when nothing competes - the figures of the BOUND;
two_waves_per_simd = 512 threads x 256 workgroups (the occupancy of the 512-thread class and of the mixed launch on a full chip): the same
chain as the kernels meet it - reported next to the bound, not part of it.  A micro-benchmark step is the operation plus the one
dependent fma that keeps the chain alive where the operation alone has no data dependence (shuffle, barrier); "transc" steps are one sigmoid
= two transcendentals and an add, so one transcendental = (step - fma) / 2.  L2 hits are not measured here: 200 cycles (MI355X_MICROARCH.md)."""
import json
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = {}
    for line in open(src):
        m = re.match(r"(\w+)\s+(\d+) threads x\s+(\d+) workgroups:\s+([\d.]+) ns", line)
        if m:
            rows[(m.group(1), int(m.group(2)), int(m.group(3)))] = float(m.group(4))

    def table(threads, wgs):
        g = lambda k: rows[(k, threads, wgs)]
        fma = g("fma")
        return {"lds": g("lds_load"), "shuffle": g("shuffle"), "dpp": g("dpp"), "fma": fma, "mfma": g("mfma32x32x2"),
                "transc": max(0.0, (g("transc") - fma) / 2.0), "handover": g("store_sync_ld"), "barrier": max(0.0, g("barrier") - fma),
                "l2": 200.0 / 2.4, "lds_gather_pair": g("lds_gather2")}
    out = {"source": src, "tool": "tools/micro/chain_latency.hip", "unit": "ns per dependent operation",
           "unloaded_ns": table(64, 1), "two_waves_per_simd_ns": table(512, 256), "raw_ns_per_step": {"%s %dx%d" % k: v for k, v in sorted(rows.items())}}
    # Round 5: the kernels take v + shfl_xor(v, 32 / 16) through gfx950's lane swaps (gnnx_kernels.hpp: xor32_sum / xor16_sum), not ds_bpermute: the
    # BOUND must price a shuffle step at the faster form.  tools/micro/permlane_swap prints "sum + one multiply" per dependent step; the multiply
    # is one dependent VALU operation (the fma figure).
    if len(sys.argv) > 3:
        swap = [float(m.group(1)) for m in (re.search(r"via v_permlane\d+_swap\s+dependent step.*?:\s+([\d.]+) ns", l) for l in open(sys.argv[3])) if m]
        if swap:
            step = max(0.0, min(swap) - out["unloaded_ns"]["fma"])
            out["unloaded_ns"]["shuffle_ds_bpermute"] = out["unloaded_ns"]["shuffle"]
            out["unloaded_ns"]["shuffle"] = min(out["unloaded_ns"]["shuffle"], step)
            out["lane_swap_source"] = sys.argv[3]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["unloaded_ns"]))


if __name__ == "__main__":
    main()
