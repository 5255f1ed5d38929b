"""Executed-work model of the edge-formulation kernels (k_sparse_resident / _mixed, k_sparse_large) - bench.py's `roofline`.

SURVEY.md section 8(d) prices the reference's DENSE formulation (6 n^2 (D + 2H) flop and 28 n^2 bytes per target and iteration,
explain.py:665-678, models.py:70-71).  The kernels that run every benchmark configuration execute the EDGE formulation (csrc/gnnx_sparse.hpp:
only the mask entries on edges are live, rows beyond two hops of the target are pruned), so that figure says nothing about how far they are
from what a compute unit can do: it yields "fractions" of 0.42, 1.04 and 4.71 on three workloads.  This module counts what one
launch EXECUTES and prices it against three ceilings; the largest of the three lower bounds on the launch time, divided by the measured
time, is `frac`:

  flops : executed f32 flops (edge gathers 2 nnz K on the row sets that have them, the row-local .W / .W^T / Jacobian products, per-entry
          products, regularisers + optimiser per edge) against 256 flop / clk / CU (MFMA f32 = packed VALU f32 = 157.3 TF / 256 CUs) of the
          CUs the launch keeps BUSY;
  lds   : executed LDS bytes (entry records, gathered rows, operand reads, hand-over stores) against 128 B / clk / CU of the busy CUs;
  chain : the dependent operations on the critical path of ONE iteration of a target - LDS round trips whose address or data hangs on the
          previous step, cross-lane shuffles, DPP steps, dependent MFMAs / FMAs / transcendentals, wave-level hand-overs, workgroup barriers -
          each priced at its UNLOADED latency measured on the GPU box (tools/micro/chain_latency.hip -> profiles/r05_chain_latency.json),
          times the iterations; a launch lasts at least as long as its slowest target's chain, and a saturated launch at least
          (sum of all chains) / (CUs x workgroups that fit a CU).

The operation counts follow the phases of sparse_resident_body (gnnx_sparse.hpp) and are stated per phase below; they are deliberately
OPTIMISTIC (the narrowest slot widths, perfectly pipelined independent loads, no bank conflicts, no issue limit), so each figure is a lower
bound and frac <= 1 has a meaning: 1 - frac is what the kernel loses to instruction issue, latency above the unloaded figure (two waves per
SIMD), bank conflicts, barrier skew and phases that could overlap but do not.
"""
import json
import math
import os

import numpy as np

CLOCK_HZ = 2.4e9
CU_F32_FLOPS = 256 * CLOCK_HZ            # per CU: 157.3 TF / 256 CUs (MI355X_MICROARCH.md: f32 MFMA 256 flop/clk/CU; packed VALU FMA the same)
CU_LDS_BPS = 128 * CLOCK_HZ              # per CU: ds_read_b32 128 B/clk (MI355X_MICROARCH.md, LDS table)
NUM_CUS = 256

# ns per dependent operation, one wave alone on its SIMD.  Defaults from MI355X_MICROARCH.md (ds_read_b32 issue -> use ~50 cyc + address
# arithmetic; dependent-accumulator 32x32x2 f32 MFMA 64 cyc; dependent v_fma ~4-8 cyc); replaced by the measured table when
# profiles/r*_chain_latency.json exists (tools/micro/chain_latency.hip on the GPU box).
DEFAULT_LATENCY_NS = {"lds": 64 / 2.4, "shuffle": 64 / 2.4, "dpp": 8 / 2.4, "fma": 6 / 2.4, "mfma": 64 / 2.4, "transc": 24 / 2.4,
                      "handover": 128 / 2.4, "barrier": 64 / 2.4, "l2": 500 / 2.4, "source": "MI355X_MICROARCH.md (defaults)"}


def load_latency_table(root):
    import glob
    cand = sorted(glob.glob(os.path.join(root, "profiles", "r*_chain_latency.json")))
    if not cand:
        return dict(DEFAULT_LATENCY_NS)
    with open(cand[-1]) as f:
        t = json.load(f)
    out = dict(DEFAULT_LATENCY_NS)
    out.update(t.get("unloaded_ns", {}))
    out["source"] = os.path.relpath(cand[-1], root)
    if "two_waves_per_simd_ns" in t:
        out["loaded"] = t["two_waves_per_simd_ns"]
    return out


def target_structure(n, eoff, rc, rows, graph_mode=False):
    """Per target, from its upper-triangle edge list (engine.EdgeMasks: rc[eoff[k]:eoff[k+1]], local ids) and its target row:
    directed entries nnz, rows / entries of row set B (the target and its neighbours: layer 2 and its backward), of row set A (rows within two
    hops: layer 1 and its backward), entries of rows in A that point into B (the only ones with a dZ2 row to gather), degree of the target and
    the largest degree in B / A (the longest row slots).  Graph mode: every row is in both sets.  -> dict of float arrays [T]."""
    T = len(n)
    keys = ("nnz", "rowsA", "rowsB", "nnzA", "nnzB", "nBinA", "deg_t", "maxdegA", "maxdegB", "edges")
    out = {k: np.zeros(T, np.float64) for k in keys}
    for k in range(T):
        a, b = int(eoff[k]), int(eoff[k + 1])
        nk = int(n[k])
        r, c = rc[a:b, 0].astype(np.int64), rc[a:b, 1].astype(np.int64)
        deg = np.bincount(r, minlength=nk) + np.bincount(c, minlength=nk)
        out["nnz"][k] = 2.0 * (b - a)
        out["edges"][k] = float(b - a)
        if graph_mode or nk == 0:
            inA = inB = np.ones(nk, bool)
            out["deg_t"][k] = 0.0
        else:
            t = int(rows[k])
            inB = np.zeros(nk, bool)
            inB[t] = True
            inB[c[r == t]] = True
            inB[r[c == t]] = True
            inA = inB.copy()
            inA[c[inB[r]]] = True
            inA[r[inB[c]]] = True
            out["deg_t"][k] = float(deg[t])
        out["rowsA"][k], out["rowsB"][k] = float(inA.sum()), float(inB.sum())
        out["nnzA"][k], out["nnzB"][k] = float(deg[inA].sum()), float(deg[inB].sum())
        out["nBinA"][k] = float((inA[r] & inB[c]).sum() + (inA[c] & inB[r]).sum())
        out["maxdegA"][k] = float(deg[inA].max()) if inA.any() else 0.0
        out["maxdegB"][k] = float(deg[inB].max()) if inB.any() else 0.0
    return out


def executed_flops_per_iter(S, D, H, O, C, xc, graph_mode=False):
    """f32 flops one iteration EXECUTES per target (arrays [T]); fused multiply-add = 2, a transcendental = 1.  Phases of sparse_resident_body:
       layer 1  : gather 2 nnzA D (algebraic constant-feature form, xc = 2: the row sum alone, 2 nnzA) + row-local rowsA (2 D H + 3 H) (xc = 2: 5 H)
       layer 2  : gather 2 nnzB H + row-local rowsB (2 H H + 3 H)
       layer 3  : row t only (node mode): 2 deg_t H + 2 H O + 3 O;   graph mode: all rows, 2 nnz H + n (2 H O + 3 O)
       head     : logits + softmax + dE: 4 (2 H + O) C + 4 C
       backward : dY3 / dZ3 (4 O + 2 O H); dZ2 on rowsB (8 H + 2 H H); the dZ2 gather 2 nBinA H; dZ1 side on rowsA (4 H Jacobian + 2 H D,
                  xc = 2: 7 H); per-entry products 2 H nnzB (+ 2 D nnzA unless the features are constant); feature-mask reduction
       edges    : ~52 per undirected edge (two sigmoids, two Adam updates, regulariser gradients)
    """
    nnzA, nnzB, rowsA, rowsB = S["nnzA"], S["nnzB"], S["rowsA"], S["rowsB"]
    if graph_mode:
        f = 2 * S["nnz"] * (D + 2 * H) + rowsA * (2 * D * H + 2 * H * H + 2 * H * O + 3 * (2 * H + O))       # three full layers
        f += 4 * (2 * H + O) * C + 4 * C + 96 * rowsA                                                       # head + the three max-pools
        f += 2 * S["nnz"] * 2 * H + rowsA * (3 * (4 * H) + 2 * H * O + 2 * H * H + 2 * H * D)                # dX2, dX1 gathers + three row-local backwards
        f += S["nnz"] * (D + 2 * H) * 2                                                                     # per-edge products of the three layers
        return f + 52.0 * S["edges"]
    l1 = (2 * nnzA + rowsA * 5 * H) if xc == 2 else (2 * nnzA * D + rowsA * (2 * D * H + 3 * H))
    l2 = 2 * nnzB * H + rowsB * (2 * H * H + 3 * H)
    l3 = 2 * S["deg_t"] * H + 2 * H * O + 3 * O
    head = 4 * (2 * H + O) * C + 4 * C + 4 * O + 2 * O * H
    dz2 = rowsB * (8 * H + 2 * H * H)
    dz1 = 2 * S["nBinA"] * H + rowsA * (7 * H if xc == 2 else 4 * H + 2 * H * D + D)
    ent = 2 * H * nnzB + (0 if xc else 2 * D * nnzA)
    fm = (2 * H * D + rowsA * H) if xc == 2 else rowsA * D
    return l1 + l2 + l3 + head + dz2 + dz1 + ent + fm + 52.0 * S["edges"]


def executed_lds_bytes_per_iter(S, D, H, O, C, xc, graph_mode=False, large=False):
    """LDS bytes one iteration moves per target (arrays [T]): entry records (4 B masked adjacency + 2 B column), gathered rows, the MFMA /
    readlane operand reads of the row-local phases (64 lanes x K/2 x 4 B per wave and phase), hand-over stores, the edge phase (6 loads +
    3 stores of 4 B per undirected edge).  k_sparse_large gathers its rows from L2, not LDS (large = True): only the entry records count."""
    nnzA, nnzB, rowsA, rowsB = S["nnzA"], S["nnzB"], S["rowsA"], S["rowsB"]
    wavesA, wavesB = np.ceil(rowsA / 32.0), np.ceil(rowsB / 32.0)
    row = 0.0 if large else 1.0
    if graph_mode:
        b = S["nnz"] * (3 * 6 + 4 * row * (D + 2 * H)) + S["nnz"] * (2 * 6 + 4 * row * 2 * H)       # five full gathers
        b += wavesA * 64 * 4 * (D + 4 * H + O) / 2 * 2 + rowsA * 4 * (4 * H + 2 * O + 6)           # operand reads + stores of six row-local phases
        b += 96 * rowsA * 4 + S["edges"] * (2 * 4 * row * (D + 2 * H + H) + 36)                    # pools, per-edge products, edge phase
        return b
    l1 = (4 * nnzA if xc == 2 else nnzA * (6 + 4 * row * D)) + rowsA * (4 * H + 4 + 8 * H)
    l2 = nnzB * (6 + 4 * row * H) + wavesB * 64 * (H / 2) * 4 + rowsB * 4 * (H + 1)
    l3 = S["deg_t"] * (6 + 4 * H) + 64 * (2 * H) * 4 * 2 + 64 * 3 * C * 4
    dz2 = rowsB * 4 * (2 * H + 3) + wavesB * 64 * (H / 2) * 4
    dz1 = S["nBinA"] * (6 + 4 * row * H) + rowsA * 16 + nnzB * (2 + 4 * row * H) + rowsB * 4 * H + nnzA * 4
    if xc != 2:
        dz1 = dz1 + wavesA * 64 * (H / 2) * 4 + rowsA * 8 * D + (0 if xc else nnzA * (2 + 4 * row * D))
    fm = wavesA * 2 * H * 4 * 2 + 64 * 2 * H * 4 + 32 * 16
    return l1 + l2 + l3 + dz2 + dz1 + fm + 36.0 * S["edges"]


def chain_ops_per_iter(S, D, H, O, C, xc, graph_mode=False, large=False, slot=4):
    """Dependent operations on the critical path of ONE iteration of every target -> dict of arrays [T] (keys of the latency table).

    Node mode (wave 0 runs layer 2 -> row t + head -> dZ2 back to back; gnnx_sparse.hpp, `fuseB`), per phase, with g(x, u) = ceil(x / u) trips
    of a gather whose slot holds x entries and u entries per trip.  `slot` = the narrowest slot width the kernel may choose (4 / 8 entries);
    a row longer than 16 slots of that width doubles it (`slot_of`):
      layer 1    : gather - one (Abar, column) load + one row load for the first trip, one more round trip per further trip (the next pair is
                   loaded while the rows are in flight); the algebraic form needs no row: g(sA, 4) loads.  FMA chain over the slot's entries;
                   split rows: log2 DPP steps; norm: 1 shuffle + sqrt + rcp; store + workgroup barrier.
      layer 2    : gather 1 + g(sB, 2) loads, 2 FMAs per trip; 10 dependent MFMAs (K = H = 20 as ten 32x32x2 steps); norm; hand-over to wave 0's
                   next phase (store -> wave sync -> load).
      row t, head: gather of row t (index -> row: 2 loads per trip, g(deg_t / 2, 1) trips); 1 shuffle; y = z W3: H/2 dependent FMAs (two chains);
                   norm: 4 DPP + 1 shuffle, sqrt + division; logits: 2 FMAs + 4 DPP + 1 shuffle; softmax: exp + rcp; dE: C FMAs; dY3: 4 DPP +
                   1 shuffle + division; dZ3: H/2 FMAs; hand-over.
      dZ2        : 1 load group; 2 shuffles; 10 dependent MFMAs; H/2 + 5 FMAs; rcp; store + workgroup barrier.
      layer-1 backward: dZ2 gather (index -> row) 2 loads; split combine; Jacobian: 2 shuffles + rcp + ~H FMAs; one number per row handed to the
                   row's other slots (hand-over); rows of t and its neighbours: their entries' products, 1 + 2 g(sB, 4) loads and H/2 FMAs per
                   trip; 4 DPP; store + workgroup barrier.   General form: + 10 dependent MFMAs + one more hand-over.
      feature mask (wave 0): partial sums 1 load + waves adds; H FMAs through v_readlane; Adam (sqrt, rcp) + sigmoid (exp, rcp); wt: 1 load +
                   D FMAs; hand-over.
      edges      : 1 load group; ~12 FMAs; sqrt + rcp + exp + rcp; stores + workgroup barrier.
    k_sparse_large: the same phases with the row gathers served by L2 ("l2" instead of "lds" for the row loads).
    Graph mode: three full layers + pools + head + three row-local backwards + two gathers; no single-wave fusion (six more barriers)."""
    z = lambda: np.zeros_like(S["nnz"])
    ops = {k: z() for k in ("lds", "l2", "shuffle", "dpp", "fma", "mfma", "transc", "handover", "barrier")}
    rowkey = "l2" if large else "lds"

    def slot_of(maxdeg, width):            # entries of the longest slot: rows longer than 16 slots double the width
        w = np.full_like(maxdeg, float(width))
        for _ in range(3):
            w = np.where(np.ceil(maxdeg / w) > 16, 2 * w, w)
        return np.minimum(np.maximum(maxdeg, 1.0), w)
    sA, sB = slot_of(S["maxdegA"], 8 if not graph_mode else slot), slot_of(S["maxdegB"], slot)
    splitA = np.ceil(np.log2(np.maximum(1.0, np.ceil(S["maxdegA"] / np.maximum(sA, 1.0))))).clip(0, 4)
    splitB = np.ceil(np.log2(np.maximum(1.0, np.ceil(S["maxdegB"] / np.maximum(sB, 1.0))))).clip(0, 4)
    norm = dict(shuffle=1, transc=2)      # row norm: cross-half sum, sqrt, rcp (its four-deep FMA tree is in each phase's fma count)

    def add(rows=0.0, **kw):      # rows: gathered-row loads - LDS round trips, or L2 hits in k_sparse_large
        ops[rowkey] = ops[rowkey] + rows
        for k, v in kw.items():
            ops[k] = ops[k] + v
    if graph_mode:
        for K in (D, H, H):     # three forward layers on all rows
            add(lds=1, rows=np.ceil(sA / 2.0), fma=sA + 4, dpp=splitA, mfma=math.ceil(K / 2), barrier=1, **norm)
        add(lds=1, fma=2 * S["rowsA"], barrier=2)       # max-pool: a thread scans one column over the rows (independent loads, a dependent compare + select per row)
        add(lds=3, shuffle=1, dpp=9, fma=12 + 2 * C, transc=2, handover=2, barrier=1)    # softmax head, dE
        for K in (O, H, H):     # three row-local backwards, two of them behind a gather
            add(lds=1, shuffle=1, transc=1, fma=H / 2 + 5, mfma=math.ceil(K / 2), barrier=1)
        for _ in range(2):
            add(lds=1, rows=np.ceil(sA / 2.0), fma=sA, dpp=splitA)
        add(lds=3, fma=(D + 2 * H) / 2 + 12, transc=4, barrier=1)                         # edge phase: three products per edge
        return ops
    # layer 1
    if xc == 2:
        add(lds=np.ceil(sA / 4.0), fma=sA + H / 4 + 4, dpp=splitA, barrier=1, **norm)
    else:
        add(lds=1, rows=np.ceil(sA / 2.0), fma=sA + 4, dpp=splitA, mfma=math.ceil(D / 2), barrier=1, **norm)
    # layer 2 (wave 0)
    add(lds=1, rows=np.ceil(sB / 2.0), fma=2 * np.ceil(sB / 2.0) + 4, dpp=splitB, mfma=math.ceil(H / 2), handover=1, **norm)
    # row t of layer 3 + head
    trips_t = np.maximum(1.0, np.ceil(S["deg_t"] / 2.0))
    add(lds=trips_t, rows=trips_t)
    add(shuffle=1 + 1 + 1 + 1, dpp=4 + 4 + 4, fma=trips_t + H / 2 + 2 + C + 4 + H / 2, transc=2 + 2 + 2, handover=1)
    # dZ2
    add(lds=1, shuffle=2, mfma=math.ceil(H / 2), fma=H / 2 + 5, transc=1, barrier=1)
    # layer-1 backward
    tripsB = np.ceil(sB / 4.0)
    add(lds=1 + 1 + 1, rows=1 + tripsB, dpp=splitA + 4, shuffle=2, transc=1, fma=H / 2 + H / 2 + (H / 2) * tripsB, handover=1, barrier=1)
    if xc != 2:
        add(mfma=math.ceil(H / 2), handover=1)
    # feature mask + wt (wave 0, between the barrier and its edges)
    add(lds=2 + (1 if xc == 2 else 0), fma=8 + H + 8 + (D if xc == 2 else 0), transc=4, handover=1)
    # edge phase
    add(lds=1, fma=12, transc=4, barrier=1)
    return ops


def chain_ns_per_iter(ops, lat):
    return sum(ops[k] * lat[k] for k in ops)


def launch_bounds(flops, lds_bytes, chain_ns, iters, wg_of_target, wgs_per_cu=1, cus=NUM_CUS):
    """The three lower bounds (seconds) on ONE launch whose targets k run in workgroups wg_of_target[k] (targets of one workgroup run
    concurrently: the mixed launch's eight single-tile targets).  A workgroup lives on one CU; at most cus x wgs_per_cu are resident."""
    wg = np.asarray(wg_of_target)
    nwg = int(wg.max()) + 1 if len(wg) else 0
    busy = max(1, min(nwg, cus))
    f_wg = np.bincount(wg, weights=flops, minlength=nwg)
    b_wg = np.bincount(wg, weights=lds_bytes, minlength=nwg)
    c_wg = np.zeros(nwg)
    np.maximum.at(c_wg, wg, chain_ns)
    slots = cus * max(1, wgs_per_cu)
    t_flop = iters * max(f_wg.max(initial=0.0) / CU_F32_FLOPS, f_wg.sum() / (busy * CU_F32_FLOPS))
    t_lds = iters * max(b_wg.max(initial=0.0) / CU_LDS_BPS, b_wg.sum() / (busy * CU_LDS_BPS))
    t_chain = iters * 1e-9 * max(c_wg.max(initial=0.0), c_wg.sum() / slots)
    return {"flops_s": t_flop, "lds_s": t_lds, "chain_s": t_chain, "workgroups": nwg, "busy_cus": busy,
            "executed_flops": float(f_wg.sum() * iters), "executed_lds_bytes": float(b_wg.sum() * iters),
            "chain_ns_per_iter_slowest": float(c_wg.max(initial=0.0)), "chain_ns_per_iter_mean": float(c_wg.mean()) if nwg else 0.0}
