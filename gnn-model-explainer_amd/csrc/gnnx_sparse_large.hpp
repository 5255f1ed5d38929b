// gnnx_sparse_large.hpp — the edge-sparse, one-workgroup-per-target optimisation for targets that are too large for
// k_sparse_resident's LDS (up to SPL_N_MAX = 16383 nodes; node mode).
//
// Same mathematics, row slots, hop pruning, lane mapping and helpers as k_sparse_resident (gnnx_sparse.hpp).  What the
// kernel exploits beyond that is how little of a large sub-graph the prediction loss can see:
//   * the loss reads row t of layer 3 only (explain.py:713), so layer 2 is needed on t and its neighbours (row set B),
//     layer 1 on the rows within two hops (row set A), and dL/dAbar is non-zero only on entries of rows in A.  On the
//     BA-House x100k target set a 5600-node sub-graph has ~150 rows in A holding ~6400 of its 29200 directed entries;
//   * LDS therefore keeps only the ACTIVE entries - the masked adjacency and the column ids of the rows in A, renumbered
//     compactly with row t's entries first - plus per-slot norms and the weights.  Nothing in LDS scales with n;
//   * an edge with BOTH endpoints beyond two hops ("far") never receives a prediction gradient: its two mask entries
//     follow a closed scalar recursion (size + entropy + Laplacian terms through Adam) that depends on nothing else.  The
//     iteration loop skips them; each thread runs its far edges' whole trajectories in registers afterwards (no barrier,
//     no memory traffic).  Near edges are ordered first in the per-edge planes;
//   * the row arrays X, U1, U2 (= dZ2) are the caller's workspace arrays in HBM / L2 (stride 32 floats, indexed by the
//     original row id), written and re-read by the same workgroup through one CU's L1 / L2 path;
//   * the mask entries on edges, their Adam moments, the edge weights and the Laplacian constants are gathered once into
//     compact per-edge planes (coalesced) and scattered back into the dense M at the end;
//   * feature matrices with few distinct rows (constant, one-hot, categorical: every configuration of the reference) are
//     kept as a dictionary of at most 32 rows + one byte per node in LDS, found by exact comparison at setup: the two passes
//     that gather X rows (layer 1 and the per-entry products) then never leave the CU.  Other inputs take the L2 path;
//   * rows of up to 1024 entries are split into slots of 64 (the BA-House x100k hubs), loops run over rows / edges
//     instead of one item per thread.
#pragma once
#include <type_traits>
#include "gnnx_sparse.hpp"

namespace gnnx {

constexpr int SPL_THREADS = 512;            // 8 waves: two per SIMD leave 256 VGPRs per lane (the 1024-thread build spilled 120)
// entries in flight per lane in the row gathers: the rows come from L2 (~1 us per dependent round trip), so as many as the
// register file allows - NQ floats per entry per lane
__host__ __device__ constexpr int spl_gather_unroll(int nq) { return nq <= 5 ? 8 : nq <= 10 ? 4 : 2; }
constexpr int SPL_CHUNK = 64;               // entries per row slot
constexpr int SPL_N_MAX = 16383;            // rows of a sub-graph (row ids are packed in 15 / 16 bits)
constexpr int SPL_A_MAX = 8192;             // rows within two hops of the target
constexpr int SPL_TDEG_MAX = SP_MAX_SPLIT * SPL_CHUNK;   // entries of one row in A (1024), also of row t
constexpr int SPL_POOL_FLOATS = 39168;      // 153 KB of LDS
// The XL form keeps nothing in LDS that scales with the entries (gnnx_sparse_large.hpp below): weights, staging tiles, layer-3 partials, the feature
// dictionary and one byte per node - 96 KB hold sub-graphs of up to 62 000 nodes (the BA-House x100k maximum is 49 028; beyond, the feature rows
// come from L2).  What matters is the 60 KB it LEAVES: the prepare kernels of the next batches (sub-graph CSRs, the engine walk of the seeded
// masks: 8 KB of LDS per workgroup) run beside an XL workgroup instead of waiting milliseconds for a compute unit to drain ("room").
constexpr int SPL_POOL_FLOATS_XL = 24576;
constexpr int SPL_XD_MAX = 32;              // distinct feature rows kept as a dictionary in LDS (constant / one-hot / categorical features)
constexpr int SPL_COUNTS = 6;               // k_count_edges_large: nnz, slots of 64 (A), slots of 16 (A), active entries, rows in A, slots of 64 (B)
__host__ __device__ constexpr int spl_stage_floats(int D) { return 16 * TILE * (D | 1); }  // a [32][D|1] dZ1 tile per wave

// x = row | nsplit << 15 | wsplit << 20 | first << 25 | rem << 26 | inB << 31, y = e0 (compact) | len << 16,
// set A: m0 / m1 = bit k: entry e0 + k points at t or a neighbour of t;  set B: m0 = compact index of the entry (t, row), ~0 for t itself
struct SlotRec { unsigned x, y, m0, m1; };

__host__ __device__ inline int sparse_slots_of_c(int deg, int chunk) { return deg <= chunk ? 1 : (deg + chunk - 1) / chunk; }

// LDS, in floats.  Kept through the iterations: the weights, the active entries (+ one dummy entry that absorbs the
// mirror of an entry whose row is not in A), then the dZ1 staging tiles, the per-slot row norms and the layer-3 partials of
// t's neighbours.  The setup's temporaries overlay everything behind the active entries.
struct SparseLargeLayout {
    int oW, oWp, oAb, oCol, oStage, oRn1, oRn2, oG3, oXd, oXi, persist;
    int tLevel, tAidx, tAlist, tAdeg, tArp, tCbase, tSlot, total;
};
__host__ __device__ inline SparseLargeLayout sparse_large_layout(int ld, int nact, int nA, int padA, int padB, int D, int H, int C) {
    SparseLargeLayout L;
    int o = 0;
    L.oW = o;      o += (D + 2 * H) * 33;
    L.oWp = o;     o += C * 96;
    L.oAb = o;     o += nact + 1;
    L.oCol = o;    o += (nact + 2) / 2;
    L.oStage = o;
    int q = o;
    q += spl_stage_floats(D);
    L.oRn1 = q;    q += padA;
    L.oRn2 = q;    q += padB;
    L.oG3 = q;     q += SPL_TDEG_MAX;
    L.oXd = q;     q += SPL_XD_MAX * (D | 1);  // the distinct feature rows (when there are few)
    L.oXi = q;     q += (ld + 3) / 4;          // uint8: which of them a row carries
    L.persist = q;
    L.tLevel = o;  o += (ld + 3) / 4;          // uint8 hop level per row
    L.tAidx = o;   o += (ld + 1) / 2;          // uint16 index into the A list per row (0xffff: not in A)
    L.tAlist = o;  o += (nA + 1) / 2;          // uint16 rows of A, ascending
    L.tAdeg = o;   o += (nA + 1) / 2;          // uint16 their degrees
    L.tArp = o;    o += nA;                    // int: their first entry in the full CSR
    L.tCbase = o;  o += nA + 1;                // int: their first compact entry
    L.tSlot = o;   o += 2 * (nA + 1 + (nA + 1) / 2 + (SPL_CHUNK + 4) / 2);   // per set: slot_start [nA + 1] (int), order [nA], bucket [CHUNK + 1] (uint16)
    L.total = o > q ? o : q;
    return L;
}
// counts: what k_count_edges_large found (SPL_COUNTS ints)
__host__ __device__ inline bool sparse_large_fits(int n, int ld, const int* counts, int D, int H, int C) {
    const int nnz = counts[0], slotsA = counts[1], nact = counts[3], nA = counts[4], slotsB = counts[5];
    if (nnz < 0 || slotsA < 0 || slotsB < 0 || nact < 0 || nA <= 0) return false;
    const int padA = (slotsA + 31) & ~31, padB = (slotsB + 31) & ~31, eup = nnz / 2;
    return n <= SPL_N_MAX && nA <= SPL_A_MAX && nact + 1 < 65536 && (nnz & 1) == 0 && 7 * eup <= 32 * ld && nact + 1 <= 32 * ld &&
           4 * (padA + padB) <= 32 * ld && C <= RES_CMAX && H >= 2 &&
           sparse_large_layout(ld, nact, nA, padA, padB, D, H, C).total <= SPL_POOL_FLOATS;
}

// exclusive prefix sum of a[0..len) in place by one wave (lane = tid & 63); returns the total
template <class T>
__device__ __forceinline__ int wave_exclusive_scan_array(T* a, int len, int lane) {
    const int per = (len + 63) / 64;
    const int lo = lane * per, hi = (lo + per < len) ? lo + per : len;
    int s = 0;
    for (int r = lo; r < hi; ++r) s += (int)a[r];
    const int incl = wave_scan_inclusive(s, lane);
    int run = incl - s;
    for (int r = lo; r < hi; ++r) {
        const int v = (int)a[r];
        a[r] = (T)run;
        run += v;
    }
    return __shfl(incl, 63);
}

// Row gather from a row array in global memory (stride FS): acc[q] += sum_e Abar_e * f(B[col_e][2 q + half]).
// What limits these gathers is the L1's line rate - every lane addresses another row, so a load instruction costs one
// cache-line access per lane whatever its width.  The slot's two lanes therefore split its ENTRIES (parity), load whole
// rows with 16-byte loads (3 accesses per 10-column row instead of 10) and swap the column sums they owe each other at
// the end.  UN entries in flight per lane.
template <bool RELU, int NQ, int UN, class CT>
__device__ __forceinline__ void sparse_gather_rows(const float* sAb, const CT* scol, const float* B, int W, int e0,
                                                   int e1, int half, float (&acc)[NQ]) {
    constexpr int NV = (2 * NQ + 3) / 4;
    float full[2 * NQ];
#pragma unroll
    for (int c = 0; c < 2 * NQ; ++c) full[c] = 0.0f;
#pragma unroll 1
    for (int e = e0 + half; e < e1; e += 2 * UN) {
        float a[UN];
        const float* br[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {   // unconditional reads through a clamped index (see sparse_gather_dict)
            const int idx = (e + 2 * j < e1) ? e + 2 * j : e1 - 1;
            const float av = sAb[idx];
            br[j] = B + (int)scol[idx] * FS;
            a[j] = (e + 2 * j < e1) ? av : 0.0f;
        }
        f32x4 v[UN][NV];
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[j][k] = *reinterpret_cast<const f32x4*>(br[j] + 4 * k);
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int c = 0; c < 2 * NQ; ++c) {
                float x = (W == 2 * NQ || c < W) ? v[j][c >> 2][c & 3] : 0.0f;
                if (RELU) x = relu_(x);
                full[c] = fmaf(a[j], x, full[c]);
            }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float mine = half ? full[2 * q + 1] : full[2 * q];
        const float owed = half ? full[2 * q] : full[2 * q + 1];   // the other half's column
        acc[q] += mine + __shfl_xor(owed, 32);
    }
}

// The same gather when the feature rows come from the LDS dictionary (xd [.][sS], xi = dictionary index per node).
template <int NQ, int UN, class CT>
__device__ __forceinline__ void sparse_gather_dict(const float* sAb, const CT* scol, const unsigned char* xi, const float* xd,
                                                   int sS, int W, int e0, int e1, int half, float (&acc)[NQ]) {
    float full[2 * NQ];
#pragma unroll
    for (int c = 0; c < 2 * NQ; ++c) full[c] = 0.0f;
#pragma unroll 1
    for (int e = e0 + half; e < e1; e += 2 * UN) {
        // every load of a trip is unconditional (a clamped index instead of a predicate) and issued before the first use: a
        // predicated load becomes an exec-mask region with its own wait, and the three dependent LDS reads per entry
        // (column -> dictionary index -> row) then serialise (measured: 8 us for a 64-entry slot instead of 1.5)
        int idx[UN], cl[UN], xr[UN];
        float a[UN], rv[UN][2 * NQ];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            idx[j] = (e + 2 * j < e1) ? e + 2 * j : e1 - 1;
            cl[j] = scol[idx[j]];
            a[j] = sAb[idx[j]];
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) xr[j] = (int)xi[cl[j]] * sS;
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int c = 0; c < 2 * NQ; ++c) rv[j][c] = xd[xr[j] + c];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const float aj = (e + 2 * j < e1) ? a[j] : 0.0f;
#pragma unroll
            for (int c = 0; c < 2 * NQ; ++c) full[c] = fmaf(aj, (W == 2 * NQ || c < W) ? rv[j][c] : 0.0f, full[c]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float mine = half ? full[2 * q + 1] : full[2 * q];
        const float owed = half ? full[2 * q] : full[2 * q + 1];
        acc[q] += mine + __shfl_xor(owed, 32);
    }
}

// ... and when the dictionary holds ONE row (every feature row of the sub-graph equals it bit for bit: the reference's synthetic datasets use constant
// features, gengraph.py:60-61): neither the column of an entry nor its dictionary index nor the row are read - the lane keeps the row's columns in
// registers.  The same products in the same order as sparse_gather_dict with every index 0, so the results are bit-identical (round 6).
template <int NQ, int UN>
__device__ __forceinline__ void sparse_gather_onerow(const float* sAb, const float (&x0)[2 * NQ], int W, int e0, int e1, int half, float (&acc)[NQ]) {
    float full[2 * NQ];
#pragma unroll
    for (int c = 0; c < 2 * NQ; ++c) full[c] = 0.0f;
#pragma unroll 1
    for (int e = e0 + half; e < e1; e += 2 * UN) {
        float a[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) a[j] = sAb[(e + 2 * j < e1) ? e + 2 * j : e1 - 1];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const float aj = (e + 2 * j < e1) ? a[j] : 0.0f;
#pragma unroll
            for (int c = 0; c < 2 * NQ; ++c) full[c] = fmaf(aj, (W == 2 * NQ || c < W) ? x0[c] : 0.0f, full[c]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const float mine = half ? full[2 * q + 1] : full[2 * q];
        const float owed = half ? full[2 * q] : full[2 * q + 1];
        acc[q] += mine + __shfl_xor(owed, 32);
    }
}

// first position in [lo, hi) of a sorted id array whose value is >= key
template <class CT>
__device__ __forceinline__ int lower_bound_ids(const CT* a, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// k_sparse_xl = k_sparse_large<.., XL = true> (round 6): the SAME iteration loop - one source, one arithmetic, results bit-identical to the
// LDS form on every target both take - for targets of ANY size, CSR-native from end to end.  What it lifts, and how:
//   * no dense n x n block anywhere.  k_sparse_large reads the weights and the initial mask from the packed dense A / M blocks and writes dense
//     Abar / M blocks back (28 n^2 bytes of packing traffic per target; 9.6 GB per array at n = 49 028).  The XL form takes the target's CSR
//     (built from the FULL graph's CSR and the k-hop list by k_xl_rowdeg / k_xl_emit, gnnx_xl.hpp), the weights per directed entry, and the mask /
//     Adam moments as EDGE LISTS in the order of gnnx_gather_edges (upper-triangle edges, row-major: XlIo), and writes edge lists back;
//   * no 16-bit ids: rows, columns, compact entries and slots are 32-bit (n > 16 383, > 65 535 entries within two hops);
//   * the active entries (masked adjacency + column per directed entry of the rows within two hops: 23 k ... 218 k entries on the BA-House x100k
//     targets beyond 16 383 nodes, 1.4 per sub-graph node) live in the target's global scratch block - L2-resident, read by coalesced-per-slot
//     loads - instead of LDS; so do the per-slot norms, the slot records and the setup's temporaries (hop levels, row lists: O(n));
//     LDS keeps the weights, the staging tiles, the layer-3 partials and the feature dictionary (+ one byte per node while that fits);
//   * nothing is sized by the plan: the kernel derives the counts k_count_edges_large hands the LDS form (rows / entries within two hops, row
//     slots) itself during setup, in a scratch block laid out from upper bounds that depend on (n, nnz) alone (xl_layout).
// Limits that remain: a row within two hops of the target has at most SPL_TDEG_MAX = 1024 entries (16 slots of 64), C <= RES_CMAX.
// ------------------------------------------------------------------------------------------------------------------------------------------
struct XlIo {
    const float* w;             // weight of every directed entry, laid out like csr_col (offset csr_off[2 t + 1]); null = all ones
    const long long* eoff;      // [T + 1] upper-triangle edges of every target (row-major order: the layout of gnnx_gather_edges)
    float* M_e;                 // [E][2] in: the mask entries (M[r][c], M[c][r]) to start from; out: after num_iters steps
    const float* m_in_e;        // [E][2] exp_avg / exp_avg_sq to start from (gnnx_run_resume), or null = zeros
    const float* v_in_e;
    float* m_out_e;             // [E][2] the moments after the run, or null
    float* v_out_e;
    float* abar_e;              // [E] out: masked adjacency of the LAST forward on every edge (explain.py:209-211)
    float* scr;                 // the targets' scratch blocks
    const long long* scr_off;   // [T] float offset of target t's block (xl_layout(n, ld, nnz).total floats)
    long long* clk;             // measurement hook, or null: [T][4] wall_clock64 ticks (100 MHz) at the workgroup's start, after the setup, after the
                                // iteration loop and at its end - the per-target cost model of the sharded job is calibrated on them (parallel.py)
};

template <bool XL> struct SplTypes { using id_t = unsigned short; static constexpr int NOTA = 0xffff; };
template <> struct SplTypes<true> { using id_t = int; static constexpr int NOTA = -1; };

// info = len | nsplit << 8 | wsplit << 13 | first << 18 | rem << 19 | inB << 24 (set A: m0 / m1 as in SlotRec; set B: m0 = compact index of (t, row), ~0 for t)
struct SlotRecXL { unsigned row, e0, info, m0, m1, pad0, pad1, pad2; };

// float offsets inside a target's scratch block; every array sized from bounds that hold for ANY hop structure: entries within two hops <= nnz,
// rows within two hops <= n, row slots <= n + nnz / 4 (one per row + one per started 64 entries + the alignment of split rows)
struct XlLayout {
    long long oAb, oCol, oGe, oRec, oRn1, oRn2, oEst, oLap, oEi, oU1, oU2, oZraw, oLevel, oAidx, oAlist, oAdeg, oArp, oCbase, oSlot, total;
    int pad_max;
};
__host__ __device__ inline XlLayout xl_layout(int n, int ld, int nnz) {
    XlLayout L;
    const long long eup = nnz / 2, nact = (long long)nnz + 1, nA = n;
    const long long pad = (((long long)n + nnz / 4 + 32) + 31) & ~31LL;
    L.pad_max = (int)pad;
    long long o = 0;
    auto take = [&](long long words) { const long long r = o; o += (words + 63) & ~63LL; return r; };   // 256-B aligned arrays
    L.oAb = take(nact + 1);
    L.oCol = take(nact + 1);
    L.oGe = take(nact + 1);
    L.oRec = take(2 * pad * 8);
    L.oRn1 = take(pad);
    L.oRn2 = take(pad);
    L.oEst = take(7 * eup);
    L.oLap = take(eup);
    L.oEi = take(5 * eup);          // per near / far edge: i, j, compact entries c_ij, c_ji, index in the caller's edge list
    L.oU1 = take((long long)ld * FS);
    L.oU2 = take((long long)ld * FS);
    L.oZraw = take((long long)ld * FS);
    L.oLevel = take((ld + 3) / 4);
    L.oAidx = take(ld);
    L.oAlist = take(nA);
    L.oAdeg = take(nA);
    L.oArp = take(nA);
    L.oCbase = take(nA + 1);
    L.oSlot = take(2 * (nA + 1 + nA + 1 + SPL_CHUNK + 4));
    L.total = o;
    return L;
}
// LDS of the XL form (floats): weights, head, staging tiles, layer-3 partials of t's neighbours, feature dictionary, one byte per node (oXi < 0:
// the sub-graph has too many nodes for that - the feature rows then come from L2)
struct XlLds { int oW, oWp, oStage, oG3, oXd, oXi, total; };
__host__ __device__ inline XlLds xl_lds(int ld, int D, int H, int C, int pool_floats) {
    XlLds L;
    int o = 0;
    L.oW = o;      o += (D + 2 * H) * 33;
    L.oWp = o;     o += C * 96;
    L.oStage = o;  o += spl_stage_floats(D);
    L.oG3 = o;     o += SPL_TDEG_MAX;
    L.oXd = o;     o += SPL_XD_MAX * (D | 1);
    const int xi = (ld + 3) / 4;
    L.oXi = (o + xi <= pool_floats) ? o : -1;
    if (L.oXi >= 0) o += xi;
    L.total = o;
    return L;
}

// csr_*: the targets' CSR structure, built once per plan by k_build_csr_large (scanning a dense block with one
// workgroup takes milliseconds - too much to repeat in every launch): rowptr at csr_off[2 t], the ascending columns and
// the row of every directed entry at csr_off[2 t + 1]; counts: k_count_edges_large's output
// LOG (round 5): the logging form, as in k_sparse_resident - per iteration the loss scalars of explain.py:808-819 (prediction; size, entropy
// and Laplacian sums over the NEAR edges, reduced in a fixed order; the far edges add theirs from their closed recursions with float atomics,
// the entries off the edges come from k_dead_entries in front of the launch; feature-size term) and the decision trace (the ReLU gates of
// layer 1 on the rows within two hops and of layer 2 on t and its neighbours, read back from the row arrays).  A separate instantiation.
template <int DQ, int HQ, bool LOG = false, bool EX = true, bool XL = false>
__global__ __launch_bounds__(SPL_THREADS) void k_sparse_large(Params p, const int32_t* targets, const float* __restrict__ adam_tab,
                                                              const int32_t* __restrict__ csr_rowptr,
                                                              const void* __restrict__ csr_col_v,     // uint16 ids (XL: int32)
                                                              const void* __restrict__ csr_row_v,
                                                              const long long* __restrict__ csr_off, const int32_t* __restrict__ counts, const XlIo xl) {
    constexpr int NT = SPL_THREADS, NW = NT / 64;
    using id_t = typename SplTypes<XL>::id_t;       // rows / columns of the sub-graph, indices into the list of rows within two hops
    constexpr int NOTA = SplTypes<XL>::NOTA;        // "not within two hops"
    // (XL: the small pool for the reference's widths; the 32-wide instantiation's staging tiles alone are 66 KB)
    constexpr int POOL = (XL && DQ != 16) ? SPL_POOL_FLOATS_XL : SPL_POOL_FLOATS;
    __shared__ float pool[POOL];
    __shared__ SparseFixed sh;
    __shared__ int s_wn[NW], s_wf[NW], s_misc[4];
    const int t = targets[blockIdx.x];
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, ld = tm.ld, tr = tm.t;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    constexpr bool EXACT = EX && (DQ != 16);   // <5, 10>: exactly D = 10, H = O = 20 (compile-time widths); EX = false: widths up to those at run time; other shapes take <16, 16>
    const int D = EXACT ? 2 * DQ : p.D, H = EXACT ? 2 * HQ : p.H, O = EXACT ? 2 * HQ : p.O, C = p.C;
    const float* Ag = XL ? nullptr : p.A + tm.offQ;      // (the XL form has no dense blocks: XlIo)
    float* Mg = XL ? nullptr : p.M + tm.offQ;
    const int32_t* grp = csr_rowptr + csr_off[2 * t];
    const id_t* gcol = static_cast<const id_t*>(csr_col_v) + csr_off[2 * t + 1];
    const id_t* grow = static_cast<const id_t*>(csr_row_v) + csr_off[2 * t + 1];
    int nnz, slotsA, nact, nA, slotsB;     // LDS form: k_count_edges_large's figures (verified below); XL form: derived by the setup
    if constexpr (XL) {
        nnz = grp[ld];
        slotsA = slotsB = nact = nA = 0;
    } else {
        const int32_t* cn = counts + (size_t)SPL_COUNTS * t;
        nnz = cn[0], slotsA = cn[1], nact = cn[3], nA = cn[4], slotsB = cn[5];
    }
    const int eup = nnz >> 1;
    int padA = (slotsA + 31) & ~31, padB = (slotsB + 31) & ~31;
    float* xscr = XL ? xl.scr + xl.scr_off[t] : nullptr;
    const XlLayout XG = XL ? xl_layout(n, ld, nnz) : XlLayout{};
    // row arrays (stride FS; dZ2 overwrites U2 row by row as in the resident kernel): the caller's workspace / XL: the target's scratch block
    const float* gX = p.X + tm.offR * FS;
    float* gU1 = XL ? xscr + XG.oU1 : p.U[0] + tm.offR * FS;
    float* gU2 = XL ? xscr + XG.oU2 : p.U[1] + tm.offR * FS;
    float* gZraw = XL ? xscr + XG.oZraw : p.Zraw + tm.offR * FS;
    float* gGe = XL ? xscr + XG.oGe : p.dZT[0] + tm.offR * FS;   // dL/dAbar per active entry (row-side products), [nact + 1]: the dummy stays zero
    float* glap = XL ? xscr + XG.oLap : p.dZT[1] + tm.offR * FS;  // per edge: c_lap / 2 (yhat_i - yhat_j)^2 / n^2 (explain.py:793-811 on a 0/1 label vector pair)
    float* est = XL ? xscr + XG.oEst : p.UT[2] + tm.offR * FS;    // per-edge planes [7][eup]: M_ij, M_ji, m_ij, m_ji, v_ij, v_ji, weight
    // LDS form: eidx [eup][2] = i | j << 16, c_ij | c_ji << 16 (compact entries; near edges); XL form: five int planes [eup]: i, j, c_ij, c_ji, q
    // (q = the edge's index in the caller's edge lists)
    unsigned* eidx = reinterpret_cast<unsigned*>(XL ? xscr + XG.oEi : p.UT[0] + tm.offR * FS);
    SlotRec* srec = reinterpret_cast<SlotRec*>(XL ? nullptr : p.UT[1] + tm.offR * FS);
    SlotRecXL* srecx = reinterpret_cast<SlotRecXL*>(XL ? xscr + XG.oRec : nullptr);
    auto edge_entries = [&](int k, int& cij, int& cji) {      // the two compact entries of near edge k
        if constexpr (XL) {
            cij = (int)eidx[2 * (size_t)eup + k];
            cji = (int)eidx[3 * (size_t)eup + k];
        } else {
            const unsigned en = eidx[2 * k + 1];
            cij = en & 0xffffu;
            cji = en >> 16;
        }
    };
    auto edge_nodes = [&](int k, int& i, int& j) {
        if constexpr (XL) {
            i = (int)eidx[k];
            j = (int)eidx[(size_t)eup + k];
        } else {
            const unsigned nd = eidx[2 * k];
            i = nd & 0xffffu;
            j = nd >> 16;
        }
    };

    auto fail_nan = [&]() {
        const float qnan = __builtin_nanf("");
        if constexpr (XL) {
            for (long long e = xl.eoff[t] + tid; e < xl.eoff[t + 1]; e += NT) xl.abar_e[e] = qnan;
        } else {
            for (size_t e = tid; e < (size_t)ld * ld; e += NT) p.Abar[tm.offQ + e] = qnan;
        }
        if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = qnan;
    };
    if constexpr (XL) {
        if ((nnz & 1) || nnz < 0 || C > RES_CMAX || H < 2 || (long long)(xl.eoff[t + 1] - xl.eoff[t]) != (long long)eup) {  // uniform
            fail_nan();
            return;
        }
    } else {
        int cc[SPL_COUNTS];
        const int32_t* cn = counts + (size_t)SPL_COUNTS * t;
#pragma unroll
        for (int k = 0; k < SPL_COUNTS; ++k) cc[k] = cn[k];
        if (!sparse_large_fits(n, ld, cc, D, H, C) || grp[ld] != nnz) {  // uniform
            fail_nan();
            return;
        }
    }
    if constexpr (XL)
        if (xl.clk && tid == 0) xl.clk[4 * t] = wall_clock64();
    const int rt0 = grp[tr], degT = grp[tr + 1] - rt0;
    if (degT > SPL_TDEG_MAX) {
        fail_nan();
        return;
    }
    const SparseLargeLayout L = XL ? SparseLargeLayout{} : sparse_large_layout(ld, nact, nA, padA, padB, D, H, C);
    const XlLds XS = XL ? xl_lds(ld, D, H, C, POOL) : XlLds{};
    if (XL && XS.total > POOL) {      // uniform (cannot happen for D, H <= 2 DQ, 2 HQ)
        fail_nan();
        return;
    }
    float* sW1 = pool + (XL ? XS.oW : L.oW);
    float* sW2 = sW1 + D * 33;
    float* sW3 = sW2 + H * 33;
    float* sWp = pool + (XL ? XS.oWp : L.oWp);
    float* sAb = XL ? xscr + XG.oAb : pool + L.oAb;           // XL: the active entries live in the scratch block (L2)
    id_t* scol = reinterpret_cast<id_t*>(XL ? xscr + XG.oCol : pool + L.oCol);
    const int sS = D | 1;
    float* stage = pool + (XL ? XS.oStage : L.oStage) + wave * (TILE * sS);
    float* sRn1 = XL ? xscr + XG.oRn1 : pool + L.oRn1;
    float* sRn2 = XL ? xscr + XG.oRn2 : pool + L.oRn2;
    float* sG3 = pool + (XL ? XS.oG3 : L.oG3);
    float* sXd = pool + (XL ? XS.oXd : L.oXd);
    const bool have_xi = !XL || XS.oXi >= 0;                  // XL: one byte per node fits LDS (else the feature rows come from L2)
    unsigned char* sXi = reinterpret_cast<unsigned char*>(pool + (XL ? (XS.oXi >= 0 ? XS.oXi : 0) : L.oXi));
    // setup temporaries
    unsigned char* level = reinterpret_cast<unsigned char*>(XL ? xscr + XG.oLevel : pool + L.tLevel);
    id_t* aidx = reinterpret_cast<id_t*>(XL ? xscr + XG.oAidx : pool + L.tAidx);
    id_t* alist = reinterpret_cast<id_t*>(XL ? xscr + XG.oAlist : pool + L.tAlist);
    id_t* adeg = reinterpret_cast<id_t*>(XL ? xscr + XG.oAdeg : pool + L.tAdeg);
    int* arp = reinterpret_cast<int*>(XL ? xscr + XG.oArp : pool + L.tArp);
    int* cbase = reinterpret_cast<int*>(XL ? xscr + XG.oCbase : pool + L.tCbase);

    // ---------------- setup 1: hop levels 0..3 from the plan's CSR ----------------
    for (int r = tid; r < ld; r += NT) {
        level[r] = (r == tr) ? 0 : 3;
        aidx[r] = (id_t)NOTA;
    }
    if (tid == 0) sh.bad = 0;
    __syncthreads();
    for (int e = tid; e < degT; e += NT) level[gcol[rt0 + e]] = 1;
    __syncthreads();
    for (int e = wave; e < degT; e += NW) {  // one wave per neighbour of t
        const int r = gcol[rt0 + e];
        const int ra = grp[r], rb = grp[r + 1];
        for (int e2 = ra + lane; e2 < rb; e2 += 64) {
            const int c = gcol[e2];
            if (level[c] == 3) level[c] = 2;  // benign race
        }
    }
    __syncthreads();
    // ---------------- setup 2: the rows of A (ascending), their degrees and compact entry ranges (row t first) ----------------
    int cntA = 0;
    for (int r0 = 0; r0 < ld; r0 += NT) {
        const int r = r0 + tid;
        const bool inA = r < n && level[r] <= 2;
        const unsigned long long b = __ballot(inA);
        if (lane == 0) s_wn[wave] = __popcll(b);
        __syncthreads();
        int base = cntA, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = s_wn[w];
            base += (w < wave) ? c : 0;
            tot += c;
        }
        if (inA) {
            const int k = base + __popcll(b & ((1ull << lane) - 1ull));
            if (XL || k < nA) {      // (XL: k < n by construction - the list is sized for n rows)
                alist[k] = (id_t)r;
                aidx[r] = (id_t)k;
            }
        }
        cntA += tot;
        __syncthreads();
    }
    if constexpr (XL) nA = cntA;      // (the XL form derives what the LDS form verifies)
    if (cntA != nA || nA <= 0) {  // uniform: the plan's counts do not describe this adjacency
        fail_nan();
        return;
    }
    for (int k = tid; k < nA; k += NT) {
        const int r = alist[k];
        const int ra = grp[r], rb = grp[r + 1];
        arp[k] = ra;
        adeg[k] = (id_t)(rb - ra);
        cbase[k] = (r == tr) ? 0 : rb - ra;
        if (rb - ra > SPL_TDEG_MAX) sh.bad = 1;
    }
    __syncthreads();
    if (wave == 0) {
        const int total = wave_exclusive_scan_array(cbase, nA, lane);
        if (lane == 0) s_misc[0] = total + degT;
    }
    __syncthreads();
    for (int k = tid; k < nA; k += NT) cbase[k] = ((int)alist[k] == tr) ? 0 : cbase[k] + degT;
    if constexpr (XL) nact = s_misc[0];
    if (s_misc[0] != nact || sh.bad) {  // uniform
        __syncthreads();
        fail_nan();
        return;
    }
    __syncthreads();
    // column ids of the active entries (one pass over the directed entries, coalesced)
    for (int e = tid; e < nnz; e += NT) {
        const int ai = (int)aidx[grow[e]];
        if (ai != NOTA) scol[cbase[ai] + (e - arp[ai])] = gcol[e];
    }
    __syncthreads();
    // ---------------- setup 3: row slots of the two row sets ----------------
    constexpr int SETSZ_U16 = SPL_CHUNK + 4;
    // ints per set: slot_start [nA + 1], then order [nA], bucket [CHUNK + 1] (uint16; XL: int)
    const int set_words = XL ? (nA + 1) + (nA + 1) + SETSZ_U16 : nA + 1 + (nA + 1) / 2 + SETSZ_U16 / 2;
    int* const slot_tables = reinterpret_cast<int*>(XL ? xscr + XG.oSlot : pool + L.tSlot);
    const int order_stride = XL ? nA + 1 : ((nA + 1) & ~1);      // ids from `order` to `bucket`
    if ((tid & 31) == 0 && (tid >> 5) < 2) {  // one thread per row set: A (level <= 2), B (level <= 1)
        const int set = tid >> 5;
        const int lvlmax = 2 - set;
        int* slot_start = slot_tables + set * set_words;
        id_t* order = reinterpret_cast<id_t*>(slot_start + nA + 1);
        id_t* bucket = order + order_stride;
        int pos = 0, pcount = 0, cnt = 0;
        for (int d = 0; d <= SPL_CHUNK; ++d) bucket[d] = 0;
        for (int k = 0; k < nA; ++k) {
            if (level[alist[k]] > lvlmax) continue;
            ++cnt;
            const int d = adeg[k];
            if (d > SPL_CHUNK) {
                const int ns = sparse_slots_of_c(d, SPL_CHUNK);
                pos = sparse_place(pos, ns);
                order[pcount] = (id_t)k;
                slot_start[pcount] = pos;
                pos += ns;
                ++pcount;
            } else {
                bucket[d]++;
            }
        }
        int run = pcount;
        for (int d = SPL_CHUNK; d >= 0; --d) {
            const int c = bucket[d];
            bucket[d] = (id_t)run;
            run += c;
        }
        for (int k = 0; k < nA; ++k) {
            if (level[alist[k]] > lvlmax) continue;
            const int d = adeg[k];
            if (d <= SPL_CHUNK) order[bucket[d]++] = (id_t)k;
        }
        for (int q = pcount; q < cnt; ++q) slot_start[q] = pos + (q - pcount);
        slot_start[cnt] = pos + (cnt - pcount);
        sh.set_rows[set] = cnt;
        sh.set_slots[set] = pos + (cnt - pcount);
        if (!XL && pos + (cnt - pcount) != (set ? slotsB : slotsA)) sh.bad = 1;
    }
    __syncthreads();
    if constexpr (XL) {
        slotsA = sh.set_slots[0];
        slotsB = sh.set_slots[1];
        padA = (slotsA + 31) & ~31;
        padB = (slotsB + 31) & ~31;
        if (padA > XG.pad_max || padB > XG.pad_max) {  // (cannot happen: xl_layout's bound) - uniform
            fail_nan();
            return;
        }
    }
    if (sh.bad) {
        fail_nan();
        return;
    }
    // slot records -> workspace (set A first, then set B, each padded to whole half-waves): the phases walk them 256 at a time
    for (int k = 0; k < 2; ++k) {
        const int* slot_start = slot_tables + k * set_words;
        const id_t* order = reinterpret_cast<const id_t*>(slot_start + nA + 1);
        const int cnt = sh.set_rows[k], nslots = k ? slotsB : slotsA, npad = k ? padB : padA, base = k ? padA : 0;
        for (int s0 = wave * TILE; s0 < npad; s0 += NW * TILE) {  // one wave per 32 slots (both half-waves compute the same)
            const int sl = s0 + li;
            int zrow = 0, ze0 = 0, zlen = 0, zns = 1, zrem = 1;
            bool zfirst = false, zinb = false;
            unsigned zm0 = 0u, zm1 = 0u;
            if (sl < nslots) {
                int lo = 0, hi = cnt;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (slot_start[mid] <= sl) lo = mid; else hi = mid;
                }
                const int ka = order[lo];
                const int row = alist[ka];
                const int ra = cbase[ka], rb = ra + (int)adeg[ka];
                const int ns = sparse_slots_of_c(rb - ra, SPL_CHUNK), kk = sl - slot_start[lo];
                if (kk < ns) {
                    zrow = row;
                    ze0 = ra + kk * SPL_CHUNK;
                    zlen = ((ze0 + SPL_CHUNK < rb) ? ze0 + SPL_CHUNK : rb) - ze0;
                    zns = ns;
                    zrem = ns - kk;
                    zfirst = (kk == 0);
                    zinb = level[row] <= 1;
                    if (k == 0) {
                        for (int k2 = 0; k2 < zlen; ++k2)
                            if (level[scol[ze0 + k2]] <= 1) (k2 < 32 ? zm0 : zm1) |= 1u << (k2 & 31);
                    } else {  // the entry (t, row): row t's compact entries are 0 .. degT - 1, ascending columns
                        const int pe = lower_bound_ids(scol, 0, degT, row);
                        zm0 = (row != tr && pe < degT && (int)scol[pe] == row) ? (unsigned)pe : 0xffffffffu;
                    }
                }
            }
            int wsplit = zfirst ? zns : 1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int other = __shfl_xor(wsplit, o);
                wsplit = other > wsplit ? other : wsplit;
            }
            if (h == 0) {
                if constexpr (XL) {
                    SlotRecXL rec;
                    rec.row = (unsigned)zrow;
                    rec.e0 = (unsigned)ze0;
                    rec.info = (unsigned)zlen | ((unsigned)zns << 8) | ((unsigned)wsplit << 13) | ((unsigned)zfirst << 18) | ((unsigned)zrem << 19) |
                               ((unsigned)zinb << 24);
                    rec.m0 = zm0;
                    rec.m1 = zm1;
                    rec.pad0 = rec.pad1 = rec.pad2 = 0u;
                    srecx[base + sl] = rec;
                } else {
                    SlotRec rec;
                    rec.x = (unsigned)zrow | ((unsigned)zns << 15) | ((unsigned)wsplit << 20) | ((unsigned)zfirst << 25) | ((unsigned)zrem << 26) |
                            ((unsigned)zinb << 31);
                    rec.y = (unsigned)ze0 | ((unsigned)zlen << 16);
                    rec.m0 = zm0;
                    rec.m1 = zm1;
                    srec[base + sl] = rec;
                }
            }
        }
    }
    // the slot record of a round as it lies in memory (slot_raw) and its decoding (slot_dec): the two long row loops of an iteration load the
    // NEXT round's record while they work on the current one (round 6: the record is the head of every round's chain of dependent loads)
    using SlotRaw = std::conditional_t<XL, SlotRecXL, SlotRec>;
    auto slot_raw = [&](int set, int round) -> SlotRaw {
        const int sl = round * (NT / 2) + wave * TILE + li;
        const int npad = set ? padB : padA;
        SlotRaw rec;
        if constexpr (XL) {
            rec.row = rec.e0 = rec.info = rec.m0 = rec.m1 = 0u;
            if (sl < npad) rec = srecx[(set ? padA : 0) + sl];
        } else {
            rec.x = rec.y = rec.m0 = rec.m1 = 0u;
            if (sl < npad) rec = srec[(set ? padA : 0) + sl];
        }
        return rec;
    };
    auto slot_dec = [&](const SlotRaw& rec, int set, int round) -> RowSlot {  // this lane's slot in the given round (256 slots per round)
        const int npad = set ? padB : padA;
        RowSlot z;
        if constexpr (XL) {
            z.row = (int)rec.row;
            z.nsplit = (rec.info >> 8) & 31u;
            z.wsplit = (rec.info >> 13) & 31u;
            z.first = (rec.info >> 18) & 1u;
            z.rem = (rec.info >> 19) & 31u;
            if (z.rem == 0) z.rem = 1;
            z.inB = (rec.info >> 24) & 1u;
            z.bmask = rec.m0;
            z.bmask_hi = rec.m1;
            z.e0 = (int)rec.e0;
            z.e1 = z.e0 + (int)(rec.info & 255u);
        } else {
            z.row = rec.x & 32767u;
            z.nsplit = (rec.x >> 15) & 31u;
            z.wsplit = (rec.x >> 20) & 31u;
            z.first = (rec.x >> 25) & 1u;
            z.rem = (rec.x >> 26) & 31u;
            if (z.rem == 0) z.rem = 1;
            z.inB = (rec.x >> 31) & 1u;
            z.bmask = rec.m0;
            z.bmask_hi = rec.m1;
            z.e0 = rec.y & 0xffffu;
            z.e1 = z.e0 + (int)(rec.y >> 16);
        }
        z.wave_active = round * (NT / 2) + wave * TILE < npad;
        if (z.nsplit == 0) z.nsplit = 1;
        if (z.wsplit == 0) z.wsplit = 1;
        return z;
    };
    auto slot_of = [&](int set, int round) -> RowSlot { return slot_dec(slot_raw(set, round), set, round); };
    const int roundsA = (padA + NT / 2 - 1) / (NT / 2), roundsB = (padB + NT / 2 - 1) / (NT / 2);
    const float inv_n2 = 1.0f / ((float)n * (float)n);
    // ---------------- setup 4: the undirected edges, near ones (an endpoint within two hops) first; per-edge planes ----------------
    int eupN = 0, eupF = 0;
    {
        bool asym = false;
        for (int e0 = 0; e0 < nnz; e0 += NT) {
            const int e = e0 + tid;
            int i = 0, j = 0;
            bool up = false;
            if (e < nnz) {
                i = grow[e];
                j = gcol[e];
                up = j > i;
            }
            const int ai = up ? (int)aidx[i] : NOTA, aj = up ? (int)aidx[j] : NOTA;
            const bool near = up && (ai != NOTA || aj != NOTA);
            const bool far = up && !near;
            const unsigned long long bn = __ballot(near), bf = __ballot(far);
            if (lane == 0) {
                s_wn[wave] = __popcll(bn);
                s_wf[wave] = __popcll(bf);
            }
            __syncthreads();
            int nb = eupN, fb = eupF, nt = 0, ft = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int a = s_wn[w], b = s_wf[w];
                nb += (w < wave) ? a : 0;
                fb += (w < wave) ? b : 0;
                nt += a;
                ft += b;
            }
            if (up) {
                const unsigned long long lower = (1ull << lane) - 1ull;
                const int nbv = nb + __popcll(bn & lower), fbv = fb + __popcll(bf & lower);
                const int k = near ? nbv : eup - 1 - fbv;
                // (XL: the upper entries before this one in CSR order = its index in the caller's row-major edge lists)
                const long long q = XL ? xl.eoff[t] + (nbv + fbv) : 0;
                float w;
                if constexpr (XL) {
                    w = xl.w ? xl.w[csr_off[2 * t + 1] + e] : 1.0f;      // (symmetry of the weights: checked when the graph is uploaded)
                } else {
                    w = Ag[(size_t)i * ld + j];
                    if (Ag[(size_t)j * ld + i] != w) asym = true;
                }
                if (k >= 0 && k < eup) {
                    int cij = nact, cji = nact;
                    if (ai != NOTA) cij = cbase[ai] + (e - arp[ai]);
                    if (aj != NOTA) {
                        const int lo = cbase[aj], hi = lo + (int)adeg[aj];
                        cji = lower_bound_ids(scol, lo, hi, i);
                        if (cji >= hi || (int)scol[cji] != i) {
                            asym = true;
                            cji = nact;
                        }
                    }
                    if constexpr (XL) {
                        eidx[k] = (unsigned)i;
                        eidx[(size_t)eup + k] = (unsigned)j;
                        eidx[2 * (size_t)eup + k] = (unsigned)cij;
                        eidx[3 * (size_t)eup + k] = (unsigned)cji;
                        eidx[4 * (size_t)eup + k] = (unsigned)(nbv + fbv);
                        est[0 * eup + k] = xl.M_e[2 * q];
                        est[1 * eup + k] = xl.M_e[2 * q + 1];
                        est[2 * eup + k] = xl.m_in_e ? xl.m_in_e[2 * q] : 0.0f;   // gnnx_xl_run with a state to resume from
                        est[3 * eup + k] = xl.m_in_e ? xl.m_in_e[2 * q + 1] : 0.0f;
                        est[4 * eup + k] = xl.v_in_e ? xl.v_in_e[2 * q] : 0.0f;
                        est[5 * eup + k] = xl.v_in_e ? xl.v_in_e[2 * q + 1] : 0.0f;
                    } else {
                        eidx[2 * k] = (unsigned)i | ((unsigned)j << 16);
                        eidx[2 * k + 1] = (unsigned)cij | ((unsigned)cji << 16);
                        est[0 * eup + k] = Mg[(size_t)i * ld + j];
                        est[1 * eup + k] = Mg[(size_t)j * ld + i];
                        est[2 * eup + k] = p.m_in ? p.m_in[tm.offQ + (size_t)i * ld + j] : 0.0f;   // gnnx_run_resume: Adam moments
                        est[3 * eup + k] = p.m_in ? p.m_in[tm.offQ + (size_t)j * ld + i] : 0.0f;
                        est[4 * eup + k] = p.v_in ? p.v_in[tm.offQ + (size_t)i * ld + j] : 0.0f;
                        est[5 * eup + k] = p.v_in ? p.v_in[tm.offQ + (size_t)j * ld + i] : 0.0f;
                    }
                    est[6 * eup + k] = w;
                    const float dy = p.yhat[tm.offR + i] - p.yhat[tm.offR + j];
                    glap[k] = p.c_lap * 0.5f * dy * dy * inv_n2;
                } else {
                    asym = true;
                }
            }
            eupN += nt;
            eupF += ft;
            __syncthreads();
        }
        if (asym || eupN + eupF != eup) sh.bad = 1;
    }
    __syncthreads();  // the setup temporaries are dead from here on
    if (sh.bad) {
        fail_nan();
        return;
    }
    // ---------------- row arrays (columns beyond the widths must read as zero), model ----------------
    for (int e = tid; e < n * FS; e += NT) {
        gU1[e] = 0.0f;
        gU2[e] = 0.0f;
    }
    for (int e = tid; e <= nact; e += NT) gGe[e] = 0.0f;
    for (int e = tid; e < D * 32; e += NT) sW1[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + e];
    for (int e = tid; e < H * 32; e += NT) sW2[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 1024 + e];
    for (int e = tid; e < H * 32; e += NT) sW3[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 2048 + e];
    if (tid < 96) sh.bias[tid >> 5][tid & 31] = p.wts[WT_B + tid];
    for (int e = tid; e < C * 96; e += NT) sWp[e] = p.wts[WT_WP + e];
    if (tid < CMAX) sh.sbp[tid] = p.wts[WT_BP + tid];
    for (int e = tid; e < SPL_TDEG_MAX; e += NT) sG3[e] = 0.0f;
    if (have_xi)
        for (int r = tid; r < ld; r += NT) sXi[r] = 255;
    if (tid == 0) s_misc[1] = 0;
    if (tid < 32) {
        const float* fs = p.fs_in ? p.fs_in + (size_t)t * 3 * FS + tid : nullptr;   // gnnx_run_resume
        sh.fcur[tid] = (fs && tid < D) ? fs[0] : 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
        sh.mf[tid] = (fs && tid < D) ? fs[FS] : 0.0f;
        sh.vf[tid] = (fs && tid < D) ? fs[2 * FS] : 0.0f;
    }
    if (tid == 0) sAb[nact] = 0.0f;
    // sigma(M) -> symmetrised masked adjacency, one float per active directed entry
    for (int k = tid; k < eupN; k += NT) {
        int cij, cji;
        edge_entries(k, cij, cji);
        const float a = est[6 * eup + k] * (0.5f * (sigmoidf_(est[0 * eup + k]) + sigmoidf_(est[1 * eup + k])));
        sAb[cij] = a;
        sAb[cji] = a;
    }
    __syncthreads();
    // ---------------- feature dictionary: the distinct rows of X, if there are at most SPL_XD_MAX (bit-exact comparison) ----------------
    bool xdict = have_xi;      // (uniform; XL without room for a byte per node: the feature rows come from L2)
    int nd = 0;                // rows of the dictionary (uniform: every thread walks the same loop)
    if (have_xi) {
        for (;;) {
            if (tid == 0) s_misc[0] = 0x7fffffff;
            __syncthreads();
            for (int r = tid; r < n; r += NT)
                if (sXi[r] == 255) {  // the first row of this thread's stride that has no dictionary entry yet
                    atomicMin(&s_misc[0], r);
                    break;
                }
            __syncthreads();
            const int c = s_misc[0];
            if (c == 0x7fffffff) break;                       // every row is in the dictionary
            if (nd == SPL_XD_MAX || (nd >= 4 && 2 * s_misc[1] < n)) {  // too many distinct rows (or clearly dense features): L2 path
                xdict = false;
                break;
            }
            if (tid < sS) sXd[nd * sS + tid] = (tid < D) ? gX[c * FS + tid] : 0.0f;
            __syncthreads();
            int mine = 0;
            for (int r = tid; r < n; r += NT) {
                if (sXi[r] != 255) continue;
                bool eq = true;
                for (int k = 0; k < D; ++k) eq &= __float_as_uint(gX[r * FS + k]) == __float_as_uint(sXd[nd * sS + k]);
                if (eq) {
                    sXi[r] = (unsigned char)nd;
                    ++mine;
                }
            }
            if (mine) atomicAdd(&s_misc[1], mine);
            ++nd;
            __syncthreads();
        }
    }

    // one dictionary row = constant feature rows: layer 1's gather and the X part of dL/dAbar read no column, no index and no row (below)
    const bool xone = xdict && nd == 1;
    if constexpr (XL)
        if (xl.clk && tid == 0) xl.clk[4 * t + 1] = wall_clock64();
    for (int iter = 0; iter < p.num_iters; ++iter) {
        if (tid < 32) sh.phi[tid] = (tid < D) ? sigmoidf_(sh.fcur[tid]) : 0.0f;
        __syncthreads();
        const float step_size = adam_tab[2 * iter], bc2s = adam_tab[2 * iter + 1];
        float rbc2 = 1.0f / bc2s;   // once per iteration (adam_update<.., HAVE_R>)
        GNNX_OPAQUE(rbc2);
        float* Lrow = nullptr;      // LOG form: this target's row of the loss array for the current iteration
        if constexpr (LOG) Lrow = p.loss ? p.loss + ((size_t)t * p.num_iters + iter) * NLOSS : nullptr;
        // LOG form: sign bits of a row of normalised pre-activations (the ReLU gates, models.py:241, 251) -> trace word (iter, row, layer)
        auto trace_row = [&](int layer, int r, const float* urow) {
            unsigned bits = 0u;
            for (int c = 0; c < H; ++c) bits |= (urow[c] > 0.0f ? 1u : 0u) << c;
            p.trace_gates[((size_t)iter * (size_t)p.trace_rows + (size_t)(tm.offR + r)) * 2 + layer] = bits;
        };

        // ======== layer 1 on the rows within two hops: Zraw = Abar . X (kept for the feature-mask gradient), U1 ========
        SlotRaw nxA = slot_raw(0, 0);
        for (int round = 0; round < roundsA; ++round) {
            const SlotRaw curA = nxA;
            if (round + 1 < roundsA) nxA = slot_raw(0, round + 1);      // (in flight during this round)
            const RowSlot SA = slot_dec(curA, 0, round);
            if (!SA.wave_active) continue;  // uniform per wave
            const bool first = SA.first;
            const int r = first ? SA.row : 0;
            float acc[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) acc[q] = 0.0f;
            if (xone) {
                float x0[2 * DQ];
#pragma unroll
                for (int c = 0; c < 2 * DQ; ++c) x0[c] = sXd[c];      // (the dictionary's rows are padded to sS >= 2 DQ floats)
                sparse_gather_onerow<DQ, (DQ <= 5 ? 8 : 2)>(sAb, x0, D, SA.e0, SA.e1, h, acc);
            } else if (xdict) sparse_gather_dict<DQ, (DQ <= 5 ? 8 : 2)>(sAb, scol, sXi, sXd, sS, D, SA.e0, SA.e1, h, acc);
            else sparse_gather_rows<false, DQ, spl_gather_unroll(DQ)>(sAb, scol, gX, D, SA.e0, SA.e1, h, acc);
            sparse_combine<DQ>(acc, SA.rem, SA.wsplit);
            // Zraw for the feature-mask gradient: read back by the SAME lane in the backward, so every lane keeps its DQ values side by side (columns
            // 16 h + q of the row: a 16-byte vector per four values instead of DQ scattered dwords - round 6)
            static_assert(DQ <= 16, "Zraw: two lanes' values per 32-float row");
            if (first) {
#pragma unroll
                for (int q4 = 0; q4 + 3 < DQ; q4 += 4) {
                    f32x4 v = {acc[q4], acc[q4 + 1], acc[q4 + 2], acc[q4 + 3]};
                    *reinterpret_cast<f32x4*>(gZraw + r * FS + 16 * h + q4) = v;
                }
#pragma unroll
                for (int q = DQ & ~3; q < DQ; ++q) gZraw[r * FS + 16 * h + q] = acc[q];
            }
#pragma unroll
            for (int q = 0; q < DQ; ++q) acc[q] = (first && 2 * q + h < D) ? acc[q] * sh.phi[2 * q + h] : 0.0f;
            sparse_forward_rowlocal_global<DQ>(acc, sW1, sh.bias[0], D, H, li, h, first, gU1 + r * FS,
                                        sRn1 + round * (NT / 2) + wave * TILE + li);
        }
        __syncthreads();
        if constexpr (LOG)
            if (p.trace_gates)
                for (int round = 0; round < roundsA; ++round) {
                    const RowSlot SA = slot_of(0, round);
                    if (SA.wave_active && SA.first && h == 0) trace_row(0, SA.row, gU1 + SA.row * FS);
                }
        // ======== layer 2 on the target and its neighbours: U2 ========
        for (int round = 0; round < roundsB; ++round) {
            const RowSlot SB = slot_of(1, round);
            if (!SB.wave_active) continue;
            const bool first = SB.first;
            const int r = first ? SB.row : 0;
            float acc[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
            sparse_gather_rows<true, HQ, spl_gather_unroll(HQ)>(sAb, scol, gU1, H, SB.e0, SB.e1, h, acc);
            sparse_combine<HQ>(acc, SB.rem, SB.wsplit);
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = first ? acc[q] : 0.0f;
            sparse_forward_rowlocal_global<HQ>(acc, sW2, sh.bias[1], H, H, li, h, first, gU2 + r * FS,
                                        sRn2 + round * (NT / 2) + wave * TILE + li);
        }
        __syncthreads();
        if constexpr (LOG)
            if (p.trace_gates)
                for (int round = 0; round < roundsB; ++round) {
                    const RowSlot SB = slot_of(1, round);
                    if (SB.wave_active && SB.first && h == 0) trace_row(1, SB.row, gU2 + SB.row * FS);
                }
        // ======== row t of layer 3, head, dE, dZ3[t] ========
        {
            float z = 0.0f;
            if (li < H)
                for (int e = 2 * wave + h; e < degT; e += 2 * NW) z = fmaf(sAb[e], relu_(gU2[(int)scol[e] * FS + li]), z);
            z = xor32_sum(z);
            if (h == 0) sh.dfw[wave][li] = z;
        }
        __syncthreads();
        if (wave == 0) {
            const int c = li;
            float z = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) z += sh.dfw[w][c];
            if (h == 0) sh.z3[c] = z;
            wave_sync();
            float y = 0.0f;
            if (c < O) {
                for (int k = 0; k < H; ++k) y = fmaf(sh.z3[k], sW3[k * 33 + c], y);
                y += sh.bias[2][c];
            }
            const float ss = sum_lanes_0_31(y * y);
            const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
            const float u3 = y / rnorm;
            if (h == 0) {
                sh.e[64 + c] = u3;
                sh.e[c] = (c < H) ? relu_(gU1[tr * FS + c]) : 0.0f;
                sh.e[32 + c] = (c < H) ? relu_(gU2[tr * FS + c]) : 0.0f;
            }
            wave_sync();
            {
                const int cls = lane >> 3, part = lane & 7;
                float s = 0.0f;
                if (cls < C)
                    for (int q = part * 12; q < part * 12 + 12; ++q) s = fmaf(sWp[cls * 96 + q], sh.e[q], s);
                s += row_shl<4>(s);
                s += row_shl<2>(s);
                s += row_shl<1>(s);
                const float zc = __shfl(s, (lane & 7) * 8);
                const float zl = (lane < C) ? zc + sh.sbp[lane] : -3.0e38f;
                float mx = zl;
                mx = fmaxf(mx, row_shl<4>(mx));
                mx = fmaxf(mx, row_shl<2>(mx));
                mx = fmaxf(mx, row_shl<1>(mx));
                mx = bcast_first(mx);
                const float ex = (lane < C) ? expf(zl - mx) : 0.0f;
                float sum = ex;
                sum += row_shl<4>(sum);
                sum += row_shl<2>(sum);
                sum += row_shl<1>(sum);
                sum = bcast_first(sum);
                if (lane < CMAX) sh.g[lane] = (lane < C) ? ex / sum - ((lane == tm.y_gt) ? 1.0f : 0.0f) : 0.0f;
                if constexpr (LOG) {
                    if (Lrow && lane == tm.y_gt) Lrow[0] = -logf(ex / sum);   // explain.py:750-753
                    if (Lrow && lane < C && lane < LOGPN) Lrow[LOGP + lane] = ex / sum;   // the class probabilities the reference prints (explain.py:710-714, 157-158; the first LOGPN classes: include/gnnx.h)
                }
            }
            wave_sync();
#pragma unroll
            for (int part = 0; part < 2; ++part) {
                const int idx = lane + 64 * part;
                if (idx < 96) {
                    float s = 0.0f;
                    for (int cc = 0; cc < C; ++cc) s = fmaf(sWp[cc * 96 + idx], sh.g[cc], s);
                    sh.dEs[idx] = s;
                }
            }
            wave_sync();
            const float du3 = (h == 0 && c < O) ? sh.dEs[64 + c] : 0.0f;
            const float uq = (h == 0) ? u3 : 0.0f;
            const float s = sum_lanes_0_31(du3 * uq);
            const float dy3 = (du3 - uq * s) / rnorm;
            if (h == 0) sh.y3[c] = dy3;
            wave_sync();
            float v = 0.0f;
            if (c < H)
                for (int c2 = 0; c2 < O; ++c2) v = fmaf(sh.y3[c2], sW3[c * 33 + c2], v);
            if (h == 0) sh.dz3[c] = (c < H) ? v : 0.0f;
        }
        __syncthreads();
        // ======== dZ2 (rank-1) on the target and its neighbours, g3; dZ2 overwrites U2 row by row ========
        for (int round = 0; round < roundsB; ++round) {
            const RowSlot SB = slot_of(1, round);
            if (!SB.wave_active) continue;
            const bool first = SB.first;
            const int r = first ? SB.row : 0;
            const bool nbr = first && SB.bmask != 0xffffffffu;            // a neighbour of t: Abar[t][r] is the active entry SB.bmask
            const float art = nbr ? sAb[SB.bmask] : 0.0f;
            float du[HQ], uu[HQ];
            float gpart = 0.0f;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int c = 2 * q + h;
                const float u = (first && c < H) ? gU2[r * FS + c] : 0.0f;
                const float dz = (c < H) ? sh.dz3[c] : 0.0f;
                gpart = fmaf(dz, relu_(u), gpart);
                float dx = art * dz;
                if (first && r == tr && c < H) dx += sh.dEs[32 + c];
                du[q] = (u > 0.0f) ? dx : 0.0f;
                uu[q] = u;
            }
            gpart = xor32_sum(gpart);
            if (nbr && h == 0) sG3[SB.bmask] = gpart;
            const f32x16 c16 = sparse_backward_rowlocal<HQ>(du, uu, first ? sRn2[round * (NT / 2) + wave * TILE + li] : 1.0f, sW2, H, H, li, h);
            sparse_store_cols(c16, gU2 + r * FS, H, first, h);
        }
        __syncthreads();
        const float* gdZ2 = gU2;
        // ======== dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 on the rows within two hops; feature-mask gradient partials ========
        {
            float dfq[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) dfq[q] = 0.0f;
            SlotRaw nxB = slot_raw(0, 0);
            for (int round = 0; round < roundsA; ++round) {
                const SlotRaw curB = nxB;
                if (round + 1 < roundsA) nxB = slot_raw(0, round + 1);      // (in flight during this round)
                const RowSlot SA = slot_dec(curB, 0, round);
                if (!SA.wave_active) continue;
                const bool first = SA.first;
                const int r = first ? SA.row : 0;
                // the row's own U1 and Zraw values (L2): requested at the top of the round, consumed behind the gather / the row-local part (round 6:
                // they used to be two more dependent round trips in the middle of the round; wave-level fences keep the compiler from moving loads)
                float u1r[HQ], zr[DQ];
                {   // the whole row as 16-byte vectors (both lanes of the slot), this lane's columns 2q + h picked from them: half the load instructions
                    constexpr int NV = (2 * HQ + 3) / 4;
                    f32x4 v[NV];
#pragma unroll
                    for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const f32x4*>(gU1 + r * FS + 4 * k);
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        const float ve = v[(2 * q) >> 2][(2 * q) & 3], vo = v[(2 * q + 1) >> 2][(2 * q + 1) & 3];
                        u1r[q] = (first && 2 * q + h < H) ? (h ? vo : ve) : 0.0f;
                    }
                    f32x4 zv[(DQ + 3) / 4];      // Zraw: this lane's own values (layout: see layer 1)
#pragma unroll
                    for (int k = 0; k < (DQ + 3) / 4; ++k) zv[k] = *reinterpret_cast<const f32x4*>(gZraw + r * FS + 16 * h + 4 * k);
#pragma unroll
                    for (int q = 0; q < DQ; ++q) zr[q] = (first && 2 * q + h < D) ? zv[q >> 2][q & 3] : 0.0f;
                }
                float acc[HQ], uu[HQ];
#pragma unroll
                for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
                // dZ2 is non-zero only on t and its neighbours: only this slot's entries that point there contribute (marked in the
                // slot record), not the whole 64-entry chunk
                // (two marked entries per trip, one per lane of the slot, whole rows in 16-byte loads: see sparse_gather_rows)
                {
                    constexpr int NV = (2 * HQ + 3) / 4;
                    float full[2 * HQ];
#pragma unroll
                    for (int c = 0; c < 2 * HQ; ++c) full[c] = 0.0f;
                    unsigned long long m = (unsigned long long)SA.bmask | ((unsigned long long)SA.bmask_hi << 32);
                    while (m) {
                        const unsigned long long m1 = m & (m - 1ull);             // without the lowest marked entry
                        const unsigned long long mine = h ? m1 : m;               // lane 1 of the slot takes the second lowest
                        const bool have = mine != 0ull;
                        const int e = SA.e0 + (have ? __ffsll((long long)mine) - 1 : 0);
                        const float a = have ? sAb[e] : 0.0f;
                        const float* br = gdZ2 + (int)scol[e] * FS;
                        f32x4 v[NV];
#pragma unroll
                        for (int k = 0; k < NV; ++k) v[k] = *reinterpret_cast<const f32x4*>(br + 4 * k);
#pragma unroll
                        for (int c = 0; c < 2 * HQ; ++c) full[c] = fmaf(a, (EXACT || c < H) ? v[c >> 2][c & 3] : 0.0f, full[c]);
                        m = m1 & (m1 - 1ull);
                    }
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        const float mine = h ? full[2 * q + 1] : full[2 * q];
                        const float owed = h ? full[2 * q] : full[2 * q + 1];
                        acc[q] = mine + __shfl_xor(owed, 32);
                    }
                }
                sparse_combine<HQ>(acc, SA.rem, SA.wsplit);
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const int c = 2 * q + h;
                    const float u = u1r[q];
                    float dx = acc[q];
                    if (first && r == tr && c < H) dx += sh.dEs[c];
                    acc[q] = (u > 0.0f) ? dx : 0.0f;
                    uu[q] = u;
                }
                const f32x16 c16 = sparse_backward_rowlocal<HQ>(acc, uu, first ? sRn1[round * (NT / 2) + wave * TILE + li] : 1.0f, sW1, D, H, li, h);
                sparse_store_cols(c16, stage + li * sS, D, true, h);  // dZ1 of the lane's row, through LDS for the other half-lane
                wave_sync();
#pragma unroll
                for (int q = 0; q < DQ; ++q)
                    if (first && 2 * q + h < D) dfq[q] = fmaf(stage[li * sS + 2 * q + h], zr[q], dfq[q]);
                if (SA.e0 < SA.e1) {
                    // dL/dAbar on this slot's entries, row side (see k_sparse_resident): G[i][j] = dZ1[i] . (X[j] * phi) +
                    // dZ2[i] . relu(U1[j]); the row's dZ1 sits in the staging tile at its FIRST slot's lane (same 16-lane group)
                    const int ri = SA.row, lf = li - (SA.nsplit - SA.rem);
                    const bool inB = SA.inB;
                    float dz[2 * DQ], d2[2 * HQ];
#pragma unroll
                    for (int c = 0; c < 2 * DQ; ++c) dz[c] = (c < D) ? stage[lf * sS + c] * sh.phi[c] : 0.0f;
#pragma unroll
                    for (int c = 0; c < 2 * HQ; ++c) d2[c] = (inB && c < H) ? gdZ2[ri * FS + c] : 0.0f;
                    // PK entries per trip and lane: their X / U1 rows (L2) are all requested before the first product
                    constexpr int PK = EXACT ? 4 : 1;
                    // One dictionary row: dZ1[i] . (X[j] * phi) is the SAME number for every entry of the row - formed once, with the two chains of
                    // the general path (bit-identical); the entries of a row beyond t's neighbours (no dZ2 part: nearly all rows within two hops)
                    // are a stream of stores that reads neither columns nor rows, the others continue the two sums with their dZ2 . relu(U1[j]) terms.
                    float c0 = 0.0f, c1 = 0.0f;
                    if (xone) {
#pragma unroll
                        for (int c = 0; c < 2 * DQ; c += 2) {
                            c0 = fmaf(dz[c], (EXACT || c < D) ? sXd[c] : 0.0f, c0);
                            c1 = fmaf(dz[c + 1], (EXACT || c + 1 < D) ? sXd[c + 1] : 0.0f, c1);
                        }
                    }
                    if (xone && !inB) {
                        const float g = c0 + c1;
                        for (int e = SA.e0 + h; e < SA.e1; e += 2) gGe[e] = g;
                    } else
                    for (int e = SA.e0 + h; e < SA.e1; e += 2 * PK) {
                        int jj[PK];
                        float s0[PK], s1[PK];
#pragma unroll
                        for (int k = 0; k < PK; ++k) {
                            jj[k] = scol[(e + 2 * k < SA.e1) ? e + 2 * k : e];
                            s0[k] = 0.0f;
                            s1[k] = 0.0f;
                        }
                        if (xone) {    // uniform: the X part is the row's constant
#pragma unroll
                            for (int k = 0; k < PK; ++k) {
                                s0[k] = c0;
                                s1[k] = c1;
                            }
                        } else if (xdict) {   // uniform: the feature rows come from the LDS dictionary
#pragma unroll
                            for (int k = 0; k < PK; ++k) {
                                const float* x = sXd + (int)sXi[jj[k]] * sS;
#pragma unroll
                                for (int c = 0; c < 2 * DQ; c += 2) {
                                    s0[k] = fmaf(dz[c], (EXACT || c < D) ? x[c] : 0.0f, s0[k]);
                                    s1[k] = fmaf(dz[c + 1], (EXACT || c + 1 < D) ? x[c + 1] : 0.0f, s1[k]);
                                }
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < PK; ++k) {
                                const float* x = gX + jj[k] * FS;
#pragma unroll
                                for (int c = 0; c < 2 * DQ; c += 2) {
                                    s0[k] = fmaf(dz[c], x[c], s0[k]);
                                    s1[k] = fmaf(dz[c + 1], x[c + 1], s1[k]);
                                }
                            }
                        }
                        if (inB) {
#pragma unroll
                            for (int k = 0; k < PK; ++k) {
                                const float* u = gU1 + jj[k] * FS;
#pragma unroll
                                for (int c = 0; c < 2 * HQ; c += 2) {
                                    s0[k] = fmaf(d2[c], relu_(u[c]), s0[k]);
                                    s1[k] = fmaf(d2[c + 1], relu_(u[c + 1]), s1[k]);
                                }
                            }
                        }
#pragma unroll
                        for (int k = 0; k < PK; ++k)
                            if (e + 2 * k < SA.e1) gGe[e + 2 * k] = s0[k] + s1[k];
                    }
                }
                wave_sync();  // the staging tile is reused in the next round
            }
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                float v = dfq[q];
                v += row_shl<8>(v);
                v += row_shl<4>(v);
                v += row_shl<2>(v);
                v += row_shl<1>(v);
                v = xor16_sum(v);
                if (li == 0) sh.dfw[wave][2 * q + h] = v;
            }
        }
        __syncthreads();
        if (tid < D) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += sh.dfw[w][tid];
            sh.dfp[tid] = s;
        }
        // ======== per near edge: G_ij + G_ji, regulariser gradients, Adam in place on both directed entries, next Abar ========
        const bool republish = iter + 1 < p.num_iters;  // the returned mask is the one of the LAST forward (explain.py:209-211)
        float ls_size = 0.0f, ls_ent = 0.0f, ls_lap = 0.0f, ls_den = 0.0f, ls_adj = 0.0f;   // LOG form: this thread's part of the logged sums (its near edges, both directions)
        // EU edges per trip: the planes come from L2, and the loads of the later edges are in flight while the first is updated.  Round 6: four instead of
        // two - the phase is a quarter of an XL target's iteration (26 of 111 us at n = 17 k, tools/probe_xl_timeline.py) and is bound by those round
        // trips; a thread visits its edges in the same order, every edge's arithmetic is its own: bit-identical.
        constexpr int EU = 4;
        for (int k0 = tid; k0 < eupN; k0 += EU * NT) {
            int kk[EU], cij[EU], cji[EU];
            bool on[EU];
            float w[EU], lap[EU], Mij[EU], Mji[EU], mij[EU], mji[EU], vij[EU], vji[EU], G[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                on[u] = k0 + u * NT < eupN;
                kk[u] = on[u] ? k0 + u * NT : k0;
                edge_entries(kk[u], cij[u], cji[u]);
                w[u] = est[6 * eup + kk[u]];
                lap[u] = glap[kk[u]];
                Mij[u] = est[0 * eup + kk[u]];
                Mji[u] = est[1 * eup + kk[u]];
                mij[u] = est[2 * eup + kk[u]];
                mji[u] = est[3 * eup + kk[u]];
                vij[u] = est[4 * eup + kk[u]];
                vji[u] = est[5 * eup + kk[u]];
            }
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                G[u] = gGe[cij[u]] + gGe[cji[u]];                    // row-side products of both directions (layer-1 backward)
                G[u] += (cij[u] < degT) ? sG3[cij[u]] : 0.0f;        // i == t: the layer-3 part of row t of G (row t's entries come first)
                G[u] += (cji[u] < degT) ? sG3[cji[u]] : 0.0f;
            }
            auto update = [&](auto ADAMc) {      // one optimiser branch around the unrolled updates (see adam_update)
                constexpr bool ADAM = decltype(ADAMc)::value;
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const float gc = (0.5f * G[u] + lap[u]) * w[u];
                    if constexpr (LOG) {   // explain.py:755-770, 780-793 on the current iterate (before its update)
                        if (on[u]) {
                            const float Sa = sigmoidf_(Mij[u]), Sb = sigmoidf_(Mji[u]);
                            ls_size += Sa + Sb;
                            ls_ent += (-Sa * logf(Sa) - (1.0f - Sa) * logf(1.0f - Sa)) + (-Sb * logf(Sb) - (1.0f - Sb) * logf(1.0f - Sb));
                            ls_lap += w[u] * (0.5f * (Sa + Sb)) * (2.0f * lap[u]);   // Abar_ij c_lap (yhat_i - yhat_j)^2 / n^2: lap = c_lap / 2 dy^2 / n^2
                        }
                    }
                    {
                        const float S = sigmoidf_(Mij[u]);
                        const float g = (gc + p.c_size - p.c_ent * Mij[u] * inv_n2) * S * (1.0f - S);
                        adam_update<ADAM, true>(Mij[u], mij[u], vij[u], g, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt, rbc2);
                    }
                    {
                        const float S = sigmoidf_(Mji[u]);
                        const float g = (gc + p.c_size - p.c_ent * Mji[u] * inv_n2) * S * (1.0f - S);
                        adam_update<ADAM, true>(Mji[u], mji[u], vji[u], g, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt, rbc2);
                    }
                    if constexpr (LOG) {   // ExplainModule.mask_density (explain.py:680-683) after optimizer.step() (:142-148): the UPDATED entries
                        if (on[u]) {
                            ls_den += w[u] * (0.5f * (sigmoidf_(Mij[u]) + sigmoidf_(Mji[u])));
                            ls_adj += w[u];
                        }
                    }
                }
            };
            if (p.opt == 0) update(std::true_type{}); else update(std::false_type{});
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                if (!on[u]) continue;
                est[0 * eup + kk[u]] = Mij[u];
                est[1 * eup + kk[u]] = Mji[u];
                est[2 * eup + kk[u]] = mij[u];
                est[3 * eup + kk[u]] = mji[u];
                est[4 * eup + kk[u]] = vij[u];
                est[5 * eup + kk[u]] = vji[u];
                if (republish) {  // nobody reads sAb any more in this iteration (the barrier above)
                    const float a = w[u] * (0.5f * (sigmoidf_(Mij[u]) + sigmoidf_(Mji[u])));
                    sAb[cij[u]] = a;
                    sAb[cji[u]] = a;
                }
            }
        }
        if constexpr (LOG) {   // wave sums in lane order, then (after the barrier) the waves in order: a fixed summation order
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                ls_size += __shfl_xor(ls_size, o);
                ls_ent += __shfl_xor(ls_ent, o);
                ls_lap += __shfl_xor(ls_lap, o);
                ls_den += __shfl_xor(ls_den, o);
                ls_adj += __shfl_xor(ls_adj, o);
            }
            if (lane == 0) {
                sh.lsum[wave][0] = ls_size;
                sh.lsum[wave][1] = ls_ent;
                sh.lsum[wave][2] = ls_lap;
                sh.lsum[wave][3] = ls_den;
                sh.lsum[wave][4] = ls_adj;
            }
        }
        __syncthreads();
        if constexpr (LOG) {
            if (Lrow && wave == 0) {   // the entries off the edges were added to [1] and [3] by k_dead_entries; the far edges add theirs below
                const float phs = sum_lanes_0_31((lane < D) ? sh.phi[lane] : 0.0f);
                float a = 0.0f, b = 0.0f, c = 0.0f, den = 0.0f, adj = 0.0f;
                for (int w = 0; w < NW; ++w) {
                    a += sh.lsum[w][0];
                    b += sh.lsum[w][1];
                    c += sh.lsum[w][2];
                    den += sh.lsum[w][3];
                    adj += sh.lsum[w][4];
                }
                if (lane == 0) {
                    atomicAdd(&Lrow[LOGD + 1], den);   // mask density: numerator and denominator of the near edges; the far edges add theirs below, the
                    atomicAdd(&Lrow[LOGD + 2], adj);   // quotient is formed at the end of the launch
                    atomicAdd(&Lrow[1], p.c_size * a);
                    atomicAdd(&Lrow[2], c);
                    atomicAdd(&Lrow[3], p.c_ent * b * inv_n2);
                    Lrow[4] = p.c_feat_size * phs / (float)D;
                }
            }
        }
        if (tid < D) {  // feature mask
            const float ph = sh.phi[tid];
            const float gf = (sh.dfp[tid] + p.c_feat_size / (float)D) * ph * (1.0f - ph);
            float fn = sh.fcur[tid], m = sh.mf[tid], v = sh.vf[tid];
            adam_update(fn, m, v, gf, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            sh.fcur[tid] = fn;
            sh.mf[tid] = m;
            sh.vf[tid] = v;
        }
        __syncthreads();
    }
    if constexpr (XL)
        if (xl.clk && tid == 0) xl.clk[4 * t + 2] = wall_clock64();
    // ---------------- results: dense Abar block (zero off the edges), M on the edges, feature mask ----------------
    // (a separate zero-fill kernel in front of this launch was measured: it queues behind the resident launch on the other
    // stream and delays this one by more than the 0.2-0.8 ms the fill costs here)
    if constexpr (!XL) {
        f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!p.edge_only)   // (gnnx_hyper.edge_results_only: the caller reads Abar on the edges only - skip the ld^2 zero-fill)
            for (size_t e = (size_t)tid * 4; e < (size_t)ld * ld; e += 4 * NT) *reinterpret_cast<f32x4*>(p.Abar + tm.offQ + e) = z4;
    }
    __threadfence_block();
    __syncthreads();
    // one edge's results: the masked adjacency of the last forward, the mask entries and (on request) their moments - into the dense blocks, or
    // (XL) into the caller's edge lists at the edge's row-major index
    auto put_edge = [&](int k, int i, int j, float a, float Mij, float Mji, float mij, float mji, float vij, float vji) {
        if constexpr (XL) {
            const long long q = xl.eoff[t] + (long long)eidx[4 * (size_t)eup + k];
            xl.abar_e[q] = a;
            xl.M_e[2 * q] = Mij;
            xl.M_e[2 * q + 1] = Mji;
            if (xl.m_out_e) {
                xl.m_out_e[2 * q] = mij;
                xl.m_out_e[2 * q + 1] = mji;
            }
            if (xl.v_out_e) {
                xl.v_out_e[2 * q] = vij;
                xl.v_out_e[2 * q + 1] = vji;
            }
        } else {
            p.Abar[tm.offQ + (size_t)i * ld + j] = a;
            p.Abar[tm.offQ + (size_t)j * ld + i] = a;
            Mg[(size_t)i * ld + j] = Mij;
            Mg[(size_t)j * ld + i] = Mji;
            if (p.m_out) {
                p.m_out[tm.offQ + (size_t)i * ld + j] = mij;
                p.m_out[tm.offQ + (size_t)j * ld + i] = mji;
            }
            if (p.v_out) {
                p.v_out[tm.offQ + (size_t)i * ld + j] = vij;
                p.v_out[tm.offQ + (size_t)j * ld + i] = vji;
            }
        }
    };
    for (int k = tid; k < eupN; k += NT) {
        int i, j, cij, cji;
        edge_nodes(k, i, j);
        edge_entries(k, cij, cji);
        const float a = sAb[cij != nact ? cij : cji];   // a near edge has at least one of its two entries in a row of A
        put_edge(k, i, j, a, est[0 * eup + k], est[1 * eup + k], est[2 * eup + k], est[3 * eup + k], est[4 * eup + k], est[5 * eup + k]);
    }
    // ---------------- far edges: the whole trajectory of both mask entries in registers ----------------
    for (int k = eupN + tid; k < eup; k += NT) {
        int i, j;
        edge_nodes(k, i, j);
        const float w = est[6 * eup + k];
        const float gc = glap[k] * w;   // (0.5 G + lap) w with G = 0 exactly
        float Mij = est[0 * eup + k], Mji = est[1 * eup + k], mij = est[2 * eup + k], mji = est[3 * eup + k], vij = est[4 * eup + k],
              vji = est[5 * eup + k];
        float a = w * (0.5f * (sigmoidf_(Mij) + sigmoidf_(Mji)));
        for (int iter = 0; iter < p.num_iters; ++iter) {
            const float step_size = adam_tab[2 * iter], bc2s = adam_tab[2 * iter + 1];
            const float Si = sigmoidf_(Mij), Sj = sigmoidf_(Mji);
            a = w * (0.5f * (Si + Sj));   // the mask of this iteration's forward
            if constexpr (LOG)
                if (p.loss) {   // this far edge's share of the logged sums (float atomics: logging only, like k_dead_entries)
                    float* L = p.loss + ((size_t)t * p.num_iters + iter) * NLOSS;
                    atomicAdd(&L[1], p.c_size * (Si + Sj));
                    atomicAdd(&L[2], a * (2.0f * glap[k]));
                    atomicAdd(&L[3], p.c_ent * inv_n2 * ((-Si * logf(Si) - (1.0f - Si) * logf(1.0f - Si)) + (-Sj * logf(Sj) - (1.0f - Sj) * logf(1.0f - Sj))));
                }
            const float gi = (gc + p.c_size - p.c_ent * Mij * inv_n2) * Si * (1.0f - Si);
            const float gj = (gc + p.c_size - p.c_ent * Mji * inv_n2) * Sj * (1.0f - Sj);
            adam_update(Mij, mij, vij, gi, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            adam_update(Mji, mji, vji, gj, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            if constexpr (LOG)
                if (p.loss) {   // mask density after the step (explain.py:142-148, 680-683)
                    float* L = p.loss + ((size_t)t * p.num_iters + iter) * NLOSS;
                    atomicAdd(&L[LOGD + 1], w * (0.5f * (sigmoidf_(Mij) + sigmoidf_(Mji))));
                    atomicAdd(&L[LOGD + 2], w);
                }
        }
        put_edge(k, i, j, a, Mij, Mji, mij, mji, vij, vji);
    }
    if constexpr (LOG)
        if (p.loss) {   // every edge has added its share: the density of each epoch
            __threadfence_block();
            __syncthreads();
            for (int it = tid; it < p.num_iters; it += NT) {
                float* L = p.loss + ((size_t)t * p.num_iters + it) * NLOSS;
                L[LOGD] = L[LOGD + 1] / L[LOGD + 2];
            }
        }
    if constexpr (XL)
        if (xl.clk && tid == 0) xl.clk[4 * t + 3] = wall_clock64();
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? sh.fcur[tid] : 0.0f;
    if (p.fs_out && tid < FS) {
        float* fs = p.fs_out + (size_t)t * 3 * FS + tid;
        fs[0] = (tid < D) ? sh.fcur[tid] : 0.0f;
        fs[FS] = (tid < D) ? sh.mf[tid] : 0.0f;
        fs[2 * FS] = (tid < D) ? sh.vf[tid] : 0.0f;
    }
}

// gnnx_plan_analyze, every target with n <= SPL_N_MAX (also those of <= 512 rows that fit no resident class, e.g. more
// than 2048 edges): the quantities k_sparse_large's layout depends on, with the same hop levels and slot placement as the
// kernel computes.  out[SPL_COUNTS t ..] = directed entries, row slots of SPL_CHUNK entries of the rows within two hops
// (-1: a row that cannot be placed), the same for slots of SP_CHUNK entries (the 512-thread class of k_sparse_resident),
// directed entries of the rows within two hops, rows within two hops, row slots of SPL_CHUNK entries of t and its
// neighbours.  Targets outside (NLO, NMAX] are left alone (two instantiations: the LDS tables of the large one would
// halve the occupancy of a launch over thousands of small targets).
// every row's number of off-diagonal non-zeros, for the whole batch: one workgroup per 32-row block (a 5600-node block is
// scanned by 175 workgroups instead of one) -> rowdeg[R]
__global__ __launch_bounds__(256) void k_row_degrees(const float* A, const ConvTile* tiles, int32_t* rowdeg) {
    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int UN = 4;
    for (int rr = wave; rr < TILE; rr += 4) {
        const int r = tl.rb * TILE + rr;
        int cnt = 0;
        if (r < tm.n) {
            const float* row = A + tm.offQ + (size_t)r * tm.ld;
            for (int c0 = 0; c0 < tm.n; c0 += 64 * UN) {
                float a[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = c0 + 64 * u + lane;
                    a[u] = (c < tm.n && c != r) ? row[c] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) cnt += __popcll(__ballot(a[u] != 0.0f));
            }
        }
        if (lane == 0) rowdeg[tm.offR + r] = cnt;
    }
}

// (NTH threads per workgroup: 1024 for the large ranges; round 5: targets of up to 512 nodes take <0, 512, 256> - 2.5 KB of LDS and four
// waves instead of 20 KB and sixteen, so that a pipelined job's analysis finds room on a chip full of optimisation workgroups.)
template <int NLO, int NMAX, int NTH = 1024>
__global__ __launch_bounds__(NTH) GNNX_SERVICE_ATTR void k_count_edges_large(const TargetMeta* meta, const float* A, const int32_t* rowdeg, int32_t* out) {
    constexpr int NW = NTH / 64, UN = 8;   // the dense rows of t and its neighbours are scanned for the hop levels: NW waves x 8 chunks in flight
    __shared__ int deg[NMAX + 1];
    __shared__ unsigned char level[NMAX + 1];
    __shared__ int part[NW];
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int32_t* o = out + (size_t)SPL_COUNTS * blockIdx.x;
    if (tm.n <= NLO || tm.n > NMAX) {
        if (tm.n > SPL_N_MAX && tid < SPL_COUNTS) o[tid] = -1;
        return;
    }
    const float* Ag = A + tm.offQ;
    int cnt = 0;
    for (int r = tid; r < tm.n; r += 64 * NW) {   // degrees: k_row_degrees counted them
        const int d = rowdeg[tm.offR + r];
        deg[r] = d;
        cnt += d;
    }
#pragma unroll
    for (int o2 = 32; o2 >= 1; o2 >>= 1) cnt += __shfl_xor(cnt, o2);
    for (int r = tid; r < tm.n; r += 64 * NW) level[r] = (r == tm.t) ? 0 : 3;
    if (lane == 0) part[wave] = cnt;
    __syncthreads();
    for (int d = 1; d <= 2; ++d) {  // hop levels from the dense rows
        for (int r = wave; r < tm.n; r += NW) {
            if (level[r] != d - 1) continue;  // uniform per wave
            for (int c0 = 0; c0 < tm.n; c0 += 64 * UN) {
                float a[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = c0 + 64 * u + lane;
                    a[u] = (c < tm.n && c != r) ? Ag[(size_t)r * tm.ld + c] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int c = c0 + 64 * u + lane;
                    if (a[u] != 0.0f && level[c] > d) level[c] = (unsigned char)d;
                }
            }
        }
        __syncthreads();
    }
    if (tid < 3) {  // thread 0: slots of SPL_CHUNK entries (A), thread 1: slots of SP_CHUNK entries (A), thread 2: slots of SPL_CHUNK (B)
        const int chunk = (tid == 1) ? SP_CHUNK : SPL_CHUNK;
        const int lvlmax = (tid == 2) ? 1 : 2;
        int pos = 0, singles = 0, rows = 0, entries = 0;
        bool placeable = true;
        for (int r = 0; r < tm.n; ++r) {
            if (level[r] > lvlmax) continue;
            ++rows;
            entries += deg[r];
            if (deg[r] > chunk) {
                const int ns = sparse_slots_of_c(deg[r], chunk);
                placeable &= ns <= SP_MAX_SPLIT;
                pos = sparse_place(pos, ns) + ns;
            } else {
                ++singles;
            }
        }
        const int slots = placeable ? pos + singles : -1;
        if (tid == 0) {
            int total = 0;
            for (int w = 0; w < NW; ++w) total += part[w];
            o[0] = total;
            o[1] = slots;
            o[3] = entries;
            o[4] = rows;
        } else if (tid == 1) {
            o[2] = slots;
        } else {
            o[5] = slots;
        }
    }
}

// gnnx_plan_analyze: CSR (rowptr [ld + 1], ascending columns, the row of every entry) of every target routed to
// k_sparse_large, from its block of the packed dense adjacency.  k_csr_rowptr_large: one workgroup per such target turns the
// row degrees (k_row_degrees) into rowptr; k_csr_emit_large: one workgroup per 32-row block of the batch (blocks of other
// targets leave at once: csr_off[2 t] < 0) writes the columns and rows of its rows' entries.
__global__ __launch_bounds__(1024) void k_csr_rowptr_large(const TargetMeta* meta, const int32_t* rowdeg, const int32_t* targets,
                                                          const long long* csr_off, int32_t* csr_rowptr) {
    const int t = targets[blockIdx.x];
    const TargetMeta tm = meta[t];
    const int n = tm.n, ld = tm.ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int32_t* rowptr = csr_rowptr + csr_off[2 * t];
    __shared__ int srp[SPL_N_MAX + 34];
    for (int r = tid; r < ld; r += 1024) srp[r] = (r < n) ? rowdeg[tm.offR + r] : 0;
    __syncthreads();
    if (wave == 0) {
        const int total = wave_exclusive_scan_array(srp, ld, lane);
        if (lane == 0) srp[ld] = total;
    }
    __syncthreads();
    for (int r = tid; r <= ld; r += 1024) rowptr[r] = srp[r];
}

__global__ __launch_bounds__(256) void k_csr_emit_large(const float* A, const ConvTile* tiles, const long long* csr_off,
                                                        const int32_t* csr_rowptr, unsigned short* csr_col, unsigned short* csr_row) {
    const ConvTile tl = tiles[blockIdx.x];
    if (csr_off[2 * tl.t] < 0) return;   // not a target of k_sparse_large
    const TargetMeta tm = tl.tm;
    const int n = tm.n, ld = tm.ld;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int UN = 4;
    const int32_t* rowptr = csr_rowptr + csr_off[2 * tl.t];
    unsigned short* col = csr_col + csr_off[2 * tl.t + 1];
    unsigned short* row = csr_row + csr_off[2 * tl.t + 1];
    for (int rr = wave; rr < TILE; rr += 4) {
        const int r = tl.rb * TILE + rr;
        if (r >= n) continue;   // uniform per wave
        int base = rowptr[r];
        const float* arow = A + tm.offQ + (size_t)r * ld;
        for (int c0 = 0; c0 < n; c0 += 64 * UN) {
            float a[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = c0 + 64 * u + lane;
                a[u] = (c < n && c != r) ? arow[c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool nz = a[u] != 0.0f;
                const unsigned long long bal = __ballot(nz);
                if (nz) {
                    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
                    col[pos] = (unsigned short)(c0 + 64 * u + lane);
                    row[pos] = (unsigned short)r;
                }
                base += __popcll(bal);
            }
        }
    }
}

}  // namespace gnnx
