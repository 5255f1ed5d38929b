// gnnx_sparse.hpp — on-chip-resident mask optimisation over the EDGES of a target (node mode, n <= 512).
//
// The reference optimises a dense n x n mask (explain.py:583-663), but only the entries on edges of the sub-graph
// ever reach an output: Abar = A (.) sym(sigma(M)) is zero elsewhere (explain.py:665-678), the prediction, Laplacian
// and feature-mask terms see M only through Abar, the size and entropy terms are separable per entry (explain.py:
// 755-770; the entropy mean only contributes the known factor 1/n^2), and Adam is per entry.  The trajectory of an
// edge entry therefore does not depend on any non-edge entry, and the returned mask (explain.py:209-211) is zero off
// the edges.  This kernel keeps exactly the live state - (M, m, v) of the 2E directed edge entries - and runs all
// iterations of explain.py:137-146 for one target inside one workgroup:
//   * setup: a CSR view (rowptr, sorted columns) of the target's sub-adjacency is built in LDS from the packed dense
//     A block (wave ballots, one prefix scan); every thread takes up to QMAX undirected edges {(i,j),(j,i)}, i < j,
//     and keeps their mask entries and Adam moments in registers for the whole run;
//   * 16 waves, one per 32-row block (n <= 512): lane (r = lane & 31, half = lane >> 5) owns row r and the columns
//     2q + half.  The masked adjacency lives in LDS as one float per directed entry; the contractions Abar.B are
//     sparse row gathers from LDS into those registers (entry loop unrolled 4 deep so the dependent column -> row
//     loads of different entries overlap); the row-local parts (.W + b, L2 normalisation and its Jacobian, .W^T)
//     run on v_mfma_f32_32x32x2_f32 in transposed form (C[c][r]): with that lane mapping the gathered registers ARE
//     the B operand of MFMA step q, so nothing is staged, and the row norm is 16 registers + one cross-half shuffle;
//   * the only workgroup barriers are the ~15 phase boundaries of an iteration;
//   * HBM is touched at the start (A, M, X, yhat, model) and at the end (M on the edges, dense Abar, feature mask).
// Dead state: the non-edge entries of M are left at their initial values (the dense paths keep updating them; they
// are needed only for the logged size/entropy scalars, so loss logging stays on the dense streaming path).
// Mathematics as in gnnx_kernels.hpp / SURVEY.md Appendix A; parity: tests/test_emu_kernels.py, tests/test_gpu_parity.py.
#pragma once
#ifndef GNNX_OPAQUE_ROWS
#define GNNX_OPAQUE_ROWS 2
#endif
#ifndef GNNX_OPAQUE_ALL
#define GNNX_OPAQUE_ALL 1
#endif
#ifndef GNNX_OPAQUE_LANE
#define GNNX_OPAQUE_LANE 2
#endif
#ifndef GNNX_OPAQUE_LANE_SLIM
#define GNNX_OPAQUE_LANE_SLIM 2
#endif
#include <type_traits>
#include "gnnx_kernels.hpp"
#include "gnnx_resident.hpp"

namespace gnnx {

// Three size classes of the kernel (template parameter NT = threads per workgroup): a 32-row block per wave and two
// undirected edges per thread, so
//   NT = 1024: n <= 512, E <= 2048, 153 KB of LDS  (one workgroup per CU: 16 waves x 128 VGPRs fill its register file)
//   NT =  256: n <= 128, E <=  512,  72 KB         (two per CU)
//   NT =   64: n <=  32, E <=  128,  20 KB         (one wave: every barrier is wave-local; six and more per CU)
//   NT =  512: n <= 512, E <= 2048, 153 KB         (node mode; 8 waves = 256 row slots are plenty once the rows beyond
//              two hops need none, 4 edges per thread; two waves per SIMD leave 256 VGPRs: no spills, no scratch)
constexpr int SP_THREADS = 1024;             // the largest class (bounds shared tables)
__host__ __device__ constexpr int sp_qmax(int nt) { return nt == 512 ? 4 : 2; }             // undirected edges per thread
__host__ __device__ constexpr int sp_ld_max(int nt) { return nt == 512 ? 512 : 32 * (nt / 64); }
constexpr int SP_LD_MAX = 32 * (SP_THREADS / 64);
#ifndef GNNX_POOL512_FLOATS       // 38 272 floats + SparseFixed = 158.5 KB: 5.3 KB of the compute unit's LDS stay free for service kernels (39 168 filled it)
#define GNNX_POOL512_FLOATS 38272
#endif
__host__ __device__ constexpr int sp_pool_floats(int nt) { return nt >= 512 ? GNNX_POOL512_FLOATS : nt >= 256 ? 18432 : 5120; }
constexpr int SP_GATHER_UNROLL = 2;          // entries in flight per lane in the sparse gathers (4: measured, no gain)
// Forms measured one by one in round 4 (tools/gpu_r4n.sh ... gpu_r4r.sh; syn1 launch 3.18 -> 2.84 ms); the losers are gone, these stay switches:
constexpr bool SP_RELU_STORE = true;         // algebraic form: sU1 holds relu(U1), the row's owner recomputes its own U1 in the backward
constexpr bool SP_MERGE_PUBLISH = true;      // algebraic form, classes of up to 512 threads: the edge phase publishes the next masked adjacency itself
constexpr int SP_FAST_HEAD = 2;              // bit 0: the head's two normalisations, bit 1: its softmax, in the hardware forms (measured: see the head)
constexpr int SP_CHUNK = 16;                 // entries per row slot: longer rows are split over adjacent lanes of one wave
// row slots of a class: NT / 2 (two lanes = column halves per slot)

// Row slots: row r takes ns(r) = max(1, ceil(deg(r) / SP_CHUNK)) consecutive slots that must not straddle a 16-lane
// DPP row (so ns <= 16, i.e. degree <= 256; the partial sums are combined with row shifts).  Shared by k_count_edges
// (fit test) and the kernel (placement order: see "slot order" there).
__host__ __device__ inline int sparse_slots_of(int deg) { return deg <= SP_CHUNK ? 1 : (deg + SP_CHUNK - 1) / SP_CHUNK; }
constexpr int SP_MAX_SPLIT = 16;
__host__ __device__ inline int sparse_place(int pos, int ns) { return ((pos & 15) + ns > 16) ? ((pos + 15) & ~15) : pos; }

// carve-out of the LDS pool (float offsets) for a target of ld rows and nnz directed entries
struct SparseLayout {
    int sD, sH, sO;  // row strides (odd: conflict-free row-strided access)
    int oX, oU1, oU2, oU3, odZ2, odZ1, oAb, oGe, oCol, oRowptr, oArt, oRn1, oRn2, oRn3, oYhat, oG3, oW, oWp, total;
};
// graph = 0: node mode (GcnEncoderNode: only row t of layer 3 is needed; dZ2 overwrites U2);
// graph = 1: graph mode (GcnEncoderGraph: all three layers in full, U3 [ld][max(H, O)] overwritten by dZ3, dZ2 separate)
// n rows of the big row arrays (rows >= n are never touched), ld = round_up(n, 32) entries of the per-row scalars
// slim (round 6), bit 0: the algebraic constant-feature form (XC = 2) - X is one row (every row equals it) and the dZ1 array one float per row
// (the per-row scalar ci = dY1 . wt is all that form keeps of dZ1); the arrays the setup's temporaries alias are padded to the temporaries'
// size.  Every class of that form uses it: 22 floats per row less (n = 310 at hidden 32 fits the 512-thread class again; on BA-House x100k
// fewer (128, 512]-node targets fall to k_sparse_large).  bit 1: no private model block (k_sparse_resident_tiny: shared by the workgroup).
__host__ __device__ inline SparseLayout sparse_layout(int n, int ld, int nnz, int D, int H, int C, int graph = 0, int O = 0, int slim = 0) {
    SparseLayout L;
    L.sD = D | 1;
    L.sH = H | 1;
    L.sO = (O > H ? O : H) | 1;
    int o = 0;
    L.oX = o;      o += ((slim & 1) ? 1 : n) * L.sD;
    L.oU1 = o;     o += n * L.sH;
    L.oU2 = o;     o += n * L.sH;   // U2; node mode: overwritten row by row with dZ2 once the row's U2 has been consumed
    L.oU3 = o;     o += graph ? n * L.sO : 0;
    L.odZ2 = graph ? o : L.oU2;
    o += graph ? n * L.sH : 0;
    L.odZ1 = o;    o += (slim & 1) ? ld : n * L.sD;
    if ((slim & 1) && o < 7 * ld + 2 * 16 + 8) o = 7 * ld + 2 * 16 + 8;   // the setup's temporaries (7 ld + 2 SP_CHUNK + 8 ints) live in front of Abar
    L.oAb = o;     o += nnz;
    L.oGe = o;     o += graph ? 0 : nnz;  // node mode: dL/dAbar per directed entry (row-side product), written by the layer-1 backward
    L.oCol = o;    o += (nnz + 1) / 2;  // uint16 columns
    L.oRowptr = o; o += ld + 1;         // int
    L.oArt = o;    o += ld;
    L.oRn1 = o;    o += ld;
    L.oRn2 = o;    o += ld;
    L.oRn3 = o;    o += graph ? ld : 0;
    L.oYhat = o;   o += ld;
    L.oG3 = o;     o += ld;
    L.oW = o;      o += (slim & 2) ? 0 : (D + 2 * H) * 33;  // rows k < D of W1, k < H of W2, k < H of W3, 33-float rows
    L.oWp = o;     o += (slim & 2) ? 0 : C * 96;
    L.total = o;
    return L;
}
// does a target fit the class of nt threads?  (slots: row slots it needs with EVERY row placed, from k_count_edges)
__host__ __device__ inline bool sparse_fits(int nt, int n, int ld, int nnz, int slots, int D, int H, int C, int graph = 0,
                                            int O = 0, int slim_pool = 0, int slim = 0) {
    // the setup's temporaries (7 ld + 2 SP_CHUNK + 8 ints) live in the X / U1 / U2 / dZ1 arrays, which are contiguous (slim layouts pad them)
    // slim: the layout's form (sparse_layout); slim_pool > 0: in a pool of that many floats (k_sparse_resident_tiny) instead of the class's
    return ld <= sp_ld_max(nt) && nnz / 2 <= sp_qmax(nt) * nt && nnz < 65536 && slots >= 0 && slots <= nt / 2 &&
           C <= RES_CMAX && H >= 2 && ((slim & 1) || 2 * n * ((H | 1) + (D | 1)) >= 7 * ld + 2 * SP_CHUNK + 8) &&
           sparse_layout(n, ld, nnz, D, H, C, graph, O, slim).total <= (slim_pool ? slim_pool : sp_pool_floats(nt));
}

// a lane's row slot in one row set: the row (valid when first), its chunk of entries, the split bookkeeping
struct RowSlot {
    int row, e0, e1, nsplit, wsplit;
    int rem;  // slots from this one to the last slot of its row, itself included (1: nothing to add)
    bool first, wave_active;
    bool inB;  // the slot's row is t or a neighbour of t (its dZ2 row can be non-zero)
    unsigned bmask;  // bit k: entry e0 + k of the slot points at t or a neighbour of t (the only entries with a non-zero dZ2 row)
    unsigned bmask_hi;  // entries 32..63 (k_sparse_large: 64-entry slots)
};

// SLIM (round 6, k_sparse_resident_tiny): the block of a single-wave target of the algebraic node-mode form - the members only the LDS
// form of the head, graph mode, the logging form or the larger classes touch shrink to one element (those code paths are never
// taken there: the register head is unconditional for one wave, see `reg_head`), dfw to the two rows one wave writes: 521 floats
// instead of 1340, so that sixteen targets fit one compute unit's LDS.
template <bool SLIM>
struct SparseFixedT {
    static constexpr bool slim = SLIM;
    static constexpr int NWF = SLIM ? 2 : SP_THREADS / 64;
    static constexpr int LX = SLIM ? 1 : 32, LX96 = SLIM ? 1 : 96;
    float sbp[CMAX];
    float phi[32], fcur[32], mf[32], vf[32], bias[3][32];
    float z3[LX], y3[LX], dz3[32], e[LX96], g[CMAX], dEs[96], dfp[32], dfw[NWF][32];
    float sr3;
    int nnz, eup, bad;
    int xconst;  // node mode: every feature row of the sub-graph equals row 0 bit for bit (constant / featureless inputs)
    int set_rows[2], set_slots[2];
    int set_chunk[2];  // entries per row slot of the set: the smallest of {4, 8, 16} (8, 16 for set A) whose slots fit the class
    int chunk_b;       // node mode, set A: slot width of the rows of t and its neighbours (<= set_chunk[0]; see "slot width per row")
    int erow[LX96];  // graph mode: arg-max row of every pooled column
    float wt[32], vsum[LX];  // algebraic constant-feature form: (x (.) phi) W1; vsum (sum over the rows of s_r dY1[r]) lives in wave 0's registers since round 4 - the slot keeps the struct's size, which the LDS budget of the mixed launch is built on
    float lsum[SLIM ? 1 : SP_THREADS / 64][6];  // LOG form: per-wave partial sums over the owned edges: the logged size / entropy / Laplacian terms, [3] the masked adjacency AFTER the step and [4] the adjacency (mask density, explain.py:680-683)
};
using SparseFixed = SparseFixedT<false>;
using SparseFixedSlim = SparseFixedT<true>;

// every lane of a wave has finished its LDS accesses before any lane continues (LDS operations of one wave
// execute in order; the fences keep the compiler from moving accesses across)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// inclusive scan over the 64 lanes of a wave
__device__ __forceinline__ int wave_scan_inclusive(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl(v, lane - d);
        if (lane >= d) v += o;
    }
    return v;
}

// first position in [lo, hi) of the sorted uint16 array whose value is >= key
__device__ __forceinline__ int lower_bound_u16(const unsigned short* a, int lo, int hi, int key) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// sparse row gather for the lane's row: acc[q] += sum_e Abar_e * f(B[col_e][2q + half]), columns < W <= 2 NQ.
// Loads are unconditional (out-of-range columns are read from the padding / the next row and discarded by a select),
// so the compiler can issue all of an entry group's loads before the first use.
// max(x, 0) in one instruction (fmaxf costs two: it canonicalises its operands first)
__device__ __forceinline__ float relu_(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, __builtin_inff()); }

template <bool RELU, int NQ, bool EXACT, int UN>
__device__ __forceinline__ void sparse_gather_impl(const float* sAb, const unsigned short* scol, const float* B, int stride,
                                                   int W, int e0, int e1, int half, float (&acc)[NQ]) {
    // Software pipeline over groups of UN entries: the (Abar, column) pairs of the NEXT group are loaded while the
    // rows of the current one are in flight, so a group costs one LDS round trip, not two.  A short last group is
    // padded with zero-weight copies of the row's first entry (no tail loop).
    float a[UN];
    int cl[UN];
#pragma unroll
    for (int j = 0; j < UN; ++j) {
        const bool in = e0 + j < e1;
        a[j] = in ? sAb[in ? e0 + j : e0] : 0.0f;
        cl[j] = scol[in ? e0 + j : e0];
    }
#pragma unroll 1
    for (int e = e0; e < e1; e += UN) {
        const float* br[UN];
        float ac[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            br[j] = B + cl[j] * stride + half;
            ac[j] = a[j];
        }
        float b[UN][NQ];
#pragma unroll
        for (int j = 0; j < UN; ++j)
#pragma unroll
            for (int q = 0; q < NQ; ++q) b[j][q] = br[j][2 * q];
        const int en = e + UN;
        if (en < e1) {
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const bool in = en + j < e1;
                const int idx = in ? en + j : en;
                a[j] = in ? sAb[idx] : 0.0f;
                cl[j] = scol[idx];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const bool ok = EXACT || 2 * q + half < W;
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                float v = ok ? b[j][q] : 0.0f;
                if (RELU) v = relu_(v);
                acc[q] = fmaf(ac[j], v, acc[q]);
            }
        }
    }
}

// W == 2 NQ (every column of the trip count is real: the reference's D = 10, H = 20) needs no column selects
template <bool RELU, int NQ, int UN = SP_GATHER_UNROLL>
__device__ __forceinline__ void sparse_gather(const float* sAb, const unsigned short* scol, const float* B, int stride, int W,
                                              int e0, int e1, int half, float (&acc)[NQ]) {
    if (W == 2 * NQ)
        sparse_gather_impl<RELU, NQ, true, UN>(sAb, scol, B, stride, W, e0, e1, half, acc);
    else
        sparse_gather_impl<RELU, NQ, false, UN>(sAb, scol, B, stride, W, e0, e1, half, acc);
}

// Constant feature rows (every X[j] == X[0] bit for bit: the reference's synthetic datasets use ConstFeatureGen, gengraph.py:60-61,
// and featureless graphs get constant inputs): Abar . X needs neither the column of an entry nor the row it points at - the lane
// keeps its NQ columns of the one row in registers and the entry loop is a stream of independent loads of Abar alone.  Same
// products in the same order as sparse_gather (fmaf(Abar_e, X[col_e][c], acc) with X[col_e][c] == xq), so the results are
// bit-identical to the general path.
template <int NQ, int UN = 4>
__device__ __forceinline__ void sparse_gather_const(const float* sAb, const float (&xq)[NQ], int e0, int e1, float (&acc)[NQ]) {
#pragma unroll 1
    for (int e = e0; e < e1; e += UN) {
        float a[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const bool in = e + j < e1;
            const float v = sAb[in ? e + j : e0];
            a[j] = in ? v : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < UN; ++j) acc[q] = fmaf(a[j], xq[q], acc[q]);
    }
}

// forward row-local part for the lane's row: Y^T[c][r] = sum_k W[k][c] Z[r][k] on MFMA (the lane's registers zq[u] =
// Z[r][2u + half] are the B operand of step u), + bias, L2 normalisation; U -> sU[r][c], norm -> srn[r].
// DOUT_C = dout when it is known at compile time (the column predicates of the epilogue fold away), else 0.
// VEC4 (round 6, k_sparse_large: the row array lives in global memory, rows of FS = 32 floats, 128-byte aligned): the accumulator registers 4j .. 4j + 3 of
// a lane are the four CONSECUTIVE columns 8j + 4h .. + 3, so the row is stored as 16-byte vectors - three / two stores per lane for 20 columns instead of ten
// (the scattered dword stores of U1 and Zraw were 47 % of the large-target kernel's layer-1 loop: tools/probe_xl_timeline.py, knock-outs).  Same values.
template <int NQ, int DOUT_C, bool VEC4 = false>
__device__ __forceinline__ void sparse_forward_rowlocal_impl(const float (&zq)[NQ], const float* sW, const float* bias, int din,
                                                             int dout_rt, int li, int h, bool store, float* sUrow, float* srn_r) {
    static_assert(!VEC4 || (DOUT_C > 0 && DOUT_C % 4 == 0), "vector stores: a compile-time width, a multiple of four");
    const int dout = DOUT_C ? DOUT_C : dout_rt;
    float bv[16];  // bias of this lane's 16 columns: loaded before the MFMA chain, consumed after it
#pragma unroll
    for (int g = 0; g < 16; ++g) bv[g] = (DOUT_C && (g & 3) + 8 * (g >> 2) >= DOUT_C) ? 0.0f : bias[acc_row(g, h)];
    f32x16 c16;
#pragma unroll
    for (int g = 0; g < 16; ++g) c16[g] = 0.0f;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int k = 2 * u + h;
        const float a = (k < din) ? sW[k * 33 + li] : 0.0f;
        const float b = (k < din) ? zq[u] : 0.0f;
        c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16, 0, 0, 0);
    }
    float sp[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // four partial sums: the 16 squares are not one dependent FMA chain
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int c = acc_row(g, h);
        c16[g] = (c < dout) ? c16[g] + bv[g] : 0.0f;
        sp[g & 3] = fmaf(c16[g], c16[g], sp[g & 3]);
    }
    float ss = (sp[0] + sp[1]) + (sp[2] + sp[3]);
    ss = xor32_sum(ss);
    const float rnorm = fmaxf(sqrt_(ss), 1e-12f);
    const float rinv = rcp_(rnorm);
    if constexpr (VEC4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (8 * j >= DOUT_C) continue;              // no lane holds a real column in this group
            const int base = 8 * j + 4 * h;
            f32x4 v = {c16[4 * j] * rinv, c16[4 * j + 1] * rinv, c16[4 * j + 2] * rinv, c16[4 * j + 3] * rinv};
            if (store && base < DOUT_C) *reinterpret_cast<f32x4*>(sUrow + base) = v;
        }
    } else {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int c = acc_row(g, h);
        if (DOUT_C && (g & 3) + 8 * (g >> 2) >= DOUT_C) continue;  // no lane of this register holds a real column
        if (store && c < dout) sUrow[c] = c16[g] * rinv;
    }
    }
    if (store && h == 0) *srn_r = rnorm;
}

// the same for a row in global memory (16-byte aligned, FS floats): vector stores at the reference's width
template <int NQ>
__device__ __forceinline__ void sparse_forward_rowlocal_global(const float (&zq)[NQ], const float* sW, const float* bias, int din,
                                                               int dout, int li, int h, bool store, float* gUrow, float* srn_r) {
    if (dout == 20)
        sparse_forward_rowlocal_impl<NQ, 20, true>(zq, sW, bias, din, dout, li, h, store, gUrow, srn_r);
    else if (dout == 16)
        sparse_forward_rowlocal_impl<NQ, 16, true>(zq, sW, bias, din, dout, li, h, store, gUrow, srn_r);
    else if (dout == 32)
        sparse_forward_rowlocal_impl<NQ, 32, true>(zq, sW, bias, din, dout, li, h, store, gUrow, srn_r);
    else
        sparse_forward_rowlocal_impl<NQ, 0>(zq, sW, bias, din, dout, li, h, store, gUrow, srn_r);
}

template <int NQ>
__device__ __forceinline__ void sparse_forward_rowlocal(const float (&zq)[NQ], const float* sW, const float* bias, int din,
                                                        int dout, int li, int h, bool store, float* sUrow, float* srn_r) {
    if (dout == 20)  // the reference's hidden / output width
        sparse_forward_rowlocal_impl<NQ, 20>(zq, sW, bias, din, dout, li, h, store, sUrow, srn_r);
    else if (dout == 16)  // (round 6: the other compile-time widths, --hidden-dim 16 / 32)
        sparse_forward_rowlocal_impl<NQ, 16>(zq, sW, bias, din, dout, li, h, store, sUrow, srn_r);
    else if (dout == 32)
        sparse_forward_rowlocal_impl<NQ, 32>(zq, sW, bias, din, dout, li, h, store, sUrow, srn_r);
    else
        sparse_forward_rowlocal_impl<NQ, 0>(zq, sW, bias, din, dout, li, h, store, sUrow, srn_r);
}

// row[k] = c16[g] for the columns k = acc_row(g, half) < kmax this lane holds; KC = kmax when known at compile time (the
// registers that hold no real column for either half are skipped without a test), else 0
template <int KC>
__device__ __forceinline__ void sparse_store_cols_impl(const f32x16& c16, float* row, int kmax_rt, bool pred, int h) {
    const int kmax = KC ? KC : kmax_rt;
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        if (KC && (g & 3) + 8 * (g >> 2) >= KC) continue;
        const int k = acc_row(g, h);
        if (pred && k < kmax) row[k] = c16[g];
    }
}
__device__ __forceinline__ void sparse_store_cols(const f32x16& c16, float* row, int kmax, bool pred, int h) {
    if (kmax == 20) sparse_store_cols_impl<20>(c16, row, kmax, pred, h);       // hidden width of the reference
    else if (kmax == 10) sparse_store_cols_impl<10>(c16, row, kmax, pred, h);  // input width of the reference
    else if (kmax == 16) sparse_store_cols_impl<16>(c16, row, kmax, pred, h);
    else if (kmax == 32) sparse_store_cols_impl<32>(c16, row, kmax, pred, h);
    else sparse_store_cols_impl<0>(c16, row, kmax, pred, h);
}

// backward row-local part for the lane's row: dY = (dU - U (dU.U)) / r (dU, U in registers, columns 2q + half), then
// dZ^T[k][r] = sum_c W[k][c] dY[r][c] on MFMA; returns dZ[r][acc_row(g, half)] in c16
template <int NQ, bool WIDE = false>
__device__ __forceinline__ f32x16 sparse_backward_rowlocal(const float (&du)[NQ], const float (&uu)[NQ], float rnorm,
                                                          const float* sW, int din, int dout, int li, int h) {
    float sd2[2] = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < NQ; ++q) sd2[q & 1] = fmaf(du[q], uu[q], sd2[q & 1]);
    float sdot = xor32_sum(sd2[0] + sd2[1]);
    const float rinv = rcp_(rnorm);
    f32x16 c16;
#pragma unroll
    for (int g = 0; g < 16; ++g) c16[g] = 0.0f;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int c = 2 * u + h;
        // unconditional load + select where the address is in range by construction (rows li < 32 of a (D + 2H >= 32)-row weight
        // block): a lane-varying condition around a load costs an exec-mask region, a select one instruction
        const float aw = (WIDE || (li < din && c < dout)) ? sW[li * 33 + c] : 0.0f;
        const float a = (li < din && c < dout) ? aw : 0.0f;
        const float b = (c < dout) ? (du[u] - uu[u] * sdot) * rinv : 0.0f;
        c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c16, 0, 0, 0);
    }
    return c16;
}

// rows split over several slots: the first slot's lanes add the partial sums of the following lanes (same column half)
// in slot order; wsplit = the wave's longest split (uniform), nsplit = this row's
// lane l <- value of lane l + S of the same 16-lane row (0 beyond the row): DPP row_shl, no LDS traffic
template <int S>
__device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + S, 0xf, 0xf, true));
}

// lane 0 of every 16-lane row <- sum of the row (the other lanes hold partial sums)
__device__ __forceinline__ float row_sum16(float v) {
    v += row_shl<8>(v);
    v += row_shl<4>(v);
    v += row_shl<2>(v);
    v += row_shl<1>(v);
    return v;
}
// every lane <- the value of the first lane (through an SGPR)
__device__ __forceinline__ float bcast_first(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// every lane <- sum over lanes 0..31 (4 DPP shifts + one cross-row shuffle + SGPR broadcast instead of 5-6 ds_bpermute)
__device__ __forceinline__ float sum_lanes_0_31(float v) {
    const float r = row_sum16(v);
    return bcast_first(xor16_sum(r));
}

// Rows split over several slots (adjacent lanes of one 16-lane DPP row, same column half): a segmented suffix sum in
// log2(16) = 4 DPP steps - after the steps S = 1, 2, 4, 8 the row's FIRST slot holds the sum of all its slots (lane l adds
// lane l + S while that lane still belongs to its row: S < rem).  wsplit = the wave's longest split (uniform): later
// steps are skipped.  Fixed order, no LDS traffic.
template <int NQ, int S>
__device__ __forceinline__ void sparse_combine_step(float (&acc)[NQ], int rem, int wsplit) {
    if constexpr (S < SP_MAX_SPLIT) {
        if (S < wsplit) {  // uniform per wave
            // one v_fmac_f32 with a DPP source per register and step (acc += shifted * m, m = 1 while the source lane still belongs to this
            // row, else 0: the same sum as a select + add, exactly) instead of v_mov_dpp + v_cndmask + v_add through one temporary
            const float m = (S < rem) ? 1.0f : 0.0f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = fmaf(row_shl<S>(acc[q]), m, acc[q]);
            sparse_combine_step<NQ, 2 * S>(acc, rem, wsplit);
        }
    }
}

template <int NQ>
__device__ __forceinline__ void sparse_combine(float (&acc)[NQ], int rem, int wsplit) {
    sparse_combine_step<NQ, 1>(acc, rem, wsplit);
}

// DQ >= ceil(D / 2), HQ >= ceil(max(H, O) / 2): compile-time trip counts of the column loops (instantiated for the
// reference's encoders and for the general 32-wide case).  GRAPH: GcnEncoderGraph (models.py:269-316: three full
// layers, per-layer max-pool over all rows, no Laplacian term) instead of GcnEncoderNode (models.py:363-376).
// The body works on NT consecutive threads tid = 0 .. NT-1 with their own LDS pool and SparseFixed: a whole workgroup
// (k_sparse_resident) or, for NT = 64, one wave of a larger workgroup (k_sparse_resident_mixed) - a single wave
// synchronises with itself, so its barriers are wave-level.
// XC != 0: every feature row of the target equals its row 0 bit for bit (decided per plan by gnnx_plan_analyze_features, verified here):
// layer 1 and the X part of dL/dAbar then need no gathers; a compile-time form because a run-time test inside the phases keeps the
// operands of both forms alive across them (the 512-thread class sits at 256 VGPRs).  Node mode, exact shapes.
//   XC = 1: the general form's products in the general form's order (sparse_gather_const): bit-identical results;
//   XC = 2: the algebraic form.  With X[j] = x for every j, row r of layer 1 depends on ONE scalar, s_r = sum_e Abar_e:
//           Y1[r] = s_r wt + b1, wt = (x (.) phi) W1 (a 20-vector, refreshed when phi changes), so the forward needs neither the D-wide
//           gather nor its MFMA chain; backward, with dY1[r] = (dU - U (dU . U)) / r in the lane's registers,
//             dL/dAbar[r][j] (layer-1 part) = dZ1[r] . (x (.) phi) = dY1[r] . wt           (one 20-term dot per row, no dZ1 at all),
//             dL/dphi[k] = sum_r dZ1[r][k] (Abar X)[r][k] = x_k W1[k] . (sum_r s_r dY1[r])   (one 20-vector reduced over the rows).
//           Same mathematics, other rounding: within round-off of XC = 1 (tests: closed form, decision parity), not bit-identical.
// LOG: the logging form - per iteration the loss scalars of explain.py:808-819 (prediction, and the size / entropy / Laplacian sums
// over the entries on EDGES; the entries off the edges follow a closed scalar recursion each and are added by k_dead_entries) and the
// decision trace (Params::trace_gates / trace_pool).  A separate instantiation: the hot form carries none of it.  Round 5: also of the
// algebraic constant-feature form (XC = 2), so that a logging / tracing run executes the arithmetic of the form the plan SHIPS - the
// decision-parity tests judge the default form, not its general sibling (the gates: relu(U1) > 0 <=> U1 > 0, so the stored activation serves).
// pair_flag != nullptr: this body shares its workgroup - and therefore every __syncthreads() - with a second 256-thread body working on
// another target (k_sparse_resident_mixed, "pair" workgroups).  The two run the same code, so their barrier sequences are identical as long
// as they take the same side of the one target-dependent choice that changes it (`fuseB`): they agree on it through *pair_flag.
// EX = false: the trip counts DQ / HQ bound the widths (D <= 2 DQ, H, O <= 2 HQ) but the widths themselves are run-time values - encoders whose
// widths are not the reference's take the SMALLEST such instantiation that holds them instead of the 32-wide one (round 5: <16, 16> carries
// 700-1000 B of scratch per lane and ran a --hidden-dim 16 encoder 4x slower than the reference's widths; profiles/r05_generic_widths_syn1.txt).
// SH = SparseFixedSlim + slim_pool > 0 (floats of `pool`): the slim LDS form of a single-wave target (sparse_layout: slim) - same arithmetic,
// same order, so bit-identical to the full form; k_sparse_resident_tiny packs sixteen such targets on a compute unit.
template <int DQ, int HQ, bool GRAPH, int NT, int XC = 0, bool LOG = false, bool EX = true, class SH = SparseFixed>
__device__ __forceinline__ void sparse_resident_body(const Params p, int t, const float* adam_tab, float* pool, SH& sh,
                                                     int tid, float* shared_w = nullptr, int* pair_flag = nullptr, int slim_pool = 0) {
    constexpr bool SLIM = SH::slim;
    constexpr int LAY = SLIM ? 3 : (XC == 2 ? 1 : 0);   // the LDS layout's form (sparse_layout: slim)
    constexpr bool SLAY = (LAY & 1) != 0;
    static_assert(!SLIM || (NT == 64 && XC == 2 && !LOG && !GRAPH && EX), "the slim form: single-wave targets of the algebraic node-mode form, exact widths");
    constexpr int SCAN = (sp_ld_max(NT) + 63) / 64;  // rows per lane in the setup prefix scans
    constexpr int SP_QMAX = sp_qmax(NT);
    auto SYNC = []() {
        if constexpr (NT == 64) wave_sync(); else __syncthreads();
    };
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, ld = tm.ld, tr = tm.t;
    const int wave = tid >> 6;
    int lane = tid & 63, li = lane & 31, h = lane >> 5;   // (not const: GNNX_OPAQUE_LANE below)
    constexpr int NW = NT / 64;
    // The <5, 10> / <7, 10> instantiations serve exactly the reference's encoders (node: D = 10, graph: D = 14; H = O = 20): the
    // widths are compile-time constants there (every column predicate, row stride and trip count folds); other shapes take <16, 16>.
    constexpr bool EXACT = EX && (DQ != 16);
    constexpr bool RS = (XC == 2) && SP_RELU_STORE;   // algebraic form: sU1 holds relu(U1) (see layer 1)
    // algebraic form: the next masked adjacency is published by the edge phase itself and the feature mask / wt are refreshed by wave 0 in
    // front of it - one workgroup barrier per iteration fewer and no serial section between two barriers (see the edge phase)
    // (measured: syn5 - 64- / 256-thread classes - 2.82 -> 2.77 ms; the 512-thread class 2.84 -> 2.87 ms while it spilled, 2.81 -> 2.77 ms since
    // it does not - see GNNX_OPAQUE at the top of the iteration)
    constexpr bool MP = (XC == 2) && SP_MERGE_PUBLISH && NT <= 512;
    const int D = EXACT ? 2 * DQ : p.D, H = EXACT ? 2 * HQ : p.H, O = EXACT ? 2 * HQ : p.O, C = p.C;
    const float* Ag = p.A + tm.offQ;
    float* Mg = p.M + tm.offQ;

    // ---------------- setup 1: degrees (wave per row, ballot over 64-column chunks) ----------------
    int* tmp_deg = reinterpret_cast<int*>(pool);  // the pool is free until the layout is fixed
    const bool ld_ok = ld <= sp_ld_max(NT);
    if (ld_ok)
        for (int r = wave; r < ld; r += NW) {
            int cnt = 0;
            if (r < n)
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int c = c0 + lane;
                    const bool nz = (c < n && c != r) ? (Ag[(size_t)r * ld + c] != 0.0f) : false;
                    cnt += __popcll(__ballot(nz));
                }
            if (lane == 0) tmp_deg[r] = cnt;
        }
    SYNC();
    if (wave == 0 && ld_ok) {  // exclusive prefix sum over the rows: SCAN rows per lane
        int loc[SCAN], s = 0;
#pragma unroll
        for (int k = 0; k < SCAN; ++k) {
            const int r = lane * SCAN + k;
            loc[k] = (r < ld) ? tmp_deg[r] : 0;
            s += loc[k];
        }
        const int incl = wave_scan_inclusive(s, lane);
        int run = incl - s;
#pragma unroll
        for (int k = 0; k < SCAN; ++k) {
            const int r = lane * SCAN + k;
            if (r < ld) tmp_deg[r] = run;  // becomes rowptr[r]
            run += loc[k];
        }
        if (lane == 63) sh.nnz = incl;
    }
    SYNC();
    const int nnz = ld_ok ? sh.nnz : 0;
    const bool fits = ld_ok && sparse_fits(NT, n, ld, nnz, 0, D, H, C, GRAPH, O, SLIM ? slim_pool : 0, LAY) && (!SLIM || (slim_pool > 0 && shared_w));  // the slot count is checked once the slots are placed
    if (!fits) {
        // the plan promised a target that fits (gnnx_plan_analyze); anything else must fail loudly, not silently
        const float qnan = __builtin_nanf("");
        for (int e = tid; e < ld * ld; e += NT) p.Abar[tm.offQ + e] = qnan;
        if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = qnan;
        return;
    }
    const SparseLayout L = sparse_layout(n, ld, nnz, D, H, C, GRAPH, O, LAY);
    // rowptr currently sits at the start of the pool = inside the future sX region: move it through registers
    const int rp_keep = (tid < ld) ? tmp_deg[tid] : nnz;
    SYNC();
    int* rowptr = reinterpret_cast<int*>(pool + L.oRowptr);
    if (tid < ld) rowptr[tid] = rp_keep;
    if (tid == 0) rowptr[ld] = nnz;
    if (tid == 0) {
        sh.bad = 0;
        sh.xconst = 1;
    }
    SYNC();
    float* sX = pool + L.oX;
    float* sU1 = pool + L.oU1;
    float* sU2 = pool + L.oU2;   // node mode: == sdZ2 (see SparseLayout)
    float* sU3 = pool + L.oU3;   // graph mode only: U3, then dZ3
    float* sdZ2w = pool + L.odZ2;
    float* sRn3 = pool + L.oRn3;
    float* sdZ1 = pool + L.odZ1;
    float* sAb = pool + L.oAb;
    float* sGe = pool + L.oGe;
    unsigned short* scol = reinterpret_cast<unsigned short*>(pool + L.oCol);
    float* sArt = pool + L.oArt;
    float* sRn1 = pool + L.oRn1;
    float* sRn2 = pool + L.oRn2;
    float* sYhat = pool + L.oYhat;
    float* sG3 = pool + L.oG3;
    // the model block (W1 | W2 | W3 | Wp, read-only): the target's own copy at the end of its pool, or one staged by the caller
    // for all the targets of a workgroup (k_sparse_resident_mixed: the single-tile targets then fit eight to a workgroup)
    float* sW1 = shared_w ? shared_w : pool + L.oW;
    float* sW2 = sW1 + D * 33;
    float* sW3 = sW2 + H * 33;
    float* sWp = shared_w ? shared_w + (D + 2 * H) * 33 : pool + L.oWp;
    const int sD = L.sD, sH = L.sH, sO = L.sO;

    // ---------------- setup 2: sorted column lists ----------------
    for (int r = wave; r < n; r += NW) {
        int base = rowptr[r];
        for (int c0 = 0; c0 < n; c0 += 64) {
            const int c = c0 + lane;
            const bool nz = (c < n && c != r) ? (Ag[(size_t)r * ld + c] != 0.0f) : false;
            const unsigned long long bal = __ballot(nz);
            if (nz) scol[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)c;
            base += __popcll(bal);
        }
    }
    SYNC();
    // ---------------- setup 3: upper entries (col > row) are the tail of every row; prefix of their counts ----------------
    int* u0 = reinterpret_cast<int*>(sX);  // [ld] first upper entry of the row   (X, U1, U2, dZ1 are free during setup)
    int* upptr = u0 + ld;                    // [ld + 1]
    if (tid < ld) {
        const int a = rowptr[tid], b = rowptr[tid + 1];
        const int f = lower_bound_u16(scol, a, b, tid + 1);
        u0[tid] = f;
        upptr[tid] = b - f;  // count, scanned below
    }
    SYNC();
    if (wave == 0) {
        int loc[SCAN], s = 0;
#pragma unroll
        for (int k = 0; k < SCAN; ++k) {
            const int r = lane * SCAN + k;
            loc[k] = (r < ld) ? upptr[r] : 0;
            s += loc[k];
        }
        const int incl = wave_scan_inclusive(s, lane);
        int run = incl - s;
#pragma unroll
        for (int k = 0; k < SCAN; ++k) {
            const int r = lane * SCAN + k;
            if (r < ld) upptr[r] = run;
            run += loc[k];
        }
        if (lane == 63) {
            upptr[ld] = incl;
            sh.eup = incl;
        }
    }
    // Hop levels (node mode).  Row t of layer 3 is all the reference reads (explain.py:713), and every layer is
    // row-local apart from the contraction with Abar, so layer 2 is needed only on t and its neighbours (level <= 1) and
    // layer 1 only on rows within two hops (level <= 2); the same holds for their gradients (dZ2 / dZ1 are exactly zero
    // elsewhere).  On syn1's largest target (n = 310) that is 5 and 69 rows.  The other rows - the third hop - still own
    // mask entries, which keep receiving their regulariser gradients in the edge phase.  Graph mode pools over all rows.
    int* level = upptr + ld + 1;  // [ld]
    if (tid < ld) level[tid] = (GRAPH || tid == tr) ? 0 : 3;
    SYNC();
    if (!GRAPH)
        for (int d = 1; d <= 2; ++d) {
            if (tid < n && level[tid] == d - 1)
                for (int e = rowptr[tid]; e < rowptr[tid + 1]; ++e)
                    if (level[scol[e]] > d) level[scol[e]] = d;  // benign race: every writer stores d
            SYNC();
        }
    // Row slots in "slot order", one table per row set (A: level <= 2, layer 1 and its backward; B: level <= 1, layer 2
    // and its backward): rows longer than SP_CHUNK first (row order, never straddling a 16-lane DPP row), then the
    // other rows by decreasing degree, so that the 32 slots of a wave have (nearly) equal trip counts in the gathers.
    int* slot_tab = level + ld;  // per set: slot_start [ld + 1], order [ld], bucket [SP_CHUNK + 1]
    constexpr int NSET = GRAPH ? 1 : 2;
    if ((tid & 31) == 0 && (tid >> 5) < NSET) {  // one thread per set
        const int set = tid >> 5;
        const int lvlmax = GRAPH ? 9 : 2 - set;
        int* slot_start = slot_tab + set * (2 * ld + SP_CHUNK + 2);
        int* order = slot_start + ld + 1;
        int* bucket = order + ld;
        // Slot width of this set.  The plan's fit test guarantees SP_CHUNK-entry slots; the gathers are latency chains over a
        // slot's entries, so when the class has slots to spare the rows are cut into shorter chunks (more lanes per row):
        // the smallest width whose slots fit (<= NT / 2 slots, <= 16 slots per row = one DPP row).
        int ch = SP_CHUNK;
        for (int cand = (set == NSET - 1 && !GRAPH) ? 4 : 8; cand < SP_CHUNK; cand <<= 1) {
            int tot = 0, singles = 0;
            bool ok = true;
            for (int rr = 0; rr < n; ++rr) {
                if (level[rr] > lvlmax) continue;
                const int d = rowptr[rr + 1] - rowptr[rr];
                if (d > cand) {
                    const int ns = (d + cand - 1) / cand;
                    ok &= ns <= SP_MAX_SPLIT;
                    tot = sparse_place(tot, ns) + ns;
                } else {
                    ++singles;
                }
            }
            if (ok && tot + singles <= NT / 2) {
                ch = cand;
                break;
            }
        }
        sh.set_chunk[set] = ch;
        // Slot width per row (node mode, set A).  In the layer-1 backward the slots of t's and its neighbours' rows carry the expensive
        // part - a 20-term product per ENTRY (dZ2[i] . relu(U1[j])), where every other row of the set only stores one number per
        // entry - and they are few rows with, in a scale-free graph, many entries (a motif node's neighbours include hubs).  They take
        // the narrowest slots that still fit the class (4, 8 entries: two to four times as many lanes share those products and the
        // gathers of those rows); a row too long for 16 slots of that width doubles it.
        int chb = ch;
        if (!GRAPH && set == 0) {
            for (int cand = 4; cand < ch; cand <<= 1) {
                int tot = 0, singles = 0;
                for (int rr = 0; rr < n; ++rr) {
                    if (level[rr] > lvlmax) continue;
                    const int d = rowptr[rr + 1] - rowptr[rr];
                    int w = level[rr] <= 1 ? cand : ch;
                    while ((d + w - 1) / w > SP_MAX_SPLIT) w <<= 1;
                    if (d > w) {
                        const int ns = (d + w - 1) / w;
                        tot = sparse_place(tot, ns) + ns;
                    } else {
                        ++singles;
                    }
                }
                if (tot + singles <= NT / 2) {
                    chb = cand;
                    break;
                }
            }
        }
        if (set == 0) sh.chunk_b = chb;
        auto row_width = [&](int rr, int d) {
            int w = (!GRAPH && set == 0 && level[rr] <= 1) ? chb : ch;
            while ((d + w - 1) / w > SP_MAX_SPLIT && w < SP_CHUNK) w <<= 1;
            return w;
        };
        int pos = 0, p = 0, cnt = 0;
        for (int d = 0; d <= SP_CHUNK; ++d) bucket[d] = 0;
        for (int rr = 0; rr < n; ++rr) {
            if (level[rr] > lvlmax) continue;
            ++cnt;
            const int d = rowptr[rr + 1] - rowptr[rr];
            const int ch = row_width(rr, d);   // (shadows the set's width from here to the end of the row's placement)
            if (d > ch) {
                const int ns = (d + ch - 1) / ch;
                if (ns > SP_MAX_SPLIT) sh.bad = 1;
                pos = sparse_place(pos, ns);
                order[p] = rr;
                slot_start[p] = pos;
                pos += ns;
                ++p;
            } else {
                bucket[d]++;
            }
        }
        int run = p;
        for (int d = ch; d >= 0; --d) {
            const int c = bucket[d];
            bucket[d] = run;
            run += c;
        }
        for (int rr = 0; rr < n; ++rr) {
            if (level[rr] > lvlmax) continue;
            const int d = rowptr[rr + 1] - rowptr[rr];
            if (d <= row_width(rr, d)) order[bucket[d]++] = rr;
        }
        for (int q = p; q < cnt; ++q) slot_start[q] = pos + (q - p);
        slot_start[cnt] = pos + (cnt - p);
        sh.set_rows[set] = cnt;
        sh.set_slots[set] = pos + (cnt - p);
        if (pos + (cnt - p) > NT / 2) sh.bad = 1;
    }
    SYNC();
    const int eup = sh.eup;
    // this lane's row slot in every set: lanes (li, half 0) and (li, half 1) of wave w share slot 32 w + li
    RowSlot rs[NSET];
#pragma unroll
    for (int k = 0; k < NSET; ++k) {
        const int* slot_start = slot_tab + k * (2 * ld + SP_CHUNK + 2);
        const int* order = slot_start + ld + 1;
        RowSlot z;
        z.row = 0;
        z.e0 = z.e1 = 0;
        z.nsplit = 1;
        z.rem = 1;
        z.first = false;
        z.inB = false;
        z.bmask = 0u;
        z.bmask_hi = 0u;
        const int sl = wave * TILE + li, cnt = sh.set_rows[k];
        if (sl < sh.set_slots[k] && !sh.bad) {
            int lo = 0, hi = cnt;  // largest position in slot order with slot_start[.] <= sl
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (slot_start[mid] <= sl) lo = mid; else hi = mid;
            }
            const int row = order[lo];
            const int ra = rowptr[row], rb = rowptr[row + 1];
            int ch = (!GRAPH && k == 0 && level[row] <= 1) ? sh.chunk_b : sh.set_chunk[k];   // slot width of this row (see above)
            while ((rb - ra + ch - 1) / ch > SP_MAX_SPLIT && ch < SP_CHUNK) ch <<= 1;
            const int ns = (rb - ra <= ch) ? 1 : (rb - ra + ch - 1) / ch, kk = sl - slot_start[lo];
            if (kk < ns) {  // otherwise: padding slot in front of a split row
                z.row = row;
                z.e0 = ra + kk * ch;
                z.e1 = (z.e0 + ch < rb) ? z.e0 + ch : rb;
                z.nsplit = ns;
                z.rem = ns - kk;
                z.first = (kk == 0);
                z.inB = GRAPH || level[row] <= 1;
                for (int e = z.e0; e < z.e1; ++e)
                    if (GRAPH || level[scol[e]] <= 1) z.bmask |= 1u << (e - z.e0);
            }
        }
        int wsplit = z.first ? z.nsplit : 1;  // longest split row of this wave in this set (uniform)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int other = __shfl_xor(wsplit, o);
            wsplit = other > wsplit ? other : wsplit;
        }
        z.wsplit = __builtin_amdgcn_readfirstlane(wsplit);   // (the same in every lane: say so, and the tests on it are scalar branches)
        z.wave_active = wave * TILE < sh.set_slots[k];
        rs[k] = z;
    }
    const RowSlot& SA = rs[0];         // rows of layer 1 / its backward (all rows in graph mode)
    const RowSlot& SB = rs[NSET - 1];  // rows of layer 2 / its backward
    // Node mode: layer 2, row t of layer 3, the head and dZ2 only concern t and its neighbours.  When their row slots fit one
    // wave (the usual case: a motif node has a handful of neighbours), wave 0 runs those four phases back to back with
    // wave-level syncs while the other waves wait at ONE workgroup barrier instead of four.
    const bool fuseB = !GRAPH && sh.set_slots[NSET - 1] <= TILE;
    // A pair workgroup's two bodies share every workgroup barrier, so they must execute the same NUMBER of them: wave-level syncs in those
    // phases only when BOTH fuse (*pair_flag starts at 1; the barriers of the setup lie in between), else workgroup barriers - padded to the
    // count of the unfused path (see the head).  Which ARITHMETIC a body runs (the register head, who gathers row t) depends on its own
    // target alone (`fuseB`), so a target's result does not depend on its partner: pair == alone, bit for bit.
    bool fuse_sync = fuseB;
    if (pair_flag) {
        if (tid == 0 && !fuseB) *pair_flag = 0;
        SYNC();
        fuse_sync = *pair_flag != 0;
    }
    auto SYNC_B = [&]() {
        if (!fuse_sync) SYNC();
        else if (wave == 0) wave_sync();
    };
    // owned undirected edges: k = tid + NT q; mask entries and Adam moments stay in registers
    float Mij[SP_QMAX], Mji[SP_QMAX], mij[SP_QMAX], mji[SP_QMAX], vij[SP_QMAX], vji[SP_QMAX], wgt[SP_QMAX];
    float Sij[SP_QMAX], Sji[SP_QMAX];  // sigma(M) of the current iterate (computed when the masked adjacency is published)
    unsigned epk[SP_QMAX], npk[SP_QMAX];   // (eij | eji << 16), (i | j << 16): rows and directed entries are < 65536 (sparse_fits) - two registers per edge instead of four
    int eflag[SP_QMAX];   // bit 0 / 1: row i / j lies within two hops of t (dZ1 can be non-zero there); bit 2 / 3: row i / j is t
                          // or a neighbour of t (dZ2 can be non-zero there); graph mode: all set
    // Edge k of the target belongs to thread k mod NT in round k / NT - except in the LAST round, which is dealt from the top thread down:
    // that round is usually partial, and wave 0, whose chain through the feature-mask update makes it the last to reach the edge phase,
    // then owns one edge fewer than the waves that wait for it (n = 310, 1432 edges, 512 threads: two instead of three).  Which thread
    // updates an edge does not enter its arithmetic.  (The logging form keeps the plain order: its per-thread sums are order dependent.)
    // (Measured in round 5 and dropped: the feature-mask chain - dL/dphi -> Adam -> phi -> wt, between the barrier behind the layer-1 backward
    // and the edge phase - on the LAST wave, which then owned no edges, instead of wave 0: launch 2.695 -> 2.69 ms, nothing.  What the timeline
    // books as "wait at barrier + dfp 0.96 us" on wave 0 is mostly the wait for the other waves' layer-1 backward, not the chain.)
    const int qlast = eup > 0 ? (eup - 1) / NT : 0;
    auto edge_k = [&](int q) { return (!LOG && q == qlast) ? (NT - 1 - tid) + NT * q : tid + NT * q; };
    {
        bool asym = (2 * eup != nnz);
#pragma unroll
        for (int q = 0; q < SP_QMAX; ++q) {
            const int k = edge_k(q);
            Mij[q] = Mji[q] = mij[q] = mji[q] = vij[q] = vji[q] = wgt[q] = 0.0f;
            Sij[q] = Sji[q] = 0.5f;
            epk[q] = npk[q] = 0u;
            eflag[q] = 0;
            if (k < eup) {
                int lo = 0, hi = ld;  // largest row i with upptr[i] <= k
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (upptr[mid] <= k) lo = mid; else hi = mid;
                }
                const int i = lo;
                const int e = u0[i] + (k - upptr[i]);
                const int j = scol[e];
                const int em = lower_bound_u16(scol, rowptr[j], rowptr[j + 1], i);
                if (em >= rowptr[j + 1] || (int)scol[em] != i) asym = true;
                npk[q] = (unsigned)i | ((unsigned)j << 16);
                epk[q] = (unsigned)e | ((unsigned)(asym ? e : em) << 16);
                wgt[q] = Ag[(size_t)i * ld + j];
                if (Ag[(size_t)j * ld + i] != wgt[q]) asym = true;
                Mij[q] = Mg[(size_t)i * ld + j];
                Mji[q] = Mg[(size_t)j * ld + i];
                if (p.m_in) {  // gnnx_run_resume: Adam moments of the two entries
                    mij[q] = p.m_in[tm.offQ + (size_t)i * ld + j];
                    mji[q] = p.m_in[tm.offQ + (size_t)j * ld + i];
                }
                if (p.v_in) {
                    vij[q] = p.v_in[tm.offQ + (size_t)i * ld + j];
                    vji[q] = p.v_in[tm.offQ + (size_t)j * ld + i];
                }
                eflag[q] = GRAPH ? 15 : ((level[i] <= 2) | ((level[j] <= 2) << 1) | ((level[i] <= 1) << 2) | ((level[j] <= 1) << 3));
            }
        }
        if (asym) sh.bad = 1;  // benign race: every writer stores 1
    }
    SYNC();  // the setup temporaries (aliasing X .. dZ1) are dead from here on
    if (sh.bad) {     // asymmetric adjacency: not a graph the reference explains; fail loudly
        const float qnan = __builtin_nanf("");
        for (int e = tid; e < ld * ld; e += NT) p.Abar[tm.offQ + e] = qnan;
        if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = qnan;
        return;
    }

    // rows outside a phase's row set are never written: their U1 / U2 (= dZ2) / dZ1 must read as zero
    for (int e = tid; e < n * sH; e += NT) {
        sU1[e] = 0.0f;
        sU2[e] = 0.0f;
    }
    for (int e = tid; e < (SLAY ? ld : n * sD); e += NT) sdZ1[e] = 0.0f;   // (slim form: one float per row, see sparse_layout)
    if (!GRAPH)  // entries of rows beyond two hops are never written: their row-side products are exactly zero
        for (int e = tid; e < nnz; e += NT) sGe[e] = 0.0f;
    // ---------------- load features, model, labels ----------------
    for (int e = tid; e < (SLAY ? 1 : n) * 32; e += NT) {   // (slim form: row 0 stands for every row; verified against the others below)
        const int r = e >> 5, c = e & 31;
        // (the padding column of an even D is written too: the run-time-width forms multiply it by an exact zero, and 0 x stale LDS garbage
        // is NaN when the garbage is - found by the D = 8 case of test_mixed_launch_with_other_encoder_widths on the GPU, round 5)
        if (c < sD) sX[r * sD + c] = (c < D) ? p.X[(tm.offR + r) * FS + c] : 0.0f;
    }
    // (a shared model block is written by every wave that uses it - the same values to the same addresses, so each wave only
    // needs its own writes to have landed, which its next wave-level sync guarantees: no workgroup barrier)
    for (int e = tid; e < D * 32; e += NT) sW1[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + e];
    for (int e = tid; e < H * 32; e += NT) sW2[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 1024 + e];
    for (int e = tid; e < H * 32; e += NT) sW3[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 2048 + e];
    for (int e = tid; e < 96; e += NT) sh.bias[e >> 5][e & 31] = p.wts[WT_B + e];
    for (int e = tid; e < C * 96; e += NT) sWp[e] = p.wts[WT_WP + e];
    if (tid < CMAX) sh.sbp[tid] = p.wts[WT_BP + tid];
    if (tid < ld) sYhat[tid] = GRAPH ? 0.0f : p.yhat[tm.offR + tid];  // graph mode has no Laplacian term (explain.py:780)
    if (tid < 32) {
        const float* fs = p.fs_in ? p.fs_in + (size_t)t * 3 * FS + tid : nullptr;   // gnnx_run_resume
        sh.fcur[tid] = (fs && tid < D) ? fs[0] : 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
        sh.mf[tid] = (fs && tid < D) ? fs[FS] : 0.0f;
        sh.vf[tid] = (fs && tid < D) ? fs[2 * FS] : 0.0f;
    }
    const float inv_n2 = 1.0f / ((float)n * (float)n);
    const int rt0 = rowptr[tr], rt1 = rowptr[tr + 1];
    // this lane's row (waves beyond the target's row blocks idle through the row phases)
    float zraw[DQ];
    float sraw = 0.0f;   // algebraic constant-feature form: s_r = sum of the row's masked adjacency (what Zraw collapses to)

    // sigma(M) -> symmetrised masked adjacency, one float per directed entry
    // sArt = Abar[t][.] as a dense row (rank-1 layer-3 backward): zero except on t's neighbours, whose entries the owners of
    // the edges at t refresh in publish_abar
    if (tid < ld) sArt[tid] = 0.0f;
    if (tid < 32)   // sigmoid of the initial feature mask (0), or of the resumed one
        sh.phi[tid] = (tid < D) ? (p.fs_in ? sigmoidf_(p.fs_in[(size_t)t * 3 * FS + tid]) : 0.5f) : 0.0f;
    SYNC();
    if constexpr (XC) {   // constant feature rows?  (benign race: every writer stores 0; visible after publish_abar's barrier)
        bool same = true;
        for (int e = tid; e < n * D; e += NT) {
            const int r = e / D, c = e - r * D;
            if constexpr (SLAY) same &= __float_as_uint(p.X[(tm.offR + r) * FS + c]) == __float_as_uint(sX[c]);   // (the other rows stay in global memory)
            else same &= __float_as_uint(sX[r * sD + c]) == __float_as_uint(sX[c]);
        }
        if (!same) sh.xconst = 0;
    }
    auto publish_abar = [&]() {
#pragma unroll
        for (int q = 0; q < SP_QMAX; ++q)
            if (edge_k(q) < eup) {
                Sij[q] = sigmoidf_(Mij[q]);   // kept for the next update: sigma'(M) = S (1 - S)
                Sji[q] = sigmoidf_(Mji[q]);
                const float a = wgt[q] * (0.5f * (Sij[q] + Sji[q]));
                sAb[epk[q] & 0xffffu] = a;
                sAb[epk[q] >> 16] = a;
                if (!GRAPH) {
                    if ((int)(npk[q] & 0xffffu) == tr) sArt[npk[q] >> 16] = a;
                    if ((int)(npk[q] >> 16) == tr) sArt[npk[q] & 0xffffu] = a;
                }
            }
        SYNC();
    };
    // algebraic constant-feature form: wt = (x (.) phi) W1, by wave 0 (its lanes c < H), whenever phi has changed
    auto update_wt = [&]() {
        if (wave == 0 && h == 0 && li < H) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 2 * DQ; ++k) a = fmaf(sX[k] * sh.phi[k], sW1[k * 33 + li], a);
            sh.wt[li] = a;
        }
    };
    float step_size = 0.0f, bc2s = 0.0f, rbc2 = 0.0f;   // this iteration's optimiser scalars (set at the top of the loop)
    auto feature_mask_step = [&](int k) {   // one thread per feature column k < D: Adam on the feature mask from dfp, then phi for the next iteration
        const float ph = sh.phi[k];
        const float gf = (sh.dfp[k] + p.c_feat_size / (float)D) * ph * (1.0f - ph);
        float fn = sh.fcur[k], m = sh.mf[k], v = sh.vf[k];
        adam_update<false, true>(fn, m, v, gf, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt, rbc2);
        sh.fcur[k] = fn;
        sh.mf[k] = m;
        sh.vf[k] = v;
        sh.phi[k] = sigmoidf_(fn);  // for the next iteration's layer 1 / edge phase (this iteration's readers are done)
    };
    if constexpr (XC == 2) update_wt();   // (sX, the model block and phi are in place since the barrier above; publish_abar's barrier publishes wt)
    publish_abar();
    if constexpr (XC) {   // the plan promised constant feature rows for THIS X (gnnx_plan_analyze_features): anything else must fail loudly
        if (sh.xconst == 0) {
            const float qnan = __builtin_nanf("");
            for (int e = tid; e < ld * ld; e += NT) p.Abar[tm.offQ + e] = qnan;
            if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = qnan;
            return;
        }
    }
    // LOG form: sign bits of a row of normalised pre-activations (the ReLU gates, models.py:241, 251) -> trace word (iter, row, layer)
    auto trace_row = [&](int iter, int layer, int r, const float* urow) {
        unsigned bits = 0u;
        for (int c = 0; c < H; ++c) bits |= (urow[c] > 0.0f ? 1u : 0u) << c;
        p.trace_gates[((size_t)iter * (size_t)p.trace_rows + (size_t)(tm.offR + r)) * 2 + layer] = bits;
    };
    float* Lrow = nullptr;   // LOG form: this target's row of the loss array for the current iteration
    for (int iter = 0; iter < p.num_iters; ++iter) {
        step_size = adam_tab[2 * iter];
        bc2s = adam_tab[2 * iter + 1];
        rbc2 = 1.0f / bc2s;      // once per iteration (adam_update<.., HAVE_R>)
        GNNX_OPAQUE(rbc2);
        // Round 5, under the 224-register cap of the mixed kernel (room for the prepare stage's kernels, gnnx_kernels.hpp): what the compiler
        // spilled were loop INVARIANTS it had hoisted out of this loop - the LDS addresses derived from a lane's row slots and (below) from its
        // owned edges' index words.  Declared modified here they are per-iteration values again: a few shifts and adds where they are used
        // instead of scratch reloads along the chain - spilled registers 74 -> 23, scratch loads per iteration 14 / 65 / 18 -> 0 / 20 / 6
        // (512-thread / pair / single-wave body), syn1 loop-only 2.805 -> 2.766 ms (the uncapped kernel's time), steady state 306.8 -> 314 k
        // nodes/s (tools/gpu_r5af.sh).  Also measured: the head's lane index (spills 9, but more instructions on wave 0's chain: 312.5 k) and the
        // lane's column index everywhere (spills 5: no difference).  (No instruction is emitted.)
        // (node mode only: the uncapped graph-mode kernels of config 4 lose 3 % with it - 52.5 -> 54.3 ms, tools/gpu_r5ah.sh)
        if constexpr (GNNX_OPAQUE_ROWS && !GRAPH) {
#pragma unroll
            for (int k = 0; k < NSET; ++k) {
                GNNX_OPAQUE(rs[k].row);
                if constexpr (GNNX_OPAQUE_ROWS > 1) {
                    GNNX_OPAQUE(rs[k].e0);
                    GNNX_OPAQUE(rs[k].e1);
                }
            }
        }
        if constexpr (NT == 512 || (GNNX_OPAQUE_ALL && !GRAPH)) {
            // The 512-thread class sits at its 256 registers and spills.  What the compiler keeps across the whole iteration, per owned edge, is
            // not only the edge's state but the LDS ADDRESSES it derives from the two packed index words (ten per edge: both entries of Abar
            // and of the row-side products, yhat, g3 and Abar[t][.] of both end points - loop invariant, hence hoisted out of the iteration
            // loop: 40 registers for four edges).  Declaring the packed words modified here makes those addresses per-iteration values again:
            // a handful of shifts and adds in the edge phase instead of scratch stores and reloads along wave 0's chain.  (No instruction is emitted.)
#pragma unroll
            for (int q = 0; q < SP_QMAX; ++q) {
                GNNX_OPAQUE(epk[q]);
                GNNX_OPAQUE(npk[q]);
            }
        }
        // Round 6: the lane's own indices.  Every LDS address a lane derives from them (its column of a weight row, its bias entries, its row of
        // the head block, ...) is loop invariant too, and with run-time array bases each (array, index pattern) pair is a register of its own
        // held across the whole iteration: declared modified here they are formed where they are used.  Levels: 1 = li, 2 = + lane, 3 = + h
        // (spilled registers of k_sparse_resident_mixed<5, 10, 2> under the 224-register cap: 30 / 9 / 2 / 0; scratch 92 / 40 / 12 / 0 B per lane).
        constexpr int OL = SLIM ? GNNX_OPAQUE_LANE_SLIM : GNNX_OPAQUE_LANE;
        if constexpr (OL > 0 && !GRAPH) {
            GNNX_OPAQUE(li);
            if constexpr (OL > 1) GNNX_OPAQUE(lane);
            if constexpr (OL > 2) GNNX_OPAQUE(h);
        }
        if constexpr (LOG) Lrow = p.loss ? p.loss + ((size_t)t * p.num_iters + iter) * NLOSS : nullptr;

        // ======== layer 1: Zraw = Abar . X (kept in registers for the feature-mask gradient), U1 ========
        if (SA.wave_active) {
            const bool first = SA.first;
            const int r = first ? SA.row : 0, re0 = SA.e0, re1 = SA.e1, nsplit = SA.nsplit, wsplit = SA.wsplit;
            if constexpr (XC == 2) {
                float a1[1] = {0.0f};
                const float one[1] = {1.0f};
                sparse_gather_const<1>(sAb, one, re0, re1, a1);
                sparse_combine<1>(a1, SA.rem, wsplit);
                sraw = first ? a1[0] : 0.0f;
                float y[HQ];
                float sp[2] = {0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    y[q] = fmaf(sraw, sh.wt[2 * q + h], sh.bias[0][2 * q + h]);
                    sp[q & 1] = fmaf(y[q], y[q], sp[q & 1]);
                }
                const float ss = xor32_sum(sp[0] + sp[1]);
                const float rnorm = fmaxf(sqrt_(ss), 1e-12f);
                const float rinv = rcp_(rnorm);
                if (first) {
                    // (RS: the array holds relu(U1) - every reader of ANOTHER row wants the activation: the layer-2 gather, the per-entry
                    // products of the backward, row t's entry of the head; the row's owner recomputes its own U1 from s_r in the backward)
#pragma unroll
                    for (int q = 0; q < HQ; ++q) sU1[r * sH + 2 * q + h] = RS ? relu_(y[q] * rinv) : y[q] * rinv;
                    if (h == 0) sRn1[r] = rnorm;
                }
            } else {
            float acc[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) acc[q] = 0.0f;
            if constexpr (XC) {
                float xq[DQ];
#pragma unroll
                for (int q = 0; q < DQ; ++q) {
                    const float xv = sX[(EXACT || 2 * q + h < D) ? 2 * q + h : 0];   // row 0 = every row
                    xq[q] = (EXACT || 2 * q + h < D) ? xv : 0.0f;
                }
                sparse_gather_const<DQ>(sAb, xq, re0, re1, acc);
            } else {
                sparse_gather<false, DQ>(sAb, scol, sX, sD, D, re0, re1, h, acc);
            }
            sparse_combine<DQ>(acc, SA.rem, wsplit);
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                zraw[q] = acc[q];
                const float ph = sh.phi[2 * q + h];   // phi[32]: always in range
                acc[q] = (first && 2 * q + h < D) ? acc[q] * ph : 0.0f;
            }
            sparse_forward_rowlocal<DQ>(acc, sW1, sh.bias[0], D, H, li, h, first, sU1 + r * sH, sRn1 + r);
            }
        }
        SYNC();
        if constexpr (LOG)
            if (p.trace_gates && SA.wave_active && SA.first && h == 0) trace_row(iter, 0, SA.row, sU1 + SA.row * sH);
        // ======== layer 2: U2 ========
        if (SB.wave_active) {
            const bool first = SB.first;
            const int r = first ? SB.row : 0, re0 = SB.e0, re1 = SB.e1, nsplit = SB.nsplit, wsplit = SB.wsplit;
            float acc[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
            sparse_gather<!RS, HQ>(sAb, scol, sU1, sH, H, re0, re1, h, acc);
            sparse_combine<HQ>(acc, SB.rem, wsplit);
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = first ? acc[q] : 0.0f;
            sparse_forward_rowlocal<HQ>(acc, sW2, sh.bias[1], H, H, li, h, first, sU2 + r * sH, sRn2 + r);
        }
        SYNC_B();
        if constexpr (LOG)   // (node mode with the fused wave-0 chain: the rows of set B all sit in wave 0, which has just synchronised)
            if (p.trace_gates && SB.wave_active && SB.first && h == 0) trace_row(iter, 1, SB.row, sU2 + SB.row * sH);
        if constexpr (GRAPH) {
        // ======== graph mode: layer 3 in full (no ReLU), U3 ========
        if (SA.wave_active) {
            const bool first = SA.first;
            const int r = first ? SA.row : 0, re0 = SA.e0, re1 = SA.e1, nsplit = SA.nsplit, wsplit = SA.wsplit;
            float acc[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
            sparse_gather<true, HQ>(sAb, scol, sU2, sH, H, re0, re1, h, acc);
            sparse_combine<HQ>(acc, SA.rem, wsplit);
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = first ? acc[q] : 0.0f;
            sparse_forward_rowlocal<HQ>(acc, sW3, sh.bias[2], H, O, li, h, first, sU3 + r * sO, sRn3 + r);
        }
        SYNC();
        // ======== graph mode: per-layer max-pool over ALL n rows (models.py:283, 291, 300; first maximal row wins), head, dE ========
        {   // thread = pooled column (96 of them), rows scanned in order: no cross-lane reduction needed
            for (int col = tid; col < 96; col += NT) {
                const int l = col >> 5, c = col & 31;
                const float* arr = (l == 0) ? sU1 : (l == 1) ? sU2 : sU3;
                const int stride = (l == 2) ? sO : sH;
                float best = -3.0e38f;
                int barg = 0;
                if (c < ((l == 2) ? O : H)) {
                    for (int i = 0; i < n; ++i) {
                        float v = arr[i * stride + c];
                        if (l < 2) v = relu_(v);
                        if (v > best) {
                            best = v;
                            barg = i;
                        }
                    }
                } else {
                    best = 0.0f;
                }
                sh.e[col] = best;
                sh.erow[col] = barg;
                if constexpr (LOG)
                    if (p.trace_pool) p.trace_pool[((size_t)t * p.num_iters + iter) * 96 + col] = (c < ((l == 2) ? O : H)) ? barg : -1;
            }
        }
        SYNC();
        if (wave == 0) {  // softmax head (explain.py:710-711, 750-753): g = p - onehot(label), dE = Wp^T g
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
        }
        SYNC();
        // ======== graph mode: dZ3 (row-local backward of layer 3; dE3 lands on the arg-max rows), overwrites U3 ========
        if (SA.wave_active) {
            const bool first = SA.first;
            const int r = first ? SA.row : 0, re0 = SA.e0, re1 = SA.e1, nsplit = SA.nsplit, wsplit = SA.wsplit;
            float du[HQ], uu[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int c = 2 * q + h;
                const bool in = first && c < O;
                uu[q] = in ? sU3[r * sO + c] : 0.0f;
                du[q] = (in && sh.erow[64 + c] == r) ? sh.dEs[64 + c] : 0.0f;
            }
            const f32x16 c16 = sparse_backward_rowlocal<HQ, EXACT>(du, uu, first ? sRn3[r] : 1.0f, sW3, H, O, li, h);
            sparse_store_cols(c16, sU3 + r * sO, H, first, h);  // dZ3[r][.]
        }
        SYNC();
        // ======== graph mode: dX2 = Abar . dZ3 (+ dE2 on the arg-max rows) -> dZ2 ========
        if (SA.wave_active) {
            const bool first = SA.first;
            const int r = first ? SA.row : 0, re0 = SA.e0, re1 = SA.e1, nsplit = SA.nsplit, wsplit = SA.wsplit;
            float acc[HQ], uu[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
            sparse_gather<false, HQ>(sAb, scol, sU3, sO, H, re0, re1, h, acc);
            sparse_combine<HQ>(acc, SA.rem, wsplit);
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int c = 2 * q + h;
                const bool in = first && c < H;
                const float u = in ? sU2[r * sH + c] : 0.0f;
                float dx = acc[q];
                if (in && sh.erow[32 + c] == r) dx += sh.dEs[32 + c];
                acc[q] = (u > 0.0f) ? dx : 0.0f;
                uu[q] = u;
            }
            const f32x16 c16 = sparse_backward_rowlocal<HQ, EXACT>(acc, uu, first ? sRn2[r] : 1.0f, sW2, H, H, li, h);
            sparse_store_cols(c16, sdZ2w + r * sH, H, first, h);
        }
        SYNC();
        } else {
        // ======== row t of layer 3 (the only row the reference reads, explain.py:713), head, dE, dZ3[t] ========
        // Register form (wave 0 alone owns t and its neighbours - fuseB - and the class leaves room for its operands): the whole
        // chain z -> y = z W3 + b -> normalise -> logits -> softmax -> dE -> dY3 -> dZ3 stays in registers.  Lane c holds column c
        // of every vector; a vector is broadcast with v_readlane (SGPR operand of the FMA), reductions are DPP row sums; the
        // weights a lane needs (column / row c of W3, its three entries of every head row) are loaded at the START of the phase,
        // before anything that depends on this iteration, so their LDS latency hides behind the row-t gather.  The LDS form below
        // (round 2) staged every intermediate through LDS: six store -> wave sync -> load round trips plus 100 dependent loads,
        // 2.5 of the 12.5 us of an iteration of syn1's largest target.
        const bool reg_head = (NT != 1024) && fuseB;
        if (reg_head) {
            if (wave == 0) {
                // (class count as a compile-time bound of the unrolled loops: the reference's heads have 2 or 4 classes)
                auto head = [&](auto CHc) {
                constexpr int CH = decltype(CHc)::value;
                const int c = li;
                const int kc = (EXACT || c < H) ? c : 0;      // row of W3 this lane reads in the backward product (rows >= H do not exist)
                float w3[2 * HQ];
#pragma unroll
                for (int k = 0; k < 2 * HQ; ++k) w3[k] = sW3[((EXACT || k < H) ? k : 0) * 33 + c];     // W3[k][c]: forward, output column c
                const float b3 = sh.bias[2][c];
                const float e1 = (c < H) ? relu_(sU1[tr * sH + kc]) : 0.0f;    // relu(U1[t]), relu(U2[t]): layer 2 of this iteration is done
                const float e2 = (c < H) ? relu_(sU2[tr * sH + kc]) : 0.0f;
                // row t of Abar . relu(U2): the two half-lanes take alternate entries
                float z = 0.0f;
                for (int e = rt0 + h; e < rt1; e += 2) z = fmaf(sAb[e], relu_(sU2[(int)scol[e] * sH + kc]), z);
                z = (c < H) ? z : 0.0f;
                z = xor32_sum(z);
                const int zi = __builtin_bit_cast(int, z);
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll
                for (int k = 0; k < 2 * HQ; k += 2) {
                    const float za = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, k));
                    const float zb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(zi, k + 1));
                    y0 = fmaf((EXACT || k < H) ? za : 0.0f, w3[k], y0);
                    y1 = fmaf((EXACT || k + 1 < H) ? zb : 0.0f, w3[k + 1], y1);
                }
                // the head rows (three entries per lane and class) are fetched while the norm is reduced
                float wp[3][CH], bpv[CH];
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    const int cr = cc < C ? cc : 0;
#pragma unroll
                    for (int l = 0; l < 3; ++l) wp[l][cc] = sWp[cr * 96 + l * 32 + c];
                    bpv[cc] = sh.sbp[cr];
                }
                // Hardware forms (rcp_, sqrt_, exp_: gnnx_kernels.hpp) instead of the IEEE sequences on this single-wave chain - measured per part
                // (session q of round 4, syn1 launch / syn5 windows of the round-3 test beyond 1e-5 / windows with a differing decision):
                // IEEE head 3.015 ms / 12 / 5; softmax only (FSM: four exponentials, four divisions by their sum) 2.928 ms / 13 / 5; normalisations
                // only (FH: one square root, two divisions) 2.987 ms / 18 / 7; both 2.885 ms / 18 (worst 7.7e-3) / 5.  The softmax takes the
                // hardware forms (three quarters of the gain, the parity figures of the IEEE head); the normalisations keep the IEEE ones.
                constexpr bool FH = (SP_FAST_HEAD & 1) != 0;    // the two normalisations (sqrt, two divisions)
                constexpr bool FSM = (SP_FAST_HEAD & 2) != 0;   // the softmax (exponentials, the division by their sum)
                const float y = (c < O) ? y0 + y1 + b3 : 0.0f;
                const float ss = sum_lanes_0_31(y * y);
                const float rnorm = fmaxf(FH ? sqrt_(ss) : sqrtf(ss), 1e-12f);
                const float rinv3 = FH ? rcp_(rnorm) : 0.0f;
                const float u3 = FH ? y * rinv3 : y / rnorm;  // U3[t][c]
                // logits: three products per lane and class, one 32-lane sum per class
                float zl[CH];
                float mx = -3.0e38f;
                // (the CH lane sums step by step over all classes: the DPP steps / the cross-row shuffle of one sum depend on each other, those
                //  of different classes do not - same sums, same order within each)
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] = fmaf(wp[0][cc], e1, fmaf(wp[1][cc], e2, wp[2][cc] * u3));
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] += row_shl<8>(zl[cc]);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] += row_shl<4>(zl[cc]);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] += row_shl<2>(zl[cc]);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] += row_shl<1>(zl[cc]);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) zl[cc] = xor16_sum(zl[cc]);
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    zl[cc] = (cc < C) ? bcast_first(zl[cc]) + bpv[cc] : -3.0e38f;
                    mx = fmaxf(mx, zl[cc]);
                }
                float sum = 0.0f;
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    zl[cc] = (cc < C) ? (FSM ? exp_(zl[cc] - mx) : expf(zl[cc] - mx)) : 0.0f;
                    sum += zl[cc];
                }
                const float rsum = FSM ? rcp_(sum) : 0.0f;
                // g = p - onehot(y_gt) (explain.py:713-714, 750-753); dE = Wp^T g, the lane's entry of each of the three slices
                float dE1 = 0.0f, dE2 = 0.0f, dE3 = 0.0f;
#pragma unroll
                for (int cc = 0; cc < CH; ++cc) {
                    const float g = (cc < C) ? (FSM ? zl[cc] * rsum : zl[cc] / sum) - ((cc == tm.y_gt) ? 1.0f : 0.0f) : 0.0f;
                    if constexpr (LOG)
                        if (Lrow && lane == 0 && cc == tm.y_gt) Lrow[0] = -logf(zl[cc] / sum);   // explain.py:750-753
                    if constexpr (LOG)
                        if (Lrow && lane == 0 && cc < C && cc < LOGPN) Lrow[LOGP + cc] = zl[cc] / sum;   // explain.py:710-714, 157-158
                    dE1 = fmaf(wp[0][cc], g, dE1);
                    dE2 = fmaf(wp[1][cc], g, dE2);
                    dE3 = fmaf(wp[2][cc], g, dE3);
                }
                // the backward operand W3[c][k] takes the registers of the forward one; its latency hides behind the dY3 reduction
#pragma unroll
                for (int k = 0; k < 2 * HQ; ++k) w3[k] = sW3[kc * 33 + ((EXACT || k < O) ? k : 0)];
                const float du3 = (c < O) ? dE3 : 0.0f;
                const float sd = sum_lanes_0_31(du3 * u3);
                const float dy3 = FH ? (du3 - u3 * sd) * rinv3 : (du3 - u3 * sd) / rnorm;  // dY3[t][c]
                const int di = __builtin_bit_cast(int, dy3);
                float v0 = 0.0f, v1 = 0.0f;
#pragma unroll
                for (int k = 0; k < 2 * HQ; k += 2) {
                    const float da = __builtin_bit_cast(float, __builtin_amdgcn_readlane(di, k));
                    const float db = __builtin_bit_cast(float, __builtin_amdgcn_readlane(di, k + 1));
                    v0 = fmaf((EXACT || k < O) ? da : 0.0f, w3[k], v0);
                    v1 = fmaf((EXACT || k + 1 < O) ? db : 0.0f, w3[k + 1], v1);
                }
                if (h == 0) {
                    sh.dz3[c] = (c < H) ? v0 + v1 : 0.0f;
                    sh.dEs[c] = dE1;
                    sh.dEs[32 + c] = dE2;
                    sh.dEs[64 + c] = dE3;
                }
                };
                if (C <= 4) head(std::integral_constant<int, 4>{}); else head(std::integral_constant<int, RES_CMAX>{});
            }
            SYNC_B();
            if (pair_flag && !fuse_sync) SYNC();   // (the partner runs the LDS form of the head: one barrier more)
        } else {
        const int nwz = fuseB ? 1 : NW;  // waves that share row t's entries
        if (!fuseB || wave == 0) {   // row t of Abar . relu(U2): its entries dealt over the waves, lane = column; partials summed in wave order
            float z = 0.0f;
            if (li < H)
                for (int e = rt0 + 2 * wave + h; e < rt1; e += 2 * nwz)
                    z = fmaf(sAb[e], relu_(sU2[(int)scol[e] * sH + li]), z);
            z = xor32_sum(z);
            if (h == 0) sh.dfw[wave][li] = z;  // dfw is free until the layer-1 backward
        }
        SYNC_B();
        if (wave == 0) {  // the head is a chain of tiny dependent steps: one wave, wave-level syncs only
            const int c = li;
            float z = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) z += (w < nwz) ? sh.dfw[w][c] : 0.0f;
            // layer 3 for row t: y = z W3 + b3, normalised (z staged in LDS so the H products' loads are independent)
            if (h == 0) sh.z3[c] = z;
            wave_sync();
            float y = 0.0f;
            if (c < O) {
                float y0 = 0.0f, y1 = 0.0f;
#pragma unroll 4
                for (int k = 0; k + 1 < H; k += 2) {
                    y0 = fmaf(sh.z3[k], sW3[k * 33 + c], y0);
                    y1 = fmaf(sh.z3[k + 1], sW3[(k + 1) * 33 + c], y1);
                }
                if (H & 1) y0 = fmaf(sh.z3[H - 1], sW3[(H - 1) * 33 + c], y0);
                y = y0 + y1 + sh.bias[2][c];
            }
            const float ss = sum_lanes_0_31(y * y);  // lanes 0..31 and 32..63 hold the same y
            const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
            const float u3 = y / rnorm;  // U3[t][c] (both halves)
            if (h == 0) {
                sh.e[64 + c] = u3;
                sh.e[c] = (c < H) ? relu_(sU1[tr * sH + c]) : 0.0f;
                sh.e[32 + c] = (c < H) ? relu_(sU2[tr * sH + c]) : 0.0f;
            }
            wave_sync();
            {   // softmax head (explain.py:713-714, 750-753): g = p - onehot(y_gt); class = lane / 8, 12 terms per lane
                const int cls = lane >> 3, part = lane & 7;
                float s = 0.0f;
                if (cls < C)
                    for (int q = part * 12; q < part * 12 + 12; ++q) s = fmaf(sWp[cls * 96 + q], sh.e[q], s);
                s += row_shl<4>(s);  // class sums in lanes 0, 8, .., 56
                s += row_shl<2>(s);
                s += row_shl<1>(s);
                const float zc = __shfl(s, (lane & 7) * 8);
                const float zl = (lane < C) ? zc + sh.sbp[lane] : -3.0e38f;
                float mx = zl;  // max / sum over lanes 0..7 (C <= 8) end up in lane 0
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
            for (int part = 0; part < 2; ++part) {  // dE = Wp^T g (96 entries over 64 lanes)
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
            const float dy3 = (du3 - uq * s) / rnorm;  // dY3[t][c] in lanes c < 32
            if (h == 0) sh.y3[c] = dy3;
            wave_sync();
            float v = 0.0f;
            if (c < H) {
                float v0 = 0.0f, v1 = 0.0f;
#pragma unroll 4
                for (int c2 = 0; c2 + 1 < O; c2 += 2) {
                    v0 = fmaf(sh.y3[c2], sW3[c * 33 + c2], v0);
                    v1 = fmaf(sh.y3[c2 + 1], sW3[c * 33 + c2 + 1], v1);
                }
                if (O & 1) v0 = fmaf(sh.y3[O - 1], sW3[c * 33 + O - 1], v0);
                v = v0 + v1;
            }
            if (h == 0) sh.dz3[c] = (c < H) ? v : 0.0f;
        }
        SYNC_B();
        }
        // ======== dZ2 (rank-1: dX2[r] = Abar[r][t] dZ3[t] + dE2 on row t) and g3; dZ2 overwrites U2 row by row ========
        if (SB.wave_active) {
            const bool first = SB.first;
            const int r = first ? SB.row : 0, re0 = SB.e0, re1 = SB.e1, nsplit = SB.nsplit, wsplit = SB.wsplit;
            const float art = sArt[r];
            float du[HQ], uu[HQ];
            float gpart = 0.0f;
#pragma unroll
            for (int q = 0; q < HQ; ++q) {
                const int c = 2 * q + h;
                const float ul = (EXACT || c < H) ? sU2[r * sH + c] : 0.0f;   // r is a valid row for every lane
                const float u = (first && c < H) ? ul : 0.0f;
                const float dz = (c < H) ? sh.dz3[c] : 0.0f;
                gpart = fmaf(dz, relu_(u), gpart);
                float dx = art * dz;
                const float de = sh.dEs[32 + c];
                dx += (first && r == tr && c < H) ? de : 0.0f;
                du[q] = (u > 0.0f) ? dx : 0.0f;
                uu[q] = u;
            }
            gpart = xor32_sum(gpart);
            if (first && h == 0) sG3[r] = gpart;
            const f32x16 c16 = sparse_backward_rowlocal<HQ, EXACT>(du, uu, first ? sRn2[r] : 1.0f, sW2, H, H, li, h);
            sparse_store_cols(c16, sU2 + r * sH, H, first, h);  // dZ2[r][.]: every U2 value of this row is already in registers
        }
        SYNC();
        }
        const float* sdZ2 = sdZ2w;  // node mode: the U2 array (overwritten above); graph mode: its own array
        float log_phs = 0.0f;       // LOG form of the algebraic form: sum of phi of this iteration (wave 0), taken before the merged feature-mask step
        // ======== dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 ; feature-mask gradient partials ========
        {
            float dfq[DQ];
            float dvq[HQ];   // algebraic constant-feature form: this lane's share of sum_r s_r dY1[r]
#pragma unroll
            for (int q = 0; q < DQ; ++q) dfq[q] = 0.0f;
#pragma unroll
            for (int q = 0; q < HQ; ++q) dvq[q] = 0.0f;
            if (SA.wave_active) {
                const bool first = SA.first;
                const int r = first ? SA.row : 0, re0 = SA.e0, re1 = SA.e1, nsplit = SA.nsplit, wsplit = SA.wsplit;
                float acc[HQ], uu[HQ];
#pragma unroll
                for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
                if constexpr (GRAPH) {
                    sparse_gather<false, HQ>(sAb, scol, sdZ2, sH, H, re0, re1, h, acc);
                } else {
                    // dZ2 is non-zero only on t and its neighbours: of this slot's entries only those pointing there contribute
                    // (typically none or one - marked in SA.bmask at setup), instead of a walk over the whole row
                    for (unsigned m = SA.bmask; m; m &= m - 1u) {
                        const int e = re0 + __ffs((int)m) - 1;
                        const float a = sAb[e];
                        const float* br = sdZ2 + (int)scol[e] * sH + h;
#pragma unroll
                        for (int q = 0; q < HQ; ++q) acc[q] = fmaf(a, (2 * q + h < H) ? br[2 * q] : 0.0f, acc[q]);
                    }
                }
                sparse_combine<HQ>(acc, SA.rem, wsplit);
                const float rinv1 = RS ? rcp_(first ? sRn1[r] : 1.0f) : 0.0f;
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const int c = 2 * q + h;
                    // (RS: the row's own U1 - negative columns included, the normalisation's Jacobian needs them - exactly as layer 1 formed it)
                    const float ul = RS ? fmaf(sraw, sh.wt[c], sh.bias[0][c]) * rinv1 : ((EXACT || c < H) ? sU1[r * sH + c] : 0.0f);
                    const float u = (first && c < H) ? ul : 0.0f;
                    float dx = acc[q];
                    const float de = sh.dEs[c];
                    dx += (GRAPH ? (first && c < H && sh.erow[c] == r) : (first && r == tr && c < H)) ? de : 0.0f;
                    acc[q] = (u > 0.0f) ? dx : 0.0f;
                    uu[q] = u;
                }
                if constexpr (XC == 2) {
                    // dY1 of the row stays in the lane's registers (columns 2q + h); the layer-1 part of dL/dAbar is ONE number per row,
                    // ci = dY1 . wt, handed to the row's other slots through its slot in the (otherwise unused) dZ1 array; the feature-mask
                    // gradient needs sum_r s_r dY1[r]: this row's share goes into the reduction below
                    float sd2[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q < HQ; ++q) sd2[q & 1] = fmaf(acc[q], uu[q], sd2[q & 1]);
                    const float sdot = xor32_sum(sd2[0] + sd2[1]);
                    const float rinv = RS ? rinv1 : rcp_(first ? sRn1[r] : 1.0f);
                    float cp[2] = {0.0f, 0.0f};
#pragma unroll
                    for (int q = 0; q < HQ; ++q) {
                        const float dy = (acc[q] - uu[q] * sdot) * rinv;
                        cp[q & 1] = fmaf(dy, sh.wt[2 * q + h], cp[q & 1]);
                        dvq[q] = sraw * dy;     // (sraw is zero on lanes that own no row)
                    }
                    const float ci = xor32_sum(cp[0] + cp[1]);
                    float* sCi = sdZ1;          // one float per row
                    if (first && h == 0) sCi[r] = ci;
                    wave_sync();
                    const int ri = SA.row;
                    if (re0 < re1) {
                        const bool inB = SA.inB;
                        const float cr = sCi[ri];
                        if (!inB) {
                            for (int e = re0 + h; e < re1; e += 2) sGe[e] = cr;
                        } else {
                            float d2[2 * HQ];
#pragma unroll
                            for (int c = 0; c < 2 * HQ; ++c) d2[c] = sdZ2[ri * sH + c];
                            for (int e = re0 + h; e < re1; e += 4) {
                                const bool two = e + 2 < re1;
                                const int j0 = scol[e], j1 = scol[two ? e + 2 : e];
                                const float* u0 = sU1 + j0 * sH;
                                const float* u1 = sU1 + j1 * sH;
                                float a0 = cr, a1 = 0.0f, b0 = cr, b1 = 0.0f;
#pragma unroll
                                for (int c = 0; c < 2 * HQ; c += 2) {
                                    a0 = fmaf(d2[c], RS ? u0[c] : relu_(u0[c]), a0);
                                    a1 = fmaf(d2[c + 1], RS ? u0[c + 1] : relu_(u0[c + 1]), a1);
                                    b0 = fmaf(d2[c], RS ? u1[c] : relu_(u1[c]), b0);
                                    b1 = fmaf(d2[c + 1], RS ? u1[c + 1] : relu_(u1[c + 1]), b1);
                                }
                                sGe[e] = a0 + a1;
                                if (two) sGe[e + 2] = b0 + b1;
                            }
                        }
                    }
                } else {
                const f32x16 c16 = sparse_backward_rowlocal<HQ, EXACT>(acc, uu, first ? sRn1[r] : 1.0f, sW1, D, H, li, h);
                sparse_store_cols(c16, sdZ1 + r * sD, D, first, h);
                wave_sync();  // the other half-lane of this row wrote the columns this lane reads next
#pragma unroll
                for (int q = 0; q < DQ; ++q)
                    {
                        const float dzl = (EXACT || 2 * q + h < D) ? sdZ1[r * sD + 2 * q + h] : 0.0f;
                        dfq[q] = (first && 2 * q + h < D) ? dzl * zraw[q] : 0.0f;
                    }
                if constexpr (!GRAPH) {
                    // dL/dAbar on this slot's entries, row side: G[i][j] = dZ1[i] . (X[j] * phi) + dZ2[i] . relu(U1[j]).  Every slot
                    // of the row (not only its first) takes its own entries; the two half-lanes take alternate entries with ALL
                    // columns, so nothing is reduced across lanes.  The edge owners then read two floats per edge instead of
                    // 40 - 120 (the rows of both endpoints): the work sits on the lanes that already own the rows within two hops.
                    const int ri = SA.row;
                    if (re0 < re1) {
                        const bool inB = SA.inB;
                        float dz[2 * DQ], d2[2 * HQ];
#pragma unroll
                        for (int c = 0; c < 2 * DQ; ++c) dz[c] = (c < D) ? sdZ1[ri * sD + c] * sh.phi[c] : 0.0f;
#pragma unroll
                        for (int c = 0; c < 2 * HQ; ++c) {
                            const float dl = (EXACT || (inB && c < H)) ? sdZ2[ri * sH + c] : 0.0f;
                            d2[c] = (inB && c < H) ? dl : 0.0f;
                        }
                        if constexpr (XC) {
                            // constant feature rows: dZ1[i] . (X[j] * phi) is the same number for every entry of the row - formed once,
                            // with the accumulation order of the general path (bit-identical); entries of rows beyond t's neighbours
                            // (no dZ2 part) are a stream of stores, the others continue the two sums with the dZ2 . relu(U1[j]) terms
                            float c0 = 0.0f, c1 = 0.0f;
#pragma unroll
                            for (int c = 0; c < 2 * DQ; c += 2) {
                                c0 = fmaf(dz[c], sX[c], c0);
                                c1 = fmaf(dz[c + 1], sX[c + 1], c1);
                            }
                            if (!inB) {
                                const float g = c0 + c1;
                                for (int e = re0 + h; e < re1; e += 2) sGe[e] = g;
                            } else {
                                for (int e = re0 + h; e < re1; e += 4) {
                                    const bool two = e + 2 < re1;
                                    const int j0 = scol[e], j1 = scol[two ? e + 2 : e];
                                    const float* u0 = sU1 + j0 * sH;
                                    const float* u1 = sU1 + j1 * sH;
                                    float a0 = c0, a1 = c1, b0 = c0, b1 = c1;
#pragma unroll
                                    for (int c = 0; c < 2 * HQ; c += 2) {
                                        a0 = fmaf(d2[c], relu_(u0[c]), a0);
                                        a1 = fmaf(d2[c + 1], relu_(u0[c + 1]), a1);
                                        b0 = fmaf(d2[c], relu_(u1[c]), b0);
                                        b1 = fmaf(d2[c + 1], relu_(u1[c + 1]), b1);
                                    }
                                    sGe[e] = a0 + a1;
                                    if (two) sGe[e + 2] = b0 + b1;
                                }
                            }
                        } else
                        // two entries per trip (e, e + 2: this half-lane's next two), two accumulators per entry: the loads of
                        // both entries are in flight together and no FMA chain is longer than half a row
                        for (int e = re0 + h; e < re1; e += 4) {
                            const bool two = e + 2 < re1;
                            const int j0 = scol[e], j1 = scol[two ? e + 2 : e];
                            const float* x0 = sX + j0 * sD;
                            const float* x1 = sX + j1 * sD;
                            float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
#pragma unroll
                            for (int c = 0; c < 2 * DQ; c += 2) {
                                a0 = fmaf(dz[c], x0[c], a0);
                                a1 = fmaf(dz[c + 1], x0[c + 1], a1);
                                b0 = fmaf(dz[c], x1[c], b0);
                                b1 = fmaf(dz[c + 1], x1[c + 1], b1);
                            }
                            if (inB) {
                                const float* u0 = sU1 + j0 * sH;
                                const float* u1 = sU1 + j1 * sH;
#pragma unroll
                                for (int c = 0; c < 2 * HQ; c += 2) {
                                    a0 = fmaf(d2[c], relu_(u0[c]), a0);
                                    a1 = fmaf(d2[c + 1], relu_(u0[c + 1]), a1);
                                    b0 = fmaf(d2[c], relu_(u1[c]), b0);
                                    b1 = fmaf(d2[c + 1], relu_(u1[c + 1]), b1);
                                }
                            }
                            sGe[e] = a0 + a1;
                            if (two) sGe[e + 2] = b0 + b1;
                        }
                    }
                }
                }
            }
            // colsum(dZ1 * Zraw): over the 16 lanes of a DPP row (row shifts), the two rows of a half (one shuffle), then
            // over the waves in fixed order   (algebraic form: sum over the rows of s_r dY1[r], H columns instead of D)
            if constexpr (XC == 2) {
                // (DFWR: the two 16-lane rows of a half store their sums side by side and the reader adds them - the same sum, r0 + r1, without
                // ten cross-row shuffles through the LDS crossbar on every wave's way to the barrier; classes of up to 8 waves: dfw has 16 rows)
                constexpr bool DFWR = NW <= 8;
                float* dfw2 = &sh.dfw[0][0];
                // (step by step over ALL columns: the DPP steps of one column depend on each other - two wait states each - those of different
                //  columns do not)
#pragma unroll
                for (int q = 0; q < HQ; ++q) dvq[q] += row_shl<8>(dvq[q]);
#pragma unroll
                for (int q = 0; q < HQ; ++q) dvq[q] += row_shl<4>(dvq[q]);
#pragma unroll
                for (int q = 0; q < HQ; ++q) dvq[q] += row_shl<2>(dvq[q]);
#pragma unroll
                for (int q = 0; q < HQ; ++q) dvq[q] += row_shl<1>(dvq[q]);
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    float v = dvq[q];
                    if constexpr (DFWR) {
                        if ((li & 15) == 0) dfw2[(2 * wave + (li >> 4)) * 32 + 2 * q + h] = v;
                    } else {
                        v = xor16_sum(v);
                        if (li == 0) sh.dfw[wave][2 * q + h] = v;
                    }
                }
            } else
            {
#pragma unroll
                for (int q = 0; q < DQ; ++q) dfq[q] += row_shl<8>(dfq[q]);
#pragma unroll
                for (int q = 0; q < DQ; ++q) dfq[q] += row_shl<4>(dfq[q]);
#pragma unroll
                for (int q = 0; q < DQ; ++q) dfq[q] += row_shl<2>(dfq[q]);
#pragma unroll
                for (int q = 0; q < DQ; ++q) dfq[q] += row_shl<1>(dfq[q]);
#pragma unroll
                for (int q = 0; q < DQ; ++q) dfq[q] = xor16_sum(dfq[q]);
#pragma unroll
                for (int q = 0; q < DQ; ++q)
                    if (li == 0) sh.dfw[wave][2 * q + h] = dfq[q];
            }
        }
        SYNC();
        if constexpr (XC == 2) {
            if (wave == 0) {   // dL/dphi[k] = x_k W1[k] . vsum, vsum = the waves' partial sums in wave order
                constexpr bool DFWR = NW <= 8;
                const float* dfw2 = &sh.dfw[0][0];
                // lane c holds vsum[c]; the product reads it with v_readlane (an SGPR operand of the FMA) instead of a store -> wave sync ->
                // load round trip, and lane k's row of W1 is fetched together with the partial sums; same FMA chain, same order
                const int kr = (lane < D) ? lane : 0;
                float w1r[2 * HQ];
#pragma unroll
                for (int c = 0; c < 2 * HQ; ++c) w1r[c] = sW1[kr * 33 + c];
                float v = 0.0f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += DFWR ? dfw2[(2 * w) * 32 + li] + dfw2[(2 * w + 1) * 32 + li] : sh.dfw[w][li];
                const int vi = __builtin_bit_cast(int, (li < H) ? v : 0.0f);
                float a = 0.0f;
#pragma unroll
                for (int c = 0; c < 2 * HQ; ++c) a = fmaf(w1r[c], __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, c)), a);
                if (lane < D) sh.dfp[lane] = sX[lane] * a;
                if constexpr (LOG) log_phs = sum_lanes_0_31((lane < D) ? sh.phi[lane] : 0.0f);   // (phi of THIS iteration: the merged step below replaces it)
                if constexpr (MP) {   // (dfp[k] is read back by the thread that wrote it; phi of all D columns before wt)
                    if (lane < D) feature_mask_step(lane);
                    wave_sync();
                    update_wt();
                }
            }
        } else
        if (tid < D) {
            float s = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += sh.dfw[w][tid];
            sh.dfp[tid] = s;
        }
        // ======== per owned edge: G_ij + G_ji, regulariser gradients, Adam on both directed entries ========
        float ls_size = 0.0f, ls_ent = 0.0f, ls_lap = 0.0f, ls_den = 0.0f, ls_adj = 0.0f;   // LOG form: this thread's part of the logged sums (its owned edges, both directions)
        auto edge_phase = [&](auto ADAMc) {
        constexpr bool ADAM = decltype(ADAMc)::value;
#pragma unroll
        for (int q = 0; q < SP_QMAX; ++q)
            if (edge_k(q) < eup) {
                const int i = (int)(npk[q] & 0xffffu), j = (int)(npk[q] >> 16);
                if constexpr (LOG) {   // explain.py:755-770, 780-793 on the current iterate (before its update)
                    const float Sa = Sij[q], Sb = Sji[q];
                    ls_size += Sa + Sb;
                    ls_ent += (-Sa * logf(Sa) - (1.0f - Sa) * logf(1.0f - Sa)) + (-Sb * logf(Sb) - (1.0f - Sb) * logf(1.0f - Sb));
                    const float dyl = sYhat[i] - sYhat[j];
                    ls_lap += wgt[q] * (0.5f * (Sa + Sb)) * dyl * dyl;   // yhat^T (D - Abar) yhat = sum over undirected edges of Abar_ij (yhat_i - yhat_j)^2
                }
                // compile-time trip counts: all loads of an edge are issued before the first use; columns beyond
                // D / H are read from the row padding / the next row and dropped by the select
                float G0 = 0.0f, G1 = 0.0f;
                if constexpr (!GRAPH) {
                    G0 = sGe[epk[q] & 0xffffu];   // row-side products of both directions, formed by the layer-1 backward
                    G1 = sGe[epk[q] >> 16];
                } else {
                const int fl = eflag[q];
                // columns per load group (one LDS round trip each): the reference's widths take the whole dZ1 . X product in one
                // group and the dZ2 . relu(U1) product in two
                constexpr int EC0 = (DQ <= 8) ? 2 * DQ : DQ, EC1 = (HQ <= 10) ? HQ : HQ / 2;
#pragma unroll 1
                for (int c0 = 0; (fl & 3) && c0 < 2 * DQ; c0 += EC0) {
#pragma unroll
                    for (int cc = 0; cc < EC0; ++cc) {
                        const int c = c0 + cc;
                        const float t1 = fmaf(sdZ1[i * sD + c], sX[j * sD + c], sdZ1[j * sD + c] * sX[i * sD + c]) * sh.phi[c];
                        G0 += (c < D) ? t1 : 0.0f;
                    }
                }
#pragma unroll 1
                for (int c0 = 0; (fl & 12) && c0 < 2 * HQ; c0 += EC1) {
#pragma unroll
                    for (int cc = 0; cc < EC1; ++cc) {
                        const int c = c0 + cc;
                        const float t2 = fmaf(sdZ2[i * sH + c], relu_(sU1[j * sH + c]),
                                              sdZ2[j * sH + c] * relu_(sU1[i * sH + c]));
                        G1 += (c < H) ? t2 : 0.0f;
                    }
                }
                }
                float G = G0 + G1;
                if constexpr (GRAPH) {
                    float G2 = 0.0f;  // layer 3: dZ3 . relu(U2)
#pragma unroll 1
                    for (int c0 = 0; c0 < 2 * HQ; c0 += 2 * HQ / 4) {
#pragma unroll
                        for (int cc = 0; cc < 2 * HQ / 4; ++cc) {
                            const int c = c0 + cc;
                            const float t3 = fmaf(sU3[i * sO + c], relu_(sU2[j * sH + c]), sU3[j * sO + c] * relu_(sU2[i * sH + c]));
                            G2 += (c < H) ? t3 : 0.0f;
                        }
                    }
                    G += G2;
                } else {
                    // (unconditional loads - i, j < ld always - and selects: a lane-varying condition around a load is an exec-mask region)
                    const float g3j = sG3[j], g3i = sG3[i];
                    G += (i == tr) ? g3j : 0.0f;
                    G += (j == tr) ? g3i : 0.0f;
                }
                const float dy = sYhat[i] - sYhat[j];
                const float gc = (0.5f * G + p.c_lap * 0.5f * dy * dy * inv_n2) * wgt[q];
                {
                    const float S = Sij[q];
                    const float g = (gc + p.c_size - p.c_ent * Mij[q] * inv_n2) * S * (1.0f - S);
                    adam_update<ADAM, true>(Mij[q], mij[q], vij[q], g, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt, rbc2);
                }
                {
                    const float S = Sji[q];
                    const float g = (gc + p.c_size - p.c_ent * Mji[q] * inv_n2) * S * (1.0f - S);
                    adam_update<ADAM, true>(Mji[q], mji[q], vji[q], g, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt, rbc2);
                }
                if constexpr (LOG) {   // ExplainModule.mask_density (explain.py:680-683), taken after optimizer.step() (:142-148): the UPDATED entries
                    ls_den += wgt[q] * (0.5f * (sigmoidf_(Mij[q]) + sigmoidf_(Mji[q])));
                    ls_adj += wgt[q];
                }
                if constexpr (MP) {
                    // publish in place (publish_abar's body for this edge): nothing reads sAb / sArt between the barrier behind the layer-1
                    // backward and the next iteration's layer 1.  Not after the last iteration: the returned mask is the one of the LAST forward.
                    if (iter + 1 < p.num_iters) {
                        Sij[q] = sigmoidf_(Mij[q]);
                        Sji[q] = sigmoidf_(Mji[q]);
                        const float a = wgt[q] * (0.5f * (Sij[q] + Sji[q]));
                        sAb[epk[q] & 0xffffu] = a;
                        sAb[epk[q] >> 16] = a;
                        if (i == tr) sArt[j] = a;
                        if (j == tr) sArt[i] = a;
                    }
                }
            }
        };
        if (p.opt == 0) edge_phase(std::true_type{}); else edge_phase(std::false_type{});   // one branch around the loop, not one per update
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
        SYNC();  // dfp complete; every reader of sAb / sArt of this iteration is done
        if constexpr (LOG) {
            if (Lrow && wave == 0) {   // the entries off the edges were added to [1] and [3] by k_dead_entries before this launch
                const float phs = (XC == 2) ? log_phs : sum_lanes_0_31((lane < D) ? sh.phi[lane] : 0.0f);   // (phi of this iteration: its update below is behind the wave sync)
                float a = 0.0f, b = 0.0f, c = 0.0f, den = 0.0f, adj = 0.0f;
                for (int w = 0; w < NW; ++w) {
                    a += sh.lsum[w][0];
                    b += sh.lsum[w][1];
                    c += sh.lsum[w][2];
                    den += sh.lsum[w][3];
                    adj += sh.lsum[w][4];
                }
                if (lane == 0) {
                    Lrow[LOGD] = den / adj;
                    Lrow[1] += p.c_size * a;
                    Lrow[2] = GRAPH ? 0.0f : p.c_lap * c * inv_n2;
                    Lrow[3] += p.c_ent * b * inv_n2;
                    Lrow[4] = p.c_feat_size * phs / (float)D;
                }
                wave_sync();
            }
        }
        if constexpr (!MP) {
        if (tid < D) feature_mask_step(tid);
        if constexpr (XC == 2) {
            if (wave == 0) wave_sync();   // phi of all D columns (threads of wave 0) before wt is refreshed; publish_abar's barrier publishes wt
            update_wt();
        }
        if (iter + 1 < p.num_iters) publish_abar();  // the returned mask is the one of the LAST forward (explain.py:209-211)
        }
    }
    SYNC();
    // ---------------- results: dense Abar block (zero off the edges), M on the edges, feature mask ----------------
    {
        f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!p.edge_only)   // (gnnx_hyper.edge_results_only: the caller reads Abar on the edges only - skip the ld^2 zero-fill)
            for (int e = tid * 4; e < ld * ld; e += 4 * NT) *reinterpret_cast<f32x4*>(p.Abar + tm.offQ + e) = z4;
    }
    __threadfence_block();
    SYNC();
#pragma unroll
    for (int q = 0; q < SP_QMAX; ++q)
        if (edge_k(q) < eup) {
            const int i = (int)(npk[q] & 0xffffu), j = (int)(npk[q] >> 16);
            const float a = sAb[epk[q] & 0xffffu];
            p.Abar[tm.offQ + (size_t)i * ld + j] = a;
            p.Abar[tm.offQ + (size_t)j * ld + i] = a;
            Mg[(size_t)i * ld + j] = Mij[q];
            Mg[(size_t)j * ld + i] = Mji[q];
            if (p.m_out) {
                p.m_out[tm.offQ + (size_t)i * ld + j] = mij[q];
                p.m_out[tm.offQ + (size_t)j * ld + i] = mji[q];
            }
            if (p.v_out) {
                p.v_out[tm.offQ + (size_t)i * ld + j] = vij[q];
                p.v_out[tm.offQ + (size_t)j * ld + i] = vji[q];
            }
        }
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? sh.fcur[tid] : 0.0f;
    if (p.fs_out && tid < FS) {
        float* fs = p.fs_out + (size_t)t * 3 * FS + tid;
        fs[0] = (tid < D) ? sh.fcur[tid] : 0.0f;
        fs[FS] = (tid < D) ? sh.mf[tid] : 0.0f;
        fs[2 * FS] = (tid < D) ? sh.vf[tid] : 0.0f;
    }
}


// second launch bound = waves per SIMD the register allocation must leave room for: two 256-thread workgroups (or six
// 64-thread ones) per CU need 2; without it the 256-thread graph-mode build took 266 registers and ran one per CU
template <int DQ, int HQ, bool GRAPH, int NT, int XC = 0, bool LOG = false, bool EX = true>
__global__ __launch_bounds__(NT, NT >= 1024 ? 4 : 2) void k_sparse_resident(Params p, const int32_t* targets, const float* adam_tab) {
    static_assert(!(XC && GRAPH) && !(XC == 1 && LOG), "constant-feature forms: node mode only; logging forms exist of the general and of the algebraic form");
    __shared__ float pool[sp_pool_floats(NT)];
    __shared__ SparseFixed sh;
    sparse_resident_body<DQ, HQ, GRAPH, NT, XC, LOG, EX>(p, targets[blockIdx.x], adam_tab, pool, sh, (int)threadIdx.x);
}

// One launch for a node-mode batch of larger targets (512-thread class) and single-tile targets (64-thread code path):
// workgroups [0, n_big) take one larger target each, the others sp_mix_tiny() single-tile targets each - one per wave, in a
// slice of the same LDS pool, the model block shared by the workgroup (with a private copy of its 8 KB per target only six
// fit).  (Separate launches of the two groups overlap only partly: measured on syn1, the single-tile launch took 6.6 ms
// beside the big one against 3.8 ms alone.)
__host__ __device__ inline int sp_model_floats(int D, int H, int C) { return (D + 2 * H) * 33 + C * 96; }
__host__ __device__ inline int sp_fixed_floats() { return (int)((sizeof(SparseFixed) + 3) / 4); }
// single-tile targets per workgroup: eight (one per wave) when eight slices (a 64-thread pool minus the shared model block)
// and their SparseFixed blocks fit the 512-thread pool, else as many as fit (at least five)
__host__ __device__ inline int sp_mix_tiny(int D, int H, int C) {
    const int wsz = sp_model_floats(D, H, C);
    for (int k = 8; k > 5; --k)
        if (wsz + k * (sp_pool_floats(64) - wsz) + k * sp_fixed_floats() <= sp_pool_floats(512)) return k;
    return 5;
}
// Pair workgroups (round 5): a target of the 256-thread class (n <= 128, <= 512 edges, 72 KB of LDS) used to take a whole 512-thread workgroup
// - a whole CU for the 2.4 ms of its chain - whenever its batch also held larger targets, because a SEPARATE 256-thread launch does not pack:
// the dispatcher spreads its workgroups one per CU, where each blocks a 153 KB workgroup of the other launch (measured in round 4:
// GNNX_KEEP_256=1 on syn1 147.9 k vs 183.9 k nodes/s).  Here two such targets share one 512-thread workgroup: threads [0, 256) run one body,
// [256, 512) the other, each in its half of the pool.  s_barrier is workgroup-wide, and the bodies are the same code with the same barrier
// sequence (see sparse_resident_body: pair_flag), so every __syncthreads() is simply a barrier for both - the pair moves in lockstep,
// phase by phase, and a CU holds two targets.  On syn1 60 of the 101 larger targets qualify: 141 -> 111 workgroups per batch.
#ifndef GNNX_MIXED_NUM_VGPR      // (see GNNX_NUM_VGPR_ATTR in gnnx_kernels.hpp; 0 = uncapped, 256 registers per lane)
#define GNNX_MIXED_NUM_VGPR 112
#endif
#if GNNX_MIXED_NUM_VGPR
#define GNNX_MIXED_ATTR GNNX_NUM_VGPR_ATTR(GNNX_MIXED_NUM_VGPR)
#else
#define GNNX_MIXED_ATTR
#endif
template <int DQ, int HQ, int XC = 0, bool LOG = false, bool EX = true>
__global__ __launch_bounds__(512) GNNX_MIXED_ATTR void k_sparse_resident_mixed(Params p, const int32_t* big_ids, int n_big, const int32_t* tiny_ids,
                                                               int n_tiny, const float* adam_tab, int per_wg, int wsz,
                                                               const int32_t* pair_ids = nullptr, int n_pair = 0) {
    __shared__ float pool[sp_pool_floats(512)];
    __shared__ SparseFixed sh_big;
    __shared__ int pair_flag;
    static_assert(5 * sp_pool_floats(64) + 5 * (int)((sizeof(SparseFixed) + 3) / 4) <= sp_pool_floats(512),
                  "five single-tile slices and their SparseFixed blocks must fit the 512-thread pool whatever the model");
    static_assert(2 * sp_pool_floats(256) + (int)((sizeof(SparseFixed) + 3) / 4) <= sp_pool_floats(512),
                  "two 256-thread pools and the second body's SparseFixed block must fit the 512-thread pool");
    if ((int)blockIdx.x < n_big) {
        sparse_resident_body<DQ, HQ, false, 512, XC, LOG, EX>(p, big_ids[blockIdx.x], adam_tab, pool, sh_big, (int)threadIdx.x);
        return;
    }
#ifdef GNNX_NO_PAIR_BODY      // (measurement knob, tools/build_variants.sh: the kernel without the pair body - what its presence costs the other two)
    const int n_pair_wg = 0;
    if (false) {
#else
    const int n_pair_wg = (n_pair + 1) >> 1;
    if ((int)blockIdx.x < n_big + n_pair_wg) {
#endif
        const int half = (int)threadIdx.x >> 8;
        const int idx = 2 * ((int)blockIdx.x - n_big) + half;
        if (threadIdx.x == 0) pair_flag = 1;      // (read behind the setup's barriers)
        if (idx >= n_pair) return;                // an odd target count: the last workgroup's second half leaves (whole waves; s_barrier waits on the surviving ones)
        // the second body's wave 0 - the wave that runs the single-wave phases - sits on another SIMD than the first body's
        // (hardware waves 0 and 4 share SIMD 0): its threads are rotated by one wave
        const int t256 = (int)threadIdx.x & 255;
        const int tid = half ? ((t256 + 192) & 255) : t256;      // hardware wave 5 -> body wave 0
        SparseFixed* shp = half ? reinterpret_cast<SparseFixed*>(pool + 2 * sp_pool_floats(256)) : &sh_big;
        sparse_resident_body<DQ, HQ, false, 256, XC, LOG, EX>(p, pair_ids[idx], adam_tab, pool + half * sp_pool_floats(256), *shp, tid, nullptr,
                                                          &pair_flag);
        return;
    }
    // per_wg = sp_mix_tiny(D, H, C), wsz = sp_model_floats(D, H, C) from the host (reading a field of p here makes the compiler
    // pass the by-value Params of the bodies through scratch: 416 bytes per lane)
    const int slice = sp_pool_floats(64) - wsz;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = ((int)blockIdx.x - n_big - n_pair_wg) * per_wg + wave;
    if (wave >= per_wg || idx >= n_tiny) return;  // whole waves leave: the 64-thread body has no workgroup barrier
    SparseFixed* shp = reinterpret_cast<SparseFixed*>(pool + wsz + per_wg * slice) + wave;
    sparse_resident_body<DQ, HQ, false, 64, XC, LOG, EX>(p, tiny_ids[idx], adam_tab, pool + wsz + wave * slice, *shp, lane, pool);
}

// Packed single-wave targets (round 6: VERDICT r5 item 2, "sixteen one-wave chains per CU").  A single-wave target's iteration is a latency
// chain that issues a fifth of one SIMD's cycles; what bounds a saturated batch is how many chains a compute unit holds.  In the mixed launch
// (and in the 64-thread class's own launch) that was 8 (6): 188-218 registers per lane and 20 KB of LDS + a 5.4 KB SparseFixed each.  Here:
//   * the slim LDS form (sparse_layout: slim, SparseFixedSlim): <= TINY_SLICE floats per target, the model block shared by the workgroup;
//   * ONE workgroup per compute unit, one target per wave: 16 waves (1024 threads, 128 registers per lane) or 12 (768 threads, 168 registers).
//     A whole-CU workgroup on purpose: measured with 8-wave workgroups two to a CU (80 KB each) - alone 1.27x the class's own launch (syn4),
//     but BESIDE the mixed launch's whole-CU workgroups the dispatcher spreads them one per CU, where each blocks a 158 KB workgroup of the
//     other launch: syn1 301.7 k -> 151-183 k nodes/s (profiles/r06_tiny_pack_ab.txt).  With whole-CU workgroups in both launches a free
//     compute unit takes either;
//   * exact widths of the reference's node encoder (D = 10, H = O = 20), C <= 4, the algebraic constant-feature form (XC = 2), no logging:
//     everything else keeps the classes above.  Same body, same arithmetic, same order: bit-identical to them (tests/test_tiny_pack.py).
// Which targets qualify is decided by the plan (gnnx_capi.hip: tiny_pack_fits); the rest of the 64-thread class runs where it ran.
constexpr int TINY_W = (10 + 2 * 20) * 33 + 4 * 96;     // the shared model block: W1 | W2 | W3 (33-float rows) | four head rows
__host__ __device__ constexpr int tiny_nwg(int per_cu) { return per_cu; }      // targets (= waves) per workgroup
// floats per target (pool slice + SparseFixedSlim): the workgroup takes what the mixed launch's workgroups take (158.4 KB: the prepare
// stage's service kernels keep their 5 KB of LDS beside it)
__host__ __device__ constexpr int tiny_slice(int per_cu) {
    return (GNNX_POOL512_FLOATS + (int)((sizeof(SparseFixed) + 3) / 4) - TINY_W) / per_cu;
}
__host__ __device__ constexpr int tiny_fixed_floats() { return (int)((sizeof(SparseFixedSlim) + 3) / 4); }
__host__ __device__ constexpr int tiny_pool_floats(int per_cu) { return tiny_slice(per_cu) - tiny_fixed_floats(); }
#define GNNX_TINY_KERNEL(NAME, PER_CU, NUM_VGPR)                                                                                              \
    template <int DQ, int HQ, int XC>                                                                                                         \
    __global__ __launch_bounds__(64 * PER_CU) GNNX_NUM_VGPR_ATTR(NUM_VGPR) void NAME(Params p, const int32_t* ids, int n_ids,               \
                                                                                     const float* adam_tab) {                                 \
        constexpr int NWG = PER_CU, SLICE = tiny_slice(PER_CU), POOL = tiny_pool_floats(PER_CU);                                              \
        __shared__ float lds[TINY_W + NWG * SLICE];                                                                                           \
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                                                                           \
        const int idx = (int)blockIdx.x * NWG + wave;                                                                                         \
        if (idx >= n_ids) return;   /* whole waves leave: the single-wave body has no workgroup barrier */                                   \
        float* mine = lds + TINY_W + wave * SLICE;                                                                                            \
        SparseFixedSlim* shp = reinterpret_cast<SparseFixedSlim*>(mine + POOL);                                                               \
        sparse_resident_body<DQ, HQ, false, 64, XC, false, true, SparseFixedSlim>(p, ids[idx], adam_tab, mine, *shp, lane, lds, nullptr, POOL); \
    }
GNNX_TINY_KERNEL(k_sparse_resident_tiny16, 16, 64)
GNNX_TINY_KERNEL(k_sparse_resident_tiny12, 12, 84)
#undef GNNX_TINY_KERNEL
static_assert((TINY_W + 16 * tiny_slice(16)) * 4 <= 160 * 1024 - 5 * 1024 && (TINY_W + 12 * tiny_slice(12)) * 4 <= 160 * 1024 - 5 * 1024, "LDS of a compute unit, less the service kernels' share");

// per target: directed off-diagonal non-zeros of its block of the packed adjacency and the row slots the sparse
// resident kernel would need (-1: more than SP_LD_MAX rows) -> out[2 t], out[2 t + 1]   (gnnx_plan_analyze)
// X (may be null) -> xconst[t] = 1 when every feature row of the target equals its first row bit for bit (all FS columns: the
// padding columns are zero in every row), else 0   (gnnx_plan_analyze_features: selects the constant-feature form of the kernel)
// rowdeg / rowcnt / ecount (gnnx_pack_csr_analyze: the packing kernel has just counted every row): the degrees are READ instead of
// rescanned from the dense block, and wave 0 turns the rows' upper-triangle counts into row starts + the target's edge count (what
// k_edge_rowscan does as a launch of its own) - one per-target launch for the routing figures and the edge layout together.
__global__ __launch_bounds__(256) GNNX_SERVICE_ATTR void k_count_edges(const TargetMeta* meta, const float* A, int32_t* out, const float* X = nullptr,
                                                     int32_t* xconst = nullptr, const int32_t* rowdeg = nullptr, int32_t* rowcnt = nullptr,
                                                     int64_t* ecount = nullptr) {
    __shared__ int deg[SP_LD_MAX];
    __shared__ int part[4];
    __shared__ int differs;
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool small = tm.ld <= SP_LD_MAX;
    if (rowcnt && wave == 0) {   // exclusive scan of the rows' upper-triangle counts, in place (k_edge_rowscan)
        int32_t* rc = rowcnt + tm.offR;
        int carry = 0;
        for (int r0 = 0; r0 < tm.n; r0 += 64) {
            const int r = r0 + lane;
            const int v = (r < tm.n) ? rc[r] : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int x = __shfl(incl, lane - d);
                if (lane >= d) incl += x;
            }
            if (r < tm.n) rc[r] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
        if (lane == 0 && ecount) ecount[blockIdx.x] = carry;
    }
    if (!small) {  // cannot take the sparse kernel whatever its edge count: skip the scan
        if (tid == 0) {
            out[2 * blockIdx.x] = -1;
            out[2 * blockIdx.x + 1] = -1;
            if (xconst) xconst[blockIdx.x] = 0;
        }
        return;
    }
    if (xconst) {
        if (tid == 0) differs = 0;
        __syncthreads();
        bool same = X != nullptr;
        if (X)
            for (int e = tid; e < tm.n * FS; e += 256)
                same &= __float_as_uint(X[(size_t)tm.offR * FS + e]) == __float_as_uint(X[(size_t)tm.offR * FS + (e & (FS - 1))]);
        if (!same) differs = 1;  // benign race: every writer stores 1
        __syncthreads();
        if (tid == 0) xconst[blockIdx.x] = differs ? 0 : 1;
    }
    int cnt = 0;
    if (rowdeg) {   // one row per thread; the wave's rows summed with a butterfly
        int d = 0;
        for (int r = tid; r < tm.n; r += 256) {
            const int dr = rowdeg[tm.offR + r];
            deg[r] = dr;
            d += dr;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) d += __shfl_xor(d, o);
        cnt = d;
    } else
    for (int r = wave; r < tm.n; r += 4) {
        int d = 0;
        for (int c0 = 0; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            const bool nz = (c < tm.n && c != r) ? (A[tm.offQ + (size_t)r * tm.ld + c] != 0.0f) : false;
            d += __popcll(__ballot(nz));
        }
        cnt += d;
        if (small && lane == 0) deg[r] = d;
    }
    if (lane == 0) part[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        int pos = -1;
        if (small) {
            pos = 0;
            int singles = 0;  // same placement as k_sparse_resident: split rows first, the others one slot each
            bool placeable = true;
            for (int r = 0; r < tm.n; ++r) {
                if (deg[r] > SP_CHUNK) {
                    const int ns = sparse_slots_of(deg[r]);
                    placeable &= ns <= SP_MAX_SPLIT;
                    pos = sparse_place(pos, ns) + ns;
                } else {
                    ++singles;
                }
            }
            pos = placeable ? pos + singles : -1;
        }
        out[2 * blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        out[2 * blockIdx.x + 1] = pos;
    }
}

// The mask entries OFF the edges of a target (the diagonal included; explain.py:645-663 initialises all n x n of them) reach no output of
// the reference - Abar = A (.) sym(sigma(M)) is zero there - but its LOGGED size and entropy terms sum over all n^2 entries
// (explain.py:755-770), and torch.optim keeps updating them.  Each follows a closed scalar recursion through the optimiser:
//     g = (c_size - c_ent M / n^2) sigma'(M)          (no prediction, Laplacian or feature term reaches it).
// This kernel runs those recursions for the targets of the edge-sparse kernels when the loss is logged (gnnx_hyper.record_loss): one
// thread per DEAD_Q entries, state in registers for all iterations, per iteration the workgroup's sums of sigma(M) and of the entropy are
// added to loss[t][iter][1] and [3] (float atomics: logging only, like k_mask's), and at the end M (and the moments, if the caller
// wants the optimiser state back) is written for these entries - so M, like the loss, is what the dense kernels would have produced.
// grid: dead_blocks[k] = (target, first entry of its ld x ld block), DEAD_THREADS x DEAD_Q entries per workgroup.
constexpr int DEAD_THREADS = 256, DEAD_Q = 8;
struct DeadBlock { int32_t t; int32_t pad; int64_t first; };
__global__ __launch_bounds__(DEAD_THREADS) void k_dead_entries(Params p, const DeadBlock* blocks, const float* adam_tab) {
    __shared__ float part[DEAD_THREADS / 64][2];
    const DeadBlock db = blocks[blockIdx.x];
    const TargetMeta tm = p.meta[db.t];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = tm.n, ld = tm.ld;
    const float inv_n2 = 1.0f / ((float)n * (float)n);
    float M[DEAD_Q], m[DEAD_Q], v[DEAD_Q];
    bool dead[DEAD_Q];
    size_t at[DEAD_Q];
#pragma unroll
    for (int q = 0; q < DEAD_Q; ++q) {
        const long long e = db.first + (long long)q * DEAD_THREADS + tid;
        const int r = (int)(e / ld), c = (int)(e - (long long)r * ld);
        at[q] = (size_t)tm.offQ + (size_t)(e < (long long)ld * ld ? e : 0);
        dead[q] = e < (long long)ld * ld && r < n && c < n && (r == c || p.A[at[q]] == 0.0f);
        M[q] = dead[q] ? p.M[at[q]] : 0.0f;
        m[q] = (dead[q] && p.m_in) ? p.m_in[at[q]] : 0.0f;
        v[q] = (dead[q] && p.v_in) ? p.v_in[at[q]] : 0.0f;
    }
    for (int iter = 0; iter < p.num_iters; ++iter) {
        const float step_size = adam_tab[2 * iter], bc2s = adam_tab[2 * iter + 1];
        float s_size = 0.0f, s_ent = 0.0f;
#pragma unroll
        for (int q = 0; q < DEAD_Q; ++q) {
            const float S = sigmoidf_(M[q]);
            s_size += dead[q] ? S : 0.0f;
            s_ent += dead[q] ? -S * logf(S) - (1.0f - S) * logf(1.0f - S) : 0.0f;
            const float g = (p.c_size - p.c_ent * M[q] * inv_n2) * S * (1.0f - S);
            adam_update(M[q], m[q], v[q], g, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
        }
        if (p.loss) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                s_size += __shfl_xor(s_size, o);
                s_ent += __shfl_xor(s_ent, o);
            }
            if (lane == 0) {
                part[wave][0] = s_size;
                part[wave][1] = s_ent;
            }
            __syncthreads();
            if (tid == 0) {
                float a = 0.0f, b = 0.0f;
                for (int w = 0; w < DEAD_THREADS / 64; ++w) {
                    a += part[w][0];
                    b += part[w][1];
                }
                float* L = p.loss + ((size_t)db.t * p.num_iters + iter) * NLOSS;
                atomicAdd(&L[1], p.c_size * a);
                atomicAdd(&L[3], p.c_ent * b * inv_n2);
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int q = 0; q < DEAD_Q; ++q)
        if (dead[q]) {
            p.M[at[q]] = M[q];
            if (p.m_out) p.m_out[at[q]] = m[q];
            if (p.v_out) p.v_out[at[q]] = v[q];
        }
}

}  // namespace gnnx
