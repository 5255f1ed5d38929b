// gnnx_att.hpp — method="att": the attention GraphConv of the reference (models.py:36-37 weights, :62-68 forward)
//
//     x_att = x W_att;  att = x_att x_att^T;  adj' = adj * att;  y = normalize((adj' x) W + b)
//
// in all three layers of the encoder, and the whole mask optimisation through it (explain.py:685-715, 740-808, 137-146): one
// workgroup per target, every iteration inside one launch.  Only entries ON EDGES are live (adj' = 0 elsewhere and a mask entry
// off the edges never reaches masked_adj * sub_adj), so the state is a CSR of the target's sub-graph built at the start of the
// launch: w_e = Abar on the edge, s_e^l = att of layer l on the edge, and per layer
//
//   forward    u = Xin W_att;   s_e = u_i . u_k;   Z_i = sum_e w_e s_e Xin_k;   Y = Z W + b;   U = Y / max(|Y|, 1e-12)
//   backward   dZ = rowlocal(dU);  g_e = dZ_i . Xin_k  (= dL/dadj'_ik);  dAbar_e += g_e s_e;  q_e = g_e w_e  (= dL/datt_ik)
//              dXin_i = sum_e w_e s_e dZ_k                       (adj' is symmetric: Abar and att both are)
//                     + (sum_e (q_e + q_mirror(e)) u_k) W_att^T  (att_ik = u_i . u_k: u_i sits on both sides)
//
// followed by the same regularisers, symmetrisation and Adam step as k_mask (gnnx_kernels.hpp), edge by edge.  Rows are worked
// on by half-waves (lane = feature column; the two halves of a wave work in lockstep, padded to the longer one), the per-edge dot
// products are 32-lane butterflies; every sum has a fixed order (deterministic, batch-invariant).  Round 5: the phases walk only the
// rows their results are read from (hop pruning) in chunks of at most 32 entries (a hub row is shared by several half-waves) - see
// k_att.  The row arrays still live in the workspace (L2): this is the complete, parallel kernel for a flag the reference's experiments
// rarely use, not a roofline kernel.
#pragma once
#include "gnnx_kernels.hpp"

namespace gnnx {

struct AttScratch {
    const float* watt;       // [3][32][32] zero padded, W_att of layer l at l * 1024 + b * 32 + a  (b = input column)
    const long long* eoff;   // [T + 1] first directed edge of target t in the edge arrays
    int32_t* rowptr;         // [R + T]: the n + 1 row pointers of target t start at offR + t (relative to eoff[t])
    int32_t *col, *mir;      // [E] column of the entry (bits 16-17: the column's level, see lev), position of the mirror entry (k, i) (relative to eoff[t])
    float *w, *s[3], *q, *dA;  // [E]
    float *xin[3], *u[3], *U[3], *dZ, *dX;  // [R][32]
    float* rn[3];            // [R]
    // round 5: hop pruning and chunked rows
    int32_t* lev;            // [R] hop distance of the row from the target's node, capped at 3 (graph mode: 0 everywhere)
    int32_t* order;          // [R + T] the rows sorted by level (stable): the rows within k hops are the prefix [0, nr[k])
    int32_t* chunk;          // [R + T + E / 32 + T] row | (first entry >> 5) << 16 of every chunk of <= 32 entries, rows in `order`
    int32_t *mrow, *mfirst;  // [E / 32 + T] the rows of more than 32 entries (in `order`) and their first chunk
    float* pacc;             // [R + T + E / 32 + T][32] partial row sums of the chunks of those rows
};

// sum over the 32 lanes of a half-wave, in every lane: four DPP rotations inside each 16-lane row (register speed) and ONE cross-row
// shuffle, instead of five ds_bpermute round trips - this reduction sits on the per-edge chain of every gather below
template <int S>
__device__ __forceinline__ float att_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + S, 0xf, 0xf, false));
}
__device__ __forceinline__ float att_sum32(float v) {
    v += att_row_ror<8>(v);
    v += att_row_ror<4>(v);
    v += att_row_ror<2>(v);
    v += att_row_ror<1>(v);
    return xor16_sum(v);      // (gfx950: a lane swap, not a ds_bpermute - gnnx_kernels.hpp)
}

constexpr int ATT_THREADS = 1024;
constexpr int ATT_UN = 8;    // edges whose row loads are in flight together in the gathers of k_att

// Round 5 (what 130 ms per 400-target syn1 batch were): every phase walked ALL rows of the sub-graph, and a half-wave walked ALL entries of its
// row - the 250 entries of a hub row one after the other, 0.12 us each, while the other 31 half-waves waited at the barrier: 30 us per phase,
// twelve phases per iteration.
//  * HOP PRUNING (node mode).  The loss reads row t of layer 3 only: layer l's output is needed on the rows within 2 - l hops of t (its
//    inputs one hop further), and the backward of layer l has dZ on those rows and passes dXin to the rows within 3 - l hops.  The rows are
//    sorted by hop distance once (`order`), every phase walks a prefix.  Rows outside a phase's set are not touched - what the old kernel
//    computed there was multiplied by zero or never read; the guards below (levels of a row / of an entry's column) keep stale values of
//    earlier phases out.  Graph mode: every row is at level 0 (the max-pools read all rows).
//  * CHUNKED ROWS.  The unit of work of an edge phase is a CHUNK of at most 32 entries of one row; the 8 chunks of a hub row go to 8
//    half-waves.  A row of one chunk finishes as before; the partial sums of the others go through `pacc` and a second, short pass adds them in
//    chunk order (fixed order: deterministic, batch-invariant) and runs the row-local part.
// `order_wg` (round 6): workgroup -> target, the targets by decreasing edge count, so that a batch of more targets than compute units (one
// 1024-thread workgroup each) starts its longest chains first.  Round 6 also measured three reformulations of this kernel and dropped them - none
// moved the 58 ms of the 400-target syn1 batch: the per-entry dot products with lane = entry (din + din / 4 instead of ~15
// wave-instructions per entry), the two gathered row arrays of every phase staged in LDS (n <= 448), and 512- / 256-thread workgroups (86 / 146 ms:
// the batch's time is the n = 310 target's own chain, 193 us per iteration, and it scales with the half-waves that share its chunks).
__global__ __launch_bounds__(ATT_THREADS) void k_att(Params p, AttScratch a, const float* __restrict__ adam, const int32_t* __restrict__ order_wg = nullptr) {
    constexpr int NWV = ATT_THREADS / 64;
    __shared__ float sW[3][32 * 33], sWa[3][32 * 33], sb[3][32];
    __shared__ float sWp[CMAX * 96 + CMAX];
    __shared__ float fcur[32], mf[32], vf[32], phi[32];
    __shared__ float emb[96], gcls[CMAX], dEs[96];
    __shared__ int erow[96];
    __shared__ float part[2 * NWV][32];
    __shared__ int s_nr[4], s_nch[4], s_nm[4];   // rows / chunks / rows of several chunks within k hops, k = 0 .. 3
    const int t = order_wg ? order_wg[blockIdx.x] : (int)blockIdx.x;
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, ld = tm.ld, tr = tm.t;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ln = lane & 31, hf = lane >> 5, hbase = lane & 32;
    const int D = p.D, H = p.H;
    const long long eb = a.eoff[t];
    const int nnz = (int)(a.eoff[t + 1] - eb);
    int32_t* rowptr = a.rowptr + tm.offR + t;
    int32_t* col = a.col + eb;
    int32_t* mir = a.mir + eb;
    float* w = a.w + eb;
    float* q = a.q + eb;
    float* dA = a.dA + eb;
    int32_t* lev = a.lev + tm.offR;
    int32_t* order = a.order + tm.offR + t;
    const size_t cbase = (size_t)tm.offR + t + (size_t)(eb >> 5);
    int32_t* chunk = a.chunk + cbase;
    float* pacc = a.pacc + cbase * 32;
    int32_t* mrow = a.mrow + (eb >> 5) + t;
    int32_t* mfirst = a.mfirst + (eb >> 5) + t;
    const size_t ro = (size_t)tm.offR * FS;
    const float* X = p.X + ro;
    const float* Ad = p.A + tm.offQ;
    float* Md = p.M + tm.offQ;
    float* md = p.mM + tm.offQ;
    float* vd = p.vM + tm.offQ;
    const float* yh = p.graph_mode ? nullptr : p.yhat + tm.offR;   // graph mode has no Laplacian term (explain.py:780)

    // ---------------- set-up: model, feature-mask state, CSR of the sub-graph ----------------
    for (int l = 0; l < 3; ++l)
        for (int e = tid; e < 1024; e += ATT_THREADS) {
            sW[l][(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + l * 1024 + e];
            sWa[l][(e >> 5) * 33 + (e & 31)] = a.watt[l * 1024 + e];
        }
    if (tid < 96) sb[tid >> 5][tid & 31] = p.wts[WT_B + tid];
    stage_head_weights(p, sWp);
    if (tid < 32) {
        const float* fs = p.fs_in ? p.fs_in + (size_t)t * 3 * FS + tid : nullptr;  // gnnx_run_resume
        fcur[tid] = (fs && tid < D) ? fs[0] : 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
        mf[tid] = (fs && tid < D) ? fs[FS] : 0.0f;
        vf[tid] = (fs && tid < D) ? fs[2 * FS] : 0.0f;
    }
    if (!p.edge_only)
        for (size_t e = tid; e < (size_t)ld * ld; e += ATT_THREADS) p.Abar[tm.offQ + e] = 0.0f;
    // row degrees (off the diagonal), one wave per row
    for (int i = wave; i < n; i += NWV) {
        int c = 0;
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            c += __popcll(__ballot(k < n && k != i && Ad[(size_t)i * ld + k] != 0.0f));
        }
        if (lane == 0) rowptr[i + 1] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int s = 0;
        rowptr[0] = 0;
        for (int i = 0; i < n; ++i) {
            s += rowptr[i + 1];
            rowptr[i + 1] = s;
        }
    }
    __syncthreads();
    for (int i = wave; i < n; i += NWV) {
        int pos = rowptr[i];
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const bool on = k < n && k != i && Ad[(size_t)i * ld + k] != 0.0f;
            const unsigned long long bal = __ballot(on);
            if (on) col[pos + __popcll(bal & ((1ull << lane) - 1ull))] = k;
            pos += __popcll(bal);
        }
    }
    for (int i = tid; i < n; i += ATT_THREADS) lev[i] = (p.graph_mode || i == tr) ? 0 : 3;
    __syncthreads();
    // mirror positions, Adam moments of the live entries, the first masked adjacency
    for (int i = wave; i < n; i += NWV) {
        for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
            const int k = col[e];
            int lo = rowptr[k], hi = rowptr[k + 1] - 1, at = -1;
            while (lo <= hi) {  // the pattern of A is symmetric (the reference's sub_adj is): (k, i) exists
                const int mid = (lo + hi) >> 1, c = col[mid];
                if (c == i) { at = mid; break; }
                if (c < i) lo = mid + 1; else hi = mid - 1;
            }
            mir[e] = at < 0 ? e : at;
            const size_t idx = (size_t)i * ld + k;
            md[idx] = p.m_in ? p.m_in[tm.offQ + idx] : 0.0f;
            vd[idx] = p.v_in ? p.v_in[tm.offQ + idx] : 0.0f;
            w[e] = Ad[idx] * (0.5f * (sigmoidf_(Md[idx]) + sigmoidf_(Md[(size_t)k * ld + i])));
        }
    }
    __syncthreads();
    // hop distances from the target's node (two rounds over the rows of the current level; concurrent writers store the same value)
    if (!p.graph_mode)
        for (int round = 0; round < 2; ++round) {
            for (int i = wave; i < n; i += NWV)
                if (lev[i] == round)
                    for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64)
                        if (lev[col[e]] > round + 1) lev[col[e]] = round + 1;
            __syncthreads();
        }
    // every entry carries its column's level; rows sorted by level, cut into chunks
    for (int i = wave; i < n; i += NWV)
        for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) col[e] |= lev[col[e]] << 16;
    if (tid == 0) {
        int pos = 0, nc = 0, nm = 0;
        for (int L = 0; L < 4; ++L) {
            for (int i = 0; i < n; ++i)
                if (lev[i] == L) {
                    order[pos++] = i;
                    const int deg = rowptr[i + 1] - rowptr[i];
                    if (deg > 32) {
                        mrow[nm] = i;
                        mfirst[nm] = nc;
                        ++nm;
                    }
                    int j0 = 0;
                    do {   // (a row without entries still has its row-local part: one empty chunk)
                        chunk[nc++] = i | ((j0 >> 5) << 16);
                        j0 += 32;
                    } while (j0 < deg);
                }
            s_nr[L] = pos;
            s_nch[L] = nc;
            s_nm[L] = nm;
        }
    }
    __syncthreads();
    const float inv_n2 = 1.0f / ((float)n * (float)n);

    // the chunk this half-wave works on in trip c0 of a walk over the first NC chunks: its row, the row's entries, the chunk's entry records
    // one per lane (coalesced; handed out by shuffles in the gathers).  `cnt` is the trip count of the WAVE (the longer of its two chunks).
    struct Chunk {
        int c, i, e0, deg, j0, mine, cnt;
        bool on, multi;
    };
    auto chunk_of = [&](int c0, int NC) {
        Chunk k;
        k.c = c0 + hf;
        k.on = k.c < NC;
        const int rec = k.on ? chunk[k.c] : 0;
        k.i = rec & 0xffff;
        k.j0 = (rec >> 16) << 5;
        k.e0 = k.on ? rowptr[k.i] : 0;
        k.deg = k.on ? rowptr[k.i + 1] - k.e0 : 0;
        const int left = k.deg - k.j0;
        k.mine = left < 0 ? 0 : (left < 32 ? left : 32);
        const int other = __shfl_xor(k.mine, 32);
        k.cnt = k.mine > other ? k.mine : other;
        k.multi = k.deg > 32;
        return k;
    };
    // the rows of several chunks among the first NM of them: their partial sums added in chunk order
    auto multi_sum = [&](int m0, int NM, int& i, bool& on) {
        const int m = m0 + hf;
        on = m < NM;
        i = on ? mrow[m] : 0;
        const int c1 = on ? mfirst[m] : 0;
        const int nck = on ? (rowptr[i + 1] - rowptr[i] + 31) >> 5 : 0;
        const int onck = __shfl_xor(nck, 32), trip = nck > onck ? nck : onck;
        float acc = 0.0f;
        for (int k = 0; k < trip; ++k) acc += (k < nck) ? pacc[(size_t)(c1 + k) * 32 + ln] : 0.0f;
        return acc;
    };

    for (int iter = 0; iter < p.num_iters; ++iter) {
        // ======== forward ========
        if (tid < 32) phi[tid] = (tid < D) ? sigmoidf_(fcur[tid]) : 0.0f;
        __syncthreads();
        // masked features and their attention projection (row-local; every row: layer 1's gathers reach three hops)
        for (int i0 = 2 * wave; i0 < n; i0 += 2 * NWV) {
            const int i = i0 + hf;
            const bool on = i < n;
            const float x = on ? X[(size_t)i * FS + ln] * phi[ln] : 0.0f;
            float uu = 0.0f;
            for (int b = 0; b < D; ++b) uu = fmaf(__shfl(x, hbase | b), sWa[0][b * 33 + ln], uu);
            if (on) {
                a.xin[0][ro + (size_t)i * FS + ln] = x;
                a.u[0][ro + (size_t)i * FS + ln] = uu;
            }
        }
        __syncthreads();
        for (int l = 0; l < 3; ++l) {
            const int din = (l == 0) ? D : H, dout = (l == 2) ? p.O : H;
            const float* xin = a.xin[l] + ro;
            const float* ul = a.u[l] + ro;
            float* sl = a.s[l] + eb;
            // row-local: Y = Z W + b, U = Y / max(|Y|, 1e-12), the next layer's input and its attention projection
            auto rowlocal = [&](int i, bool ron, float acc) {
                float y = 0.0f;
                for (int b = 0; b < din; ++b) y = fmaf(__shfl(acc, hbase | b), sW[l][b * 33 + ln], y);
                y = (ln < dout) ? y + sb[l][ln] : 0.0f;
                const float rnorm = fmaxf(sqrtf(att_sum32(y * y)), 1e-12f);
                const float un = y / rnorm;
                const float xn = fmaxf(un, 0.0f);
                float uu = 0.0f;
                if (l < 2)
                    for (int b = 0; b < dout; ++b) uu = fmaf(__shfl(xn, hbase | b), sWa[l + 1][b * 33 + ln], uu);
                if (ron) {
                    a.U[l][ro + (size_t)i * FS + ln] = un;
                    if (ln == 0) a.rn[l][tm.offR + i] = rnorm;
                    if (l < 2) {
                        a.xin[l + 1][ro + (size_t)i * FS + ln] = xn;
                        a.u[l + 1][ro + (size_t)i * FS + ln] = uu;
                    }
                }
            };
            const int NC = s_nch[2 - l], NM = s_nm[2 - l];
            for (int c0 = 2 * wave; c0 < NC; c0 += 2 * NWV) {
                const Chunk k = chunk_of(c0, NC);
                const float ui = k.on ? ul[(size_t)k.i * FS + ln] : 0.0f;
                const bool eon = ln < k.mine;
                const int ck = eon ? (col[k.e0 + k.j0 + ln] & 0xffff) : 0;
                const float cw = eon ? w[k.e0 + k.j0 + ln] : 0.0f;
                float cs = 0.0f, acc = 0.0f;
                // ATT_UN edges per trip: the row loads of all of them (L2: the row arrays live in the workspace) are issued before the first
                // butterfly, so an entry costs a fraction of one L2 round trip; the products are taken in entry order
                for (int jj = 0; jj < k.cnt; jj += ATT_UN) {
                    int kk[ATT_UN];
                    float uk[ATT_UN], xk[ATT_UN];
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u) kk[u] = __shfl(ck, hbase | ((jj + u < k.cnt) ? jj + u : jj));
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u) {
                        uk[u] = ul[(size_t)kk[u] * FS + ln];
                        xk[u] = xin[(size_t)kk[u] * FS + ln];
                    }
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u)
                        if (jj + u < k.cnt) {     // uniform over the wave (cnt is)
                            const float se = att_sum32(ui * uk[u]);
                            acc = fmaf(__shfl(cw, hbase | (jj + u)) * se, xk[u], acc);
                            cs = (jj + u == ln) ? se : cs;
                        }
                }
                if (eon) sl[k.e0 + k.j0 + ln] = cs;
                if (k.on && k.multi) pacc[(size_t)k.c * 32 + ln] = acc;
                rowlocal(k.i, k.on && !k.multi, acc);
            }
            __syncthreads();
            for (int m0 = 2 * wave; m0 < NM; m0 += 2 * NWV) {
                int i;
                bool on;
                const float acc = multi_sum(m0, NM, i, on);
                rowlocal(i, on, acc);
            }
            __syncthreads();
        }
        // head: row t of the three layers in node mode (explain.py:713, models.py:372-380); in graph mode the column-wise max over
        // ALL n rows of every layer (models.py:283, 291, 300; the first maximal row wins, as torch.max)
        if (tid < 96) {
            const int l = tid >> 5, c = tid & 31;
            if (p.graph_mode) {
                const int dl = (l == 2) ? p.O : H;
                float best = 0.0f;
                int arg = 0;
                if (c < dl) {
                    best = -3.0e38f;
                    for (int i = 0; i < n; ++i) {
                        float v = a.U[l][ro + (size_t)i * FS + c];
                        if (l < 2) v = fmaxf(v, 0.0f);
                        if (v > best) { best = v; arg = i; }
                    }
                }
                emb[tid] = best;
                erow[tid] = arg;
            } else {
                const float v = a.U[l][ro + (size_t)tr * FS + c];
                emb[tid] = (l < 2) ? fmaxf(v, 0.0f) : v;
                erow[tid] = tr;
            }
        }
        for (int e = tid; e < nnz; e += ATT_THREADS) dA[e] = 0.0f;   // (the rows a layer's backward does not reach add nothing)
        __syncthreads();
        head_softmax(p, tm, t, iter, true, sWp, emb, gcls, dEs);

        // ======== backward ========
        for (int l = 2; l >= 0; --l) {
            const int din = (l == 0) ? D : H, dout = (l == 2) ? p.O : H;
            const float* xin = a.xin[l] + ro;
            const float* ul = a.u[l] + ro;
            const float* sl = a.s[l] + eb;
            float* dZ = a.dZ + ro;
            float* dX = a.dX + ro;
            const int LZ = 2 - l;                    // rows within LZ hops have a dZ of this layer
            const int NRZ = s_nr[LZ];
            const int NC = s_nch[LZ + 1 > 3 ? 3 : LZ + 1], NM = s_nm[LZ + 1 > 3 ? 3 : LZ + 1];   // dXin: one hop further
            // row-local: gradient of the layer's output -> dZ
            for (int r0 = 2 * wave; r0 < NRZ; r0 += 2 * NWV) {
                const bool ron = r0 + hf < NRZ;
                const int i = ron ? order[r0 + hf] : 0;
                const size_t r = (size_t)i * FS + ln;
                float dx = (l < 2) ? dX[r] : 0.0f;
                if (ron && erow[l * 32 + ln] == i) dx += dEs[l * 32 + ln];   // the direct part lands on row t / on the arg-max row of the column
                const float un = a.U[l][ro + r];
                float du = (ln < dout) ? dx : 0.0f;
                if (l < 2) du = (un > 0.0f) ? du : 0.0f;
                const float sd = att_sum32(du * un);
                const float dy = (du - un * sd) / a.rn[l][tm.offR + i];
                float dz = 0.0f;
                for (int c = 0; c < dout; ++c) dz = fmaf(__shfl(dy, hbase | c), sW[l][ln * 33 + c], dz);
                if (ron) dZ[r] = (ln < din) ? dz : 0.0f;
            }
            __syncthreads();
            // edges: dL/dadj' -> dAbar, dL/datt, and the adj'-path of dXin.  A row beyond LZ hops has no dZ (g = 0) but takes the adj' path
            // from its neighbours within LZ hops - adj' is symmetric, the attention value of such an entry is its mirror's.
            for (int c0 = 2 * wave; c0 < NC; c0 += 2 * NWV) {
                const Chunk k = chunk_of(c0, NC);
                const bool inz = k.on && lev[k.i] <= LZ;
                const float dzi = inz ? dZ[(size_t)k.i * FS + ln] : 0.0f;
                const bool eon = ln < k.mine;
                const int e = k.e0 + k.j0 + ln;
                const int ckp = eon ? col[e] : 0;
                const int ck = ckp & 0xffff;
                const bool kin = eon && (ckp >> 16) <= LZ;
                const float cw = eon ? w[e] : 0.0f;
                const float cs = !eon ? 0.0f : inz ? sl[e] : kin ? sl[mir[e]] : 0.0f;
                const float cc = kin ? cw * cs : 0.0f;
                float cg = 0.0f, acc = 0.0f;
                for (int jj = 0; jj < k.cnt; jj += ATT_UN) {
                    int kk[ATT_UN];
                    float xk[ATT_UN], zk[ATT_UN];
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u) kk[u] = __shfl(ck, hbase | ((jj + u < k.cnt) ? jj + u : jj));
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u) {
                        xk[u] = xin[(size_t)kk[u] * FS + ln];
                        zk[u] = dZ[(size_t)kk[u] * FS + ln];
                    }
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u)
                        if (jj + u < k.cnt) {
                            const float g = att_sum32(dzi * xk[u]);
                            acc = fmaf(__shfl(cc, hbase | (jj + u)), zk[u], acc);
                            cg = (jj + u == ln) ? g : cg;
                        }
                }
                if (eon) {
                    if (inz) dA[e] += cg * cs;
                    q[e] = inz ? cg * cw : 0.0f;
                }
                if (k.on) {
                    if (k.multi) pacc[(size_t)k.c * 32 + ln] = acc;
                    else dX[(size_t)k.i * FS + ln] = acc;
                }
            }
            __syncthreads();
            for (int m0 = 2 * wave; m0 < NM; m0 += 2 * NWV) {
                int i;
                bool on;
                const float acc = multi_sum(m0, NM, i, on);
                if (on) dX[(size_t)i * FS + ln] = acc;
            }
            __syncthreads();
            // edges: the attention path, then dXin complete (+ the feature-mask partials in layer 1)
            float fp = 0.0f;
            auto finish = [&](int i, bool ron, float du) {
                float dxa = 0.0f;
                for (int c = 0; c < din; ++c) dxa = fmaf(__shfl(du, hbase | c), sWa[l][ln * 33 + c], dxa);
                if (ron) {
                    const size_t r = (size_t)i * FS + ln;
                    const float dxi = (ln < din) ? dX[r] + dxa : 0.0f;
                    dX[r] = dxi;
                    if (l == 0) fp = fmaf(dxi, X[r], fp);
                }
            };
            for (int c0 = 2 * wave; c0 < NC; c0 += 2 * NWV) {
                const Chunk k = chunk_of(c0, NC);
                const bool inz = k.on && lev[k.i] <= LZ;
                const bool eon = ln < k.mine;
                const int e = k.e0 + k.j0 + ln;
                const int ckp = eon ? col[e] : 0;
                const int ck = ckp & 0xffff;
                const bool kin = eon && (ckp >> 16) <= LZ;
                const float cq = (inz && eon ? q[e] : 0.0f) + (kin ? q[mir[e]] : 0.0f);   // q lives on the rows within LZ hops
                float du = 0.0f;
                for (int jj = 0; jj < k.cnt; jj += ATT_UN) {
                    float uk[ATT_UN];
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u) uk[u] = ul[(size_t)__shfl(ck, hbase | ((jj + u < k.cnt) ? jj + u : jj)) * FS + ln];
#pragma unroll
                    for (int u = 0; u < ATT_UN; ++u)
                        if (jj + u < k.cnt) du = fmaf(__shfl(cq, hbase | (jj + u)), uk[u], du);
                }
                if (k.on && k.multi) pacc[(size_t)k.c * 32 + ln] = du;
                finish(k.i, k.on && !k.multi, du);
            }
            __syncthreads();
            for (int m0 = 2 * wave; m0 < NM; m0 += 2 * NWV) {
                int i;
                bool on;
                const float du = multi_sum(m0, NM, i, on);
                finish(i, on, du);
            }
            if (l == 0) part[2 * wave + hf][ln] = fp;
            __syncthreads();
        }

        // ======== mask entries on the edges: regularisers, symmetrisation, Adam (as k_mask) ========
        const float step_size = adam[2 * iter], bc2s = adam[2 * iter + 1];
        for (int i = wave; i < n; i += NWV) {
            const float yi = p.graph_mode ? 0.0f : yh[i];
            for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
                const int k = col[e] & 0xffff;
                const size_t idx = (size_t)i * ld + k;
                float Gs = 0.5f * (dA[e] + dA[mir[e]]);
                if (!p.graph_mode) {
                    const float dy = yi - yh[k];
                    Gs += p.c_lap * 0.5f * dy * dy * inv_n2;
                }
                float Mij = Md[idx], mij = md[idx], vij = vd[idx];
                const float Sij = sigmoidf_(Mij);
                const float gij = (Gs * Ad[idx] + p.c_size + p.c_ent * (-Mij) * inv_n2) * (Sij * (1.0f - Sij));
                adam_update(Mij, mij, vij, gij, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
                Md[idx] = Mij;
                md[idx] = mij;
                vd[idx] = vij;
            }
        }
        if (tid < D) {  // feature mask
            float dsum = 0.0f;
            for (int r = 0; r < 2 * NWV; ++r) dsum += part[r][tid];
            const float ph = phi[tid];
            const float gf = (dsum + p.c_feat_size / (float)D) * ph * (1.0f - ph);
            float fn = fcur[tid], m = mf[tid], v = vf[tid];
            adam_update(fn, m, v, gf, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            fcur[tid] = fn;
            mf[tid] = m;
            vf[tid] = v;
        }
        __syncthreads();
        if (iter + 1 < p.num_iters) {  // the result is the masked adjacency of the LAST forward (explain.py:209-211)
            for (int i = wave; i < n; i += NWV)
                for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
                    const int k = col[e] & 0xffff;
                    const size_t idx = (size_t)i * ld + k;
                    w[e] = Ad[idx] * (0.5f * (sigmoidf_(Md[idx]) + sigmoidf_(Md[(size_t)k * ld + i])));
                }
            __syncthreads();
        }
    }

    // ---------------- results ----------------
    for (int i = wave; i < n; i += NWV)
        for (int e = rowptr[i] + lane; e < rowptr[i + 1]; e += 64) {
            const size_t idx = (size_t)i * ld + (col[e] & 0xffff);
            p.Abar[tm.offQ + idx] = w[e];
            if (p.m_out) p.m_out[tm.offQ + idx] = md[idx];
            if (p.v_out) p.v_out[tm.offQ + idx] = vd[idx];
        }
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? fcur[tid] : 0.0f;
    if (p.fs_out && tid < FS) {
        float* fs = p.fs_out + (size_t)t * 3 * FS + tid;
        fs[0] = (tid < D) ? fcur[tid] : 0.0f;
        fs[FS] = (tid < D) ? mf[tid] : 0.0f;
        fs[2 * FS] = (tid < D) ? vf[tid] : 0.0f;
    }
}

// directed off-diagonal entries of every target (the host sizes the edge arrays from them)
__global__ __launch_bounds__(256) void k_att_count(const TargetMeta* meta, const float* A, int32_t* cnt) {
    __shared__ int red[4];
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int c = 0;
    for (int i = wave; i < tm.n; i += 4)
        for (int k0 = 0; k0 < tm.n; k0 += 64) {
            const int k = k0 + lane;
            c += __popcll(__ballot(k < tm.n && k != i && A[tm.offQ + (size_t)i * tm.ld + k] != 0.0f));
        }
    if (lane == 0) red[wave] = c;
    __syncthreads();
    if (tid == 0) cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

}  // namespace gnnx
