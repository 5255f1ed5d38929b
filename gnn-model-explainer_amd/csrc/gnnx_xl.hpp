// gnnx_xl.hpp — CSR-native preparation of the XL route (k_sparse_large<.., XL = true>, gnnx_sparse_large.hpp): a target's sub-graph
// `adj[nb][:, nb]`, `feat[nb]`, `argmax(pred[nb])` (Explainer.extract_neighborhood / explain, explain.py:492-501, 94-106) as a CSR of LOCAL ids built
// straight from the resident full-graph CSR and the ascending k-hop list (gnnx_khop) - no dense n x n block is ever written.  The dense
// routes pack 4 n^2 bytes per target and scan them back (k_pack, k_row_degrees, k_csr_emit_large); at n = 20 000 that is 1.6 GB of packing
// traffic for a sub-graph whose CSR is 0.5 MB.  HBM-bound integer work, one thread per sub-graph row:
//   k_xl_rowdeg   row r -> global node g = nb[r]; its neighbours that are members of the list (binary search in the ascending list) are the row's
//                 entries: the row's degree, how many of them lie above the diagonal (the upper-triangle edge order of the results), the packed
//                 feature row and the predicted class id;
//   k_xl_rowptr   one workgroup per target: exclusive scans of the two counts -> rowptr [ld + 1], uprow [ld + 1]; totals to the host;
//   k_xl_emit     the rows' local column ids (ascending: the list and the full graph's rows both are), the row of every entry, its weight,
//                 and the (r, c) ids of the upper-triangle edges in row-major order - the order of gnnx_gather_edges.
#pragma once
#include "gnnx_sparse_large.hpp"

namespace gnnx {

struct XlBlock { int32_t t, r0; };      // a workgroup's 256 rows of target t
constexpr int XL_ROWS_PER_BLOCK = 256;

// position of global node g in the ascending list nb[0..n), or -1
__device__ __forceinline__ int xl_local_id(const int32_t* nb, int n, int g) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nb[mid] < g) lo = mid + 1; else hi = mid;
    }
    return (lo < n && nb[lo] == g) ? lo : -1;
}

__global__ __launch_bounds__(XL_ROWS_PER_BLOCK) void k_xl_rowdeg(const TargetMeta* meta, const XlBlock* blocks, const int64_t* indptr,
                                                                                    const int32_t* indices, const float* weights, const int32_t* nb, const int64_t* nb_off,
                                                                                    const float* feat, int feat_stride, int D, const float* pred_label,
                                                                                    float* X, float* yhat, int32_t* deg, int32_t* updeg) {
    const XlBlock b = blocks[blockIdx.x];
    const TargetMeta tm = meta[b.t];
    const int r = b.r0 + (int)threadIdx.x;
    if (r >= tm.ld) return;
    const int32_t* list = nb + nb_off[b.t];
    int d = 0, up = 0;
    float* xrow = X + (tm.offR + r) * FS;
    if (r < tm.n) {
        const int g = list[r];
        for (int64_t e = indptr[g]; e < indptr[g + 1]; ++e) {
            const int c = indices[e];
            if (c == g) continue;                              // the diagonal is masked out (explain.py:618, 678)
            if (e > indptr[g] && indices[e - 1] == c) continue;    // a repeated CSR entry counts once (the dense packing overwrites it)
            if (weights && weights[e] == 0.0f) continue;           // an explicit zero is no edge (the dense routes test A != 0)
            const int lc = xl_local_id(list, tm.n, c);
            if (lc >= 0) {
                ++d;
                up += lc > r ? 1 : 0;
            }
        }
        for (int k = 0; k < FS; ++k) xrow[k] = (k < D) ? feat[(size_t)g * feat_stride + k] : 0.0f;
        yhat[tm.offR + r] = pred_label ? pred_label[g] : 0.0f;
    } else {
        for (int k = 0; k < FS; ++k) xrow[k] = 0.0f;
        yhat[tm.offR + r] = 0.0f;
    }
    deg[tm.offR + r] = d;
    updeg[tm.offR + r] = up;
}

// exclusive scans of deg / updeg over the ld rows of one target -> rowptr [ld + 1] (at rp_off[t]), uprow [ld + 1]; totals[2 t], [2 t + 1]
// (256 threads: four waves find a place beside a workgroup of the XL loop - 230 registers per lane, two waves per SIMD - where sixteen would wait
// for a whole compute unit: 171 ms behind a batch of the largest sub-graphs, profiles/r06_kernel_stats_ba100k_all.csv)
constexpr int XL_ROWPTR_THREADS = 256;
__global__ __launch_bounds__(XL_ROWPTR_THREADS) void k_xl_rowptr(const TargetMeta* meta, const int32_t* deg, const int32_t* updeg, const long long* rp_off,
                                                                 int32_t* rowptr, int32_t* uprow, int32_t* totals) {
    constexpr int NT = XL_ROWPTR_THREADS, NW = NT / 64;
    __shared__ int part[2][NW];
    const int t = blockIdx.x;
    const TargetMeta tm = meta[t];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int per = (tm.ld + NT - 1) / NT;
    const int lo = tid * per, hi = (lo + per < tm.ld) ? lo + per : tm.ld;
    const int32_t* src[2] = {deg + tm.offR, updeg + tm.offR};
    int32_t* dst[2] = {rowptr + rp_off[t], uprow + rp_off[t]};
    int sum[2] = {0, 0}, incl[2];
    for (int a = 0; a < 2; ++a) {
        for (int r = lo; r < hi; ++r) sum[a] += src[a][r];
        incl[a] = wave_scan_inclusive(sum[a], lane);
        if (lane == 63) part[a][wave] = incl[a];
    }
    __syncthreads();
    for (int a = 0; a < 2; ++a) {
        int base = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
            base += (w < wave) ? part[a][w] : 0;
            total += part[a][w];
        }
        int run = base + incl[a] - sum[a];
        for (int r = lo; r < hi; ++r) {
            const int v = src[a][r];
            dst[a][r] = run;
            run += v;
        }
        if (tid == 0) {
            dst[a][tm.ld] = total;
            totals[2 * t + a] = total;
        }
    }
}

// CT = int: the ids of the XL form
__global__ __launch_bounds__(XL_ROWS_PER_BLOCK) void k_xl_emit(const TargetMeta* meta, const XlBlock* blocks, const int64_t* indptr,
                                                                                  const int32_t* indices, const float* weights, const int32_t* nb,
                                                                                  const int64_t* nb_off, const long long* csr_off, const int32_t* rowptr,
                                                                                  const int32_t* uprow, const long long* eoff, int32_t* col, int32_t* row,
                                                                                  float* w, int32_t* rc) {
    const XlBlock b = blocks[blockIdx.x];
    const TargetMeta tm = meta[b.t];
    const int r = b.r0 + (int)threadIdx.x;
    if (r >= tm.n) return;
    const int32_t* list = nb + nb_off[b.t];
    const int32_t* rp = rowptr + csr_off[2 * b.t];
    const int32_t* ur = uprow + csr_off[2 * b.t];
    int32_t* c_out = col + csr_off[2 * b.t + 1];
    int32_t* r_out = row + csr_off[2 * b.t + 1];
    float* w_out = w ? w + csr_off[2 * b.t + 1] : nullptr;
    int e_out = rp[r];
    long long q = eoff[b.t] + ur[r];
    const int g = list[r];
    for (int64_t e = indptr[g]; e < indptr[g + 1]; ++e) {
        const int c = indices[e];
        if (c == g) continue;
        if (e > indptr[g] && indices[e - 1] == c) continue;
        if (weights && weights[e] == 0.0f) continue;
        const int lc = xl_local_id(list, tm.n, c);
        if (lc < 0) continue;
        c_out[e_out] = lc;
        r_out[e_out] = r;
        if (w_out) w_out[e_out] = weights ? weights[e] : 1.0f;
        ++e_out;
        if (lc > r) {
            rc[2 * q] = r;
            rc[2 * q + 1] = lc;
            ++q;
        }
    }
}

}  // namespace gnnx

namespace gnnx {

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The seeded initial masks of XL targets with the engine walked on the DEVICE, NO n^2 scratch, and the walk of ONE target spread over many
// workgroups (round 6).  gnnx_mt_edge_words (gnnx_graph.hpp) writes the raw mt19937 state words of all n^2 draws of a target into its dense block
// and gathers the pairs afterwards - 9.6 GB of words for a 49 028-node target - and walks every target's engine in one serial chain of n^2 / 624
// block updates: 3.7 million (1.7 s of one workgroup) for the 47 913-node sub-graph of BA-House x100k.
//   * The directed entries of a sub-graph CSR are ASCENDING in their stream position p = r n + c (row-major rows, ascending columns), so one pass
//     suffices: a workgroup walks its part of the engine block by block (624 words per update, three data-parallel sweeps) and hands every entry
//     whose Box-Muller pair lies in the current block its two raw words on the spot.  Entries are staged 256 at a time in LDS (block index, word
//     offset, destination 4 q + 2 dir in the edge-list layout of gnnx_host_transform_edge_words); a block without entries - nine out of ten on
//     BA-House x100k - costs one LDS compare beside its update.  The last 16 values of a ragged stream are redrawn from the 16 draws that follow
//     the fill (ATen's normal_fill): those words are kept as they pass and handed out at the end.
//   * mt19937 is linear over GF(2): x_{m + J} = XOR of x_{m + i} over the set coefficients of g(x) = x^J mod phi(x) (utils/mt_jump.py: phi by
//     Berlekamp-Massey, g by square-and-multiply; the published jump-ahead of Haramoto et al. 2008).  k_mt_segment_starts: one workgroup per target
//     seeds the engine, produces the first block, and then jumps from segment start to segment start - 33 blocks of the plain sequence in LDS,
//     10 047 XORs of sliding 624-word windows: ~0.1 ms per jump instead of 13 440 block updates (6.3 ms).  k_mt_edge_words_seg: one workgroup
//     per (target, segment of `jump` draws) walks its segment from its start state.  The 47 913-node target: 274 segments, ~30 ms of jumps + 6 ms
//     of walking instead of 1.7 s; a 17 000-node target: 34 segments.  Bit-identical to the serial walk (tests/test_xl_route.py, tests/test_mt_jump.py).
// words [E][4] (uint32): for every upper-triangle edge (r, c): {w(j), w(j + 8)} of entry (r, c), then of entry (c, r).
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int MTX_THREADS = 256, MTX_CHUNK = 256;
constexpr int MT_DEG = 19937;

struct MtSeg { int32_t t, s; };      // work item of k_mt_edge_words_seg: segment s of target t

// draws of target t: n^2 (+ 16 redrawn when ragged); segments of `jump` draws, the last one takes the remainder (so it always holds the tail)
__host__ __device__ inline long long mt_total_draws(long long n) { const long long nn = n * n; return nn + ((nn & 15) ? 16 : 0); }
__host__ __device__ inline int mt_segments(long long n, long long jump) {
    const long long total = mt_total_draws(n);
    if (n * n < 16 || jump <= 0) return (n * n < 16) ? 0 : 1;
    const long long k = total / jump;
    return (int)(k < 1 ? 1 : k);
}

// seg_state [seg_off[t] + s][624]: the block of draws [s jump, s jump + 624) of target t - the state its segment starts from.
// The jump needs the plain sequence x_m .. x_{m + 19936 + 623} only as a SLIDING window: term i touches y[i .. i + 623], so three 624-word blocks in a
// ring (the block that holds i, the next one, and the one being generated) are enough - 7.5 KB of LDS instead of 85, which lets the kernel run
// beside the 96 KB workgroups of the XL loop instead of waiting for a compute unit to drain (profiles/r06_kernel_stats_ba100k_all.csv: 471 ms
// behind a batch of the largest sub-graphs with the 85 KB form).
// Hierarchical strides (round 6, second half): a target's segment starts were ONE serial chain of K - 1 jumps (0.9 ms each in the ring form): 250 ms
// for the 47 913-node sub-graph's 274 segments - the pole of the batch that holds it and of the rank that owns it.  With the polynomials of the
// strides J, 4 J and 16 J (g^4, g^16 mod phi: utils/mt_jump.py) the starts form a radix-4 tree: one chain of (K - 1) / 16 jumps of 16 segments, then - in
// parallel, one workgroup per base - chains of at most three jumps of 4 segments and of 1 segment: the same K - 1 jumps, (K - 1) / 16 + 6 deep.
// An item = one chain: from the state of segment s0 (seeded and produced here when `init`), `count` jumps of `step` segments with the level's polynomial.
constexpr int MTJ_RING = 3 * MT_N;
constexpr int MTJ_RADIX = 4, MTJ_MAX_LEVELS = 3;
struct MtJumpItem { int32_t t, s0, step, count, init, pad; };
__global__ __launch_bounds__(MTX_THREADS) void k_mt_segment_starts(const TargetMeta* meta, const int64_t* seeds, const long long* seg_off,
                                                                   const uint32_t* poly, long long jump, uint32_t* seg_state, const MtJumpItem* items) {
    __shared__ uint32_t y[MTJ_RING];
    __shared__ uint32_t spoly[MT_N];      // the polynomial, once per workgroup: a jump reads its 624 words one after the other - from global memory each
                                          // word was a dependent ~2 us round trip, 1.2 of the 1.6 ms of a jump (profiles/r06_probe_xl_jump_walk.txt)
    const MtJumpItem it = items[blockIdx.x];
    const int t = it.t;
    const int tid = threadIdx.x;
    const long long n = meta[t].n;
    const int K = mt_segments(n, jump);
    if (K == 0) return;
    if (it.count > 0)
        for (int k = tid; k < MT_N; k += MTX_THREADS) spoly[k] = poly[k];
    // block `nb` of the ring <- the block after block `ob` (three sweeps: the recurrence's dependency distance is 227 words)
    auto next_block = [&](int ob, int nb) {
        const uint32_t* o = y + ob * MT_N;
        uint32_t* nw = y + nb * MT_N;
        if (tid < MT_N - MT_M) nw[tid] = o[tid + MT_M] ^ mt_twist(o[tid], o[tid + 1]);
        __syncthreads();
        if (tid < MT_N - MT_M) {
            const int k = tid + (MT_N - MT_M);
            nw[k] = nw[k - (MT_N - MT_M)] ^ mt_twist(o[k], o[k + 1]);
        }
        __syncthreads();
        if (tid < MT_N - 2 * (MT_N - MT_M)) {
            const int k = tid + 2 * (MT_N - MT_M);
            nw[k] = nw[k - (MT_N - MT_M)] ^ mt_twist(o[k], k + 1 < MT_N ? o[k + 1] : nw[0]);
        }
        __syncthreads();
    };
    uint32_t* out = seg_state + ((size_t)seg_off[t] + it.s0) * MT_N;
    int jcl[3];            // this thread's three window words, clamped to the window
#pragma unroll
    for (int q = 0; q < 3; ++q) jcl[q] = (tid + q * MTX_THREADS < MT_N) ? tid + q * MTX_THREADS : MT_N - 1;
    uint32_t win[3];       // the current window (segment start), three words per thread
    if (it.init) {
        if (tid == 0) {   // at::mt19937::init_with_uint32
            uint32_t x = (uint32_t)((unsigned long long)seeds[t] & 0xffffffffull);
            y[0] = x;
            for (int j = 1; j < MT_N; ++j) {
                x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j;
                y[j] = x;
            }
        }
        __syncthreads();
        next_block(0, 1);      // the first block: draws [0, 624) = x_624 .. x_1247
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = tid + q * MTX_THREADS;
            win[q] = (j < MT_N) ? y[MT_N + j] : 0u;
            if (j < MT_N) out[j] = win[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {      // the chain's first state: written by the level above (an earlier launch on this stream)
            const int j = tid + q * MTX_THREADS;
            win[q] = (j < MT_N) ? out[j] : 0u;
        }
    }
    __syncthreads();
    for (int s = 1; s <= it.count; ++s) {
        // ring block 0 <- the window, block 1 <- its successor
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = tid + q * MTX_THREADS;
            if (j < MT_N) y[j] = win[q];
        }
        __syncthreads();
        next_block(0, 1);
        uint32_t acc[3] = {0u, 0u, 0u};
        int b0 = 0;        // ring position of the block that holds the terms i of this round
        for (int blk = 0; blk * MT_N < MT_DEG; ++blk) {
            const int b1 = (b0 + 1) % 3, b2 = (b0 + 2) % 3;
            // terms i in [624 blk, 624 (blk + 1)): word j of the window at i is ring[(b0 624 + (i - 624 blk) + j) mod 1872] (blocks b0, b1)
            const int ilo = blk * MT_N, ihi = (ilo + MT_N < MT_DEG) ? ilo + MT_N : MT_DEG;
            uint32_t gnext = spoly[ilo >> 5];
            for (int w = ilo >> 5; 32 * w < ihi; ++w) {
                uint32_t gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)gnext);      // uniform: the branches below are scalar
                gnext = spoly[(w + 1 < MT_N) ? w + 1 : w];                                // (the next word's LDS read overlaps this word's terms)
                // (a polynomial word may straddle two rounds: 624 is not a multiple of 32 - mask the bits of the other round)
                const int lo = (32 * w < ilo) ? ilo - 32 * w : 0, hi = (32 * w + 32 > ihi) ? ihi - 32 * w : 32;
                if (lo > 0) gw &= ~0u << lo;
                if (hi < 32) gw &= (1u << hi) - 1u;
                while (gw) {                                                           // uniform loop over the set bits, four per trip: one wave per
                    int base[4];                                                       // SIMD has nothing to hide an LDS round trip behind but its own
#pragma unroll                                                                         // independent loads - twelve in flight instead of three
                    for (int u = 0; u < 4; ++u) {
                        base[u] = -1;
                        if (gw) {
                            base[u] = b0 * MT_N + (32 * w + (__ffs((int)gw) - 1) - ilo);
                            gw &= gw - 1u;
                        }
                    }
                    // every load unconditional (a predicated load becomes an exec-mask region with its own wait and branch - twelve of them per trip made
                    // a jump 1.5 ms): the lanes of the third column group beyond word 623 read word 623's slot again, their sums are never stored
                    uint32_t v[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            int idx = (base[u] < 0 ? 0 : base[u]) + jcl[q];
                            idx -= (idx >= MTJ_RING) ? MTJ_RING : 0;
                            v[u][q] = y[idx];
                        }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t m = base[u] >= 0 ? 0xffffffffu : 0u;            // uniform
#pragma unroll
                        for (int q = 0; q < 3; ++q) acc[q] ^= v[u][q] & m;
                    }
                }
            }
            __syncthreads();                     // every read of block b0 is done: it becomes the block after b1's successor
            if ((blk + 1) * MT_N < MT_DEG) next_block(b1, b2), b0 = b1;
            // (the ring now holds blocks blk + 1 (b0) and blk + 2; block blk + 2 was generated from blk + 1 - into the slot block blk left)
        }
        out += (size_t)it.step * MT_N;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int j = tid + q * MTX_THREADS;
            win[q] = acc[q];
            if (j < MT_N) out[j] = acc[q];
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(MTX_THREADS) void k_mt_edge_words_seg(const TargetMeta* meta, const MtSeg* segs, const long long* seg_off, long long jump,
                                                                   const uint32_t* seg_state, const long long* csr_off, const int32_t* rowptr,
                                                                   const int32_t* uprow, const int32_t* col, const int32_t* row, const long long* eoff,
                                                                   uint32_t* words) {
    __shared__ uint32_t st[2][MT_N];
    __shared__ int s_blk[MTX_CHUNK], s_off[MTX_CHUNK];
    __shared__ long long s_out[MTX_CHUNK];
    __shared__ uint32_t tailw[16];
    const MtSeg sg = segs[blockIdx.x];
    const int t = sg.t;
    const TargetMeta tm = meta[t];
    const int tid = threadIdx.x;
    const long long n = tm.n, nn = n * n;
    if (nn < 16) return;                       // the host draws such a target whole (ATen's scalar path)
    const long long reg_end = (nn & 15) ? nn - 16 : nn;
    const long long total = nn + ((nn & 15) ? 16 : 0);
    const int K = mt_segments(n, jump);
    const long long d_begin = (long long)sg.s * jump, d_end = (sg.s == K - 1) ? total : d_begin + jump;
    const bool last_seg = sg.s == K - 1;
    const int32_t* rp = rowptr + csr_off[2 * t];
    const int32_t* ur = uprow + csr_off[2 * t];
    const int32_t* cl = col + csr_off[2 * t + 1];
    const int32_t* rw = row + csr_off[2 * t + 1];
    const int nnz = rp[tm.ld];
    uint32_t* wout = words + 4 * eoff[t];
    // the position key of directed entry e: the first draw of its 16-value group, or nn for the entries of the redrawn tail (ascending in e)
    auto key = [&](int e) -> long long {
        const long long p = (long long)rw[e] * n + cl[e];
        return p < reg_end ? (p & ~15ll) : nn;
    };
    auto first_at = [&](long long d) {        // first entry whose key is >= d
        int lo = 0, hi = nnz;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (key(mid) < d) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    // this segment's entries: keys in [d_begin, d_end) - the last segment also takes the tail entries (key nn < total)
    const int e_lo = first_at(d_begin);
    const int e_hi = last_seg ? nnz : first_at(d_end);
    const uint32_t* v0 = seg_state + ((size_t)seg_off[t] + sg.s) * MT_N;
    for (int k = tid; k < MT_N; k += MTX_THREADS) st[0][k] = v0[k];
    // destination of directed entry e = (i, j): word pair 2 dir of edge q (the edge's index among the target's upper-triangle edges)
    auto dest = [&](int e, int i, int j) -> long long {
        if (j > i) {
            const int firstup = rp[i + 1] - (ur[i + 1] - ur[i]);
            return 4ll * (ur[i] + (e - firstup));
        }
        int lo = rp[j], hi = rp[j + 1];
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cl[mid] < i) lo = mid + 1; else hi = mid;
        }
        const int firstup = rp[j + 1] - (ur[j + 1] - ur[j]);
        return 4ll * (ur[j] + (lo - firstup)) + 2;
    };
    int ec = e_lo, ci = 0, cn = 0;      // next entry to stage, cursor / fill of the staged chunk (uniform)
    auto stage = [&]() {                // the next MTX_CHUNK entries -> LDS (block-uniform call)
        __syncthreads();
        const int e = ec + tid;
        int blk = 0x7fffffff, off = 0;
        long long out = 0;
        if (e < e_hi) {
            const int i = rw[e], j = cl[e];
            const long long p = (long long)i * n + j;
            if (p < reg_end) {
                const long long base = p & ~15ll;
                blk = (int)(base / MT_N);
                off = (int)(base % MT_N) + (int)(p & 7);
            } else {
                blk = 0x7ffffffe;    // the redrawn tail: handed out after the walk
                off = (int)(p - (nn - 16)) & 7;
            }
            out = dest(e, i, j);
        }
        s_blk[tid] = blk;
        s_off[tid] = off;
        s_out[tid] = out;
        cn = (e_hi - ec < MTX_CHUNK) ? e_hi - ec : MTX_CHUNK;
        ec += cn;
        ci = 0;
        __syncthreads();
    };
    stage();
    int cur = 0;
    int b = (int)(d_begin / MT_N);
    bool have = true;                  // the segment's start state IS its first block (draws [d_begin, d_begin + 624))
    for (long long d0 = d_begin; d0 < d_end; d0 += MT_N, ++b) {
        if (!have) {
            const uint32_t* o = st[cur];
            uint32_t* nw = st[cur ^ 1];
            if (tid < MT_N - MT_M) nw[tid] = o[tid + MT_M] ^ mt_twist(o[tid], o[tid + 1]);
            __syncthreads();
            if (tid < MT_N - MT_M) {
                const int k = tid + (MT_N - MT_M);
                nw[k] = nw[k - (MT_N - MT_M)] ^ mt_twist(o[k], o[k + 1]);
            }
            __syncthreads();
            if (tid < MT_N - 2 * (MT_N - MT_M)) {
                const int k = tid + 2 * (MT_N - MT_M);      // 454 .. 623
                nw[k] = nw[k - (MT_N - MT_M)] ^ mt_twist(o[k], k + 1 < MT_N ? o[k + 1] : nw[0]);
            }
            __syncthreads();
            cur ^= 1;
        }
        have = false;
        const uint32_t* nw = st[cur];
        // the redrawn tail's 16 words (draws nn .. nn + 15) as they pass
        if (last_seg && (nn & 15) && tid < 16) {
            const long long d = nn + tid - d0;
            if (d >= 0 && d < MT_N) tailw[tid] = nw[d];
        }
        // the entries whose pair lies in this block (uniform control flow: ci / cn / ec are the same in every thread)
        while (ci < cn && s_blk[ci] == b) {
            const int k = ci + tid;
            const bool mine = k < cn && s_blk[k] == b;
            if (mine) {
                wout[s_out[k]] = nw[s_off[k]];
                wout[s_out[k] + 1] = nw[s_off[k] + 8];
            }
            // entries of one block are contiguous: the first staged entry beyond it (a short scan over LDS, the same in every thread)
            int nx = ci + 1;
            while (nx < cn && s_blk[nx] == b) ++nx;
            ci = nx;
            if (ci == cn && ec < e_hi) stage();      // the block's entries may continue in the next chunk
        }
    }
    __syncthreads();
    if (!last_seg) return;
    // the entries of the last 16 stream positions: their words come from the redrawn tail
    for (;;) {
        const int k = ci + tid;
        if (k < cn && s_blk[k] == 0x7ffffffe) {
            wout[s_out[k]] = tailw[s_off[k]];
            wout[s_out[k] + 1] = tailw[s_off[k] + 8];
        }
        if (ec >= e_hi) break;
        stage();
    }
}

}  // namespace gnnx
