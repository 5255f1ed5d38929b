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
__global__ __launch_bounds__(1024) void k_xl_rowptr(const TargetMeta* meta, const int32_t* deg, const int32_t* updeg, const long long* rp_off,
                                                    int32_t* rowptr, int32_t* uprow, int32_t* totals) {
    constexpr int NT = 1024, NW = NT / 64;
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
// The seeded initial masks of XL targets with the engine walked on the DEVICE and NO n^2 scratch (round 6).  gnnx_mt_edge_words (gnnx_graph.hpp)
// writes the raw mt19937 state words of all n^2 draws of a target into its dense block and gathers the pairs afterwards - 9.6 GB of words for
// a 49 028-node target.  The directed entries of a sub-graph CSR are ASCENDING in their stream position p = r n + c (row-major rows, ascending
// columns), so one pass suffices: the workgroup walks the engine block by block (624 words per update, three data-parallel sweeps) and hands
// every entry whose Box-Muller pair lies in the current block its two raw words on the spot.  Entries are staged 256 at a time in LDS (block
// index, word offset, destination 4 q + 2 dir in the edge-list layout of gnnx_host_transform_edge_words); a block without entries - nine out
// of ten on BA-House x100k - costs one LDS compare beside its update.  The last 16 values of a ragged stream are redrawn from the 16 draws that
// follow the fill (ATen's normal_fill): those words are kept as they pass and handed out at the end.
// words [E][4] (uint32): for every upper-triangle edge (r, c): {w(j), w(j + 8)} of entry (r, c), then of entry (c, r).
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int MTX_THREADS = 256, MTX_CHUNK = 256;

__global__ __launch_bounds__(MTX_THREADS) void k_mt_edge_words_xl(const TargetMeta* meta, const int64_t* seeds, const long long* csr_off,
                                                                  const int32_t* rowptr, const int32_t* uprow, const int32_t* col, const int32_t* row,
                                                                  const long long* eoff, uint32_t* words) {
    __shared__ uint32_t st[2][MT_N];
    __shared__ int s_blk[MTX_CHUNK], s_off[MTX_CHUNK];
    __shared__ long long s_out[MTX_CHUNK];
    __shared__ uint32_t tailw[16];
    const int t = blockIdx.x;
    const TargetMeta tm = meta[t];
    const int tid = threadIdx.x;
    const long long n = tm.n, nn = n * n;
    if (nn < 16) return;                       // the host draws such a target whole (ATen's scalar path)
    const long long reg_end = (nn & 15) ? nn - 16 : nn;
    const int32_t* rp = rowptr + csr_off[2 * t];
    const int32_t* ur = uprow + csr_off[2 * t];
    const int32_t* cl = col + csr_off[2 * t + 1];
    const int32_t* rw = row + csr_off[2 * t + 1];
    const int nnz = rp[tm.ld];
    uint32_t* wout = words + 4 * eoff[t];
    if (tid == 0) {   // at::mt19937::init_with_uint32
        uint32_t x = (uint32_t)((unsigned long long)seeds[t] & 0xffffffffull);
        st[0][0] = x;
        for (int j = 1; j < MT_N; ++j) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j;
            st[0][j] = x;
        }
    }
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
    int ec = 0, ci = 0, cn = 0;      // next entry to stage, cursor / fill of the staged chunk (uniform)
    auto stage = [&]() {             // the next MTX_CHUNK entries -> LDS (block-uniform call)
        __syncthreads();
        const int e = ec + tid;
        int blk = 0x7fffffff, off = 0;
        long long out = 0;
        if (e < nnz) {
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
        cn = (nnz - ec < MTX_CHUNK) ? nnz - ec : MTX_CHUNK;
        ec += cn;
        ci = 0;
        __syncthreads();
    };
    stage();
    const long long total = nn + ((nn & 15) ? 16 : 0);
    int cur = 0;
    int b = 0;
    for (long long d0 = 0; d0 < total; d0 += MT_N, ++b) {
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
        // the redrawn tail's 16 words (draws nn .. nn + 15) as they pass
        if ((nn & 15) && tid < 16) {
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
            if (ci == cn && ec < nnz) stage();      // the block's entries may continue in the next chunk
        }
    }
    __syncthreads();
    // the entries of the last 16 stream positions: their words come from the redrawn tail
    for (;;) {
        const int k = ci + tid;
        if (k < cn && s_blk[k] == 0x7ffffffe) {
            wout[s_out[k]] = tailw[s_off[k]];
            wout[s_out[k] + 1] = tailw[s_off[k] + 8];
        }
        if (ec >= nnz) break;
        stage();
    }
}

}  // namespace gnnx
