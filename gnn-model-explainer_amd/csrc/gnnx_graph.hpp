// gnnx_graph.hpp — the integer / index work either side of the optimisation loop, on the device:
//   k_khop          k-hop walk sets of a batch of targets over the resident CSR graph, as ascending id lists
//                   (replaces graph_utils.neighborhoods, utils/graph_utils.py:147-158 - a dense O(N^3) product on the
//                   whole graph - and the row lookup of Explainer.extract_neighborhood, explain.py:492-497);
//   k_scatter_masks initial edge masks from the host's raw RNG stream (one contiguous n x n draw per target,
//                   explain.py:645-652) into the padded layout;
//   k_edge_counts / k_gather_edges   the returned masks as edge lists (they are exactly zero off the edges,
//                   explain.py:209-211): upper-triangle entries of every target in row-major order.
// All of it is HBM / LDS-bound integer and byte work: bitmaps in LDS, ballots and popcounts, coalesced row scans.
#pragma once
#include "gnnx_kernels.hpp"

namespace gnnx {

constexpr int KH_THREADS = 512;
constexpr int KH_LDS_WORDS = 4096;   // per bitmap: graphs of up to 131072 nodes keep their three bitmaps in LDS (48 KB)

struct KhopArgs {
    const int64_t* indptr;   // CSR of the full graph [N + 1]
    const int32_t* indices;  // [nnz]
    int32_t num_nodes, n_hops;
    const int32_t* targets;  // [T] node ids
    int32_t num_targets;
    int32_t* sizes;          // [T] out (size pass)
    const int64_t* nb_off;   // [T + 1] (emit pass)
    int32_t* nb;             // concatenated ascending neighbour lists (emit pass)
    int32_t* target_row;     // [T] out (emit pass): position of the target in its own list, -1 if absent
    uint32_t* scratch;       // global bitmaps for graphs beyond the LDS capacity: gridDim.x * 3 * words
    int32_t words;           // ceil(N / 32)
};

__device__ __forceinline__ int kh_block_scan_exclusive(int v, int tid, int* s_wave, int* total) {
    // exclusive prefix sum over the KH_THREADS threads of the workgroup
    const int lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl(incl, lane - d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < KH_THREADS / 64; ++w) {
        const int c = s_wave[w];
        if (w < wave) base += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

// One workgroup per target (grid-stride over the batch).  Walk semantics of the reference: the set is
// {u : (A + A^2 + ... + A^k)[v][u] > 0}, i.e. every node reached by a walk of length 1..k - v itself only through a
// walk back to it (an isolated node has an EMPTY neighbourhood).  Level-synchronous expansion over three bitmaps
// (reached, current frontier, next frontier): a node is expanded at the first level it is reached at, which covers
// every later visit.  EMIT = false: only the sizes (the host needs them to lay out the batch); EMIT = true: the walk is
// repeated and the reached bitmap is written out as an ascending id list (a bitmap scan IS a sort).
template <bool EMIT, bool IN_LDS>
__global__ __launch_bounds__(KH_THREADS) void k_khop(KhopArgs a) {
    __shared__ uint32_t s_bm[IN_LDS ? 3 * KH_LDS_WORDS : 1];
    __shared__ int s_wave[KH_THREADS / 64];
    const int tid = threadIdx.x, words = a.words;
    uint32_t* reach = IN_LDS ? s_bm : a.scratch + (size_t)blockIdx.x * 3 * words;
    uint32_t* front = reach + words;
    uint32_t* next = front + words;
    for (int t = blockIdx.x; t < a.num_targets; t += gridDim.x) {
        const int v = a.targets[t];
        for (int w = tid; w < words; w += KH_THREADS) {
            reach[w] = 0u;
            next[w] = 0u;
            front[w] = (w == (v >> 5)) ? (1u << (v & 31)) : 0u;
        }
        __syncthreads();
        for (int hop = 0; hop < a.n_hops; ++hop) {
            const bool last = hop + 1 == a.n_hops;
            for (int w = tid; w < words; w += KH_THREADS) {
                uint32_t bits = front[w];
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    const int u = w * 32 + b;
                    const int64_t e1 = a.indptr[u + 1];
                    for (int64_t e = a.indptr[u]; e < e1; ++e) {
                        const int x = a.indices[e];
                        const uint32_t bit = 1u << (x & 31);
                        const uint32_t old = atomicOr(&reach[x >> 5], bit);
                        if (!last && !(old & bit)) atomicOr(&next[x >> 5], bit);
                    }
                }
            }
            __syncthreads();
            if (!last) {
                for (int w = tid; w < words; w += KH_THREADS) {
                    front[w] = next[w];
                    next[w] = 0u;
                }
                __syncthreads();
            }
        }
        // contiguous chunk of words per thread, so that thread order == id order
        const int cw = (words + KH_THREADS - 1) / KH_THREADS;
        const int w0 = tid * cw, w1 = (w0 + cw < words) ? w0 + cw : words;
        int cnt = 0;
        for (int w = w0; w < w1; ++w) cnt += __popc(reach[w]);
        int total;
        const int base = kh_block_scan_exclusive(cnt, tid, s_wave, &total);
        if (!EMIT) {
            if (tid == 0) a.sizes[t] = total;
        } else {
            int32_t* out = a.nb + a.nb_off[t];
            int pos = base;
            for (int w = w0; w < w1; ++w) {
                uint32_t bits = reach[w];
                if (w == (v >> 5)) {  // the target's own position (explain.py:496: number of neighbours with a smaller id)
                    const uint32_t bit = 1u << (v & 31);
                    a.target_row[t] = (bits & bit) ? pos + __popc(bits & (bit - 1u)) : -1;
                }
                while (bits) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    out[pos++] = w * 32 + b;
                }
            }
        }
        __syncthreads();  // the bitmaps are reused by the next target of this workgroup
    }
}

// Initial edge masks: `raw` holds, target after target, the n x n values of ONE normal_ draw each exactly as the torch
// CPU generator produced them (explain.py:645-652); this spreads them over the padded ld x ld blocks (padding = 0).
// One workgroup per 32-row block.
__global__ __launch_bounds__(256) void k_scatter_masks(const float* raw, const int64_t* raw_off, float* M, const ConvTile* tiles) {
    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const float* src = raw + raw_off[tl.t];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int rr = wave; rr < TILE; rr += 4) {
        const int i = tl.rb * TILE + rr;
        float* dst = M + tm.offQ + (size_t)i * tm.ld;
        for (int c = lane; c < tm.ld; c += 64) dst[c] = (i < tm.n && c < tm.n) ? src[(size_t)i * tm.n + c] : 0.0f;
    }
}

// upper-triangle non-zeros (c > r) of every target's block of the packed adjacency -> counts[t]
__global__ __launch_bounds__(256) void k_edge_counts(const TargetMeta* meta, const float* A, int64_t* counts) {
    __shared__ int part[4];
    const TargetMeta tm = meta[blockIdx.x];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int cnt = 0;
    for (int r = wave; r < tm.n; r += 4) {
        const float* row = A + tm.offQ + (size_t)r * tm.ld;
        for (int c0 = (r + 1) & ~63; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            cnt += __popcll(__ballot(c > r && c < tm.n && row[c] != 0.0f));
        }
    }
    if (lane == 0) part[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (int64_t)part[0] + part[1] + part[2] + part[3];
}

// The explanation of every target as an edge list: for the upper-triangle edges (r < c) of its sub-graph, in row-major
// order, the pair (r, c), the masked adjacency of the last forward and the final mask parameters M[r][c], M[c][r].
// One workgroup per target: pass 1 counts per row (ballots), a prefix scan places the rows, pass 2 writes.
struct EdgeOut {
    const int64_t* eoff;  // [T + 1] prefix sums of k_edge_counts
    int32_t* rc;          // [E][2]
    float* abar;          // [E]    or null
    float* m_rc;          // [E][2] or null: M[r][c], M[c][r]
    int32_t* rowcnt;      // scratch [R] (one int per row of the batch)
};
__global__ __launch_bounds__(256) void k_gather_edges(const TargetMeta* meta, const float* A, const float* Abar, const float* M,
                                                      EdgeOut o) {
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int32_t* rowcnt = o.rowcnt + tm.offR;
    for (int r = wave; r < tm.n; r += 4) {
        const float* row = A + tm.offQ + (size_t)r * tm.ld;
        int cnt = 0;
        for (int c0 = (r + 1) & ~63; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            cnt += __popcll(__ballot(c > r && c < tm.n && row[c] != 0.0f));
        }
        if (lane == 0) rowcnt[r] = cnt;
    }
    __syncthreads();
    // exclusive scan of the row counts, 64 rows at a time (wave 0; the running total stays in a register)
    if (wave == 0) {
        int carry = 0;
        for (int r0 = 0; r0 < tm.n; r0 += 64) {
            const int r = r0 + lane;
            const int v = (r < tm.n) ? rowcnt[r] : 0;
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int x = __shfl(incl, lane - d);
                if (lane >= d) incl += x;
            }
            if (r < tm.n) rowcnt[r] = carry + incl - v;
            carry += __shfl(incl, 63);
        }
    }
    __syncthreads();
    const int64_t e0 = o.eoff[blockIdx.x];
    for (int r = wave; r < tm.n; r += 4) {
        const float* row = A + tm.offQ + (size_t)r * tm.ld;
        int pos = rowcnt[r];
        for (int c0 = (r + 1) & ~63; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            const bool nz = c > r && c < tm.n && row[c] != 0.0f;
            const unsigned long long bal = __ballot(nz);
            if (nz) {
                const int64_t e = e0 + pos + __popcll(bal & ((1ull << lane) - 1ull));
                o.rc[2 * e] = r;
                o.rc[2 * e + 1] = c;
                if (o.abar) o.abar[e] = Abar[tm.offQ + (size_t)r * tm.ld + c];
                if (o.m_rc) {
                    o.m_rc[2 * e] = M[tm.offQ + (size_t)r * tm.ld + c];
                    o.m_rc[2 * e + 1] = M[tm.offQ + (size_t)c * tm.ld + r];
                }
            }
            pos += __popcll(bal);
        }
    }
}

}  // namespace gnnx
