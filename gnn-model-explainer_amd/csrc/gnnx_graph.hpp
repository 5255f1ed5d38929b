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
    int32_t* sizes;          // [T] out (size pass; optional in the emit pass)
    const int64_t* nb_off;   // [T + 1] (emit pass): the prefix sums of the sizes, or any offsets that leave room for every list
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
// LDSW = words per bitmap kept in LDS (0: the bitmaps live in global scratch).  Two sizes: 4096 (graphs of up to 131 072 nodes, 48 KB) and -
// round 5 - 256 (up to 8192 nodes, 3 KB): a pipelined job's k-hop pass has to find room on CUs that optimisation workgroups fill for 2.4 ms at
// a time; with 48 KB of LDS three of its workgroups fit a freed CU, with 3 KB as many as its wave slots allow.
template <bool EMIT, int LDSW>
__global__ __launch_bounds__(KH_THREADS) GNNX_SERVICE_ATTR void k_khop(KhopArgs a) {
    constexpr bool IN_LDS = LDSW > 0;
    __shared__ uint32_t s_bm[IN_LDS ? 3 * LDSW : 1];
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
            // node_idx_new falls out of the size pass as well (the reached bits below the target's own; explain.py:496), so the host learns
            // sizes AND rows from ONE copy and the emit pass has nothing to report back
            const int vw = v >> 5;
            if (a.target_row && vw >= w0 && vw < w1) {
                int pos = base;
                for (int w = w0; w < vw; ++w) pos += __popc(reach[w]);
                const uint32_t bits = reach[vw], bit = 1u << (v & 31);
                a.target_row[t] = (bits & bit) ? pos + __popc(bits & (bit - 1u)) : -1;
            }
        } else {
            if (tid == 0 && a.sizes) a.sizes[t] = total;   // one-pass use: lists at caller-chosen (padded) offsets, sizes reported alongside
            int32_t* out = a.nb + a.nb_off[t];
            int pos = base;
            for (int w = w0; w < w1; ++w) {
                uint32_t bits = reach[w];
                if (a.target_row && w == (v >> 5)) {  // the target's own position (explain.py:496: number of neighbours with a smaller id)
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

// ---- edge lists: upper-triangle non-zeros (c > r) of every target's block of the packed adjacency, row-major ----
// Three small kernels over the 32-row blocks of the batch (so a 4000-node target is scanned by 125 workgroups, not one):
//   k_edge_rowcount  per row: number of upper-triangle non-zeros (wave ballots over 64-column chunks) -> rowcnt[R]
//   k_edge_rowscan   per target: exclusive scan of its rows' counts (in place) and the total -> counts[t]
//   k_edge_emit      per row: (r, c) pairs, values and positions at eoff[t] + rowstart[r] + rank within the row
struct EdgeOut {
    const int64_t* eoff;  // [T + 1] prefix sums of the per-target counts
    int32_t* rc;          // [E][2]
    float* abar;          // [E]    or null
    float* m_rc;          // [E][2] or null: M[r][c], M[c][r]
    int32_t* rowcnt;      // scratch [R] (one int per row of the batch): counts, then row starts
    int64_t* epos;        // [E][2] or null: float index of (r, c) and of (c, r) in the packed square arrays (for k_gather_values)
};

__global__ __launch_bounds__(256) void k_edge_rowcount(const float* A, const ConvTile* tiles, int32_t* rowcnt) {
    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int rr = wave; rr < TILE; rr += 4) {
        const int r = tl.rb * TILE + rr;
        int cnt = 0;
        if (r < tm.n) {
            const float* row = A + tm.offQ + (size_t)r * tm.ld;
            for (int c0 = (r + 1) & ~63; c0 < tm.n; c0 += 64) {
                const int c = c0 + lane;
                cnt += __popcll(__ballot(c > r && c < tm.n && row[c] != 0.0f));
            }
        }
        if (lane == 0) rowcnt[tm.offR + r] = cnt;
    }
}

__global__ __launch_bounds__(64) void k_edge_rowscan(const TargetMeta* meta, int32_t* rowcnt, int64_t* counts) {
    const TargetMeta tm = meta[blockIdx.x];
    const int lane = threadIdx.x;
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
    if (lane == 0 && counts) counts[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_edge_emit(const float* A, const float* Abar, const float* M, const ConvTile* tiles, EdgeOut o) {
    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t e0 = o.eoff[tl.t];
    for (int rr = wave; rr < TILE; rr += 4) {
        const int r = tl.rb * TILE + rr;
        if (r >= tm.n) continue;
        const float* row = A + tm.offQ + (size_t)r * tm.ld;
        int pos = o.rowcnt[tm.offR + r];
        for (int c0 = (r + 1) & ~63; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            const bool nz = c > r && c < tm.n && row[c] != 0.0f;
            const unsigned long long bal = __ballot(nz);
            if (nz) {
                const int64_t e = e0 + pos + __popcll(bal & ((1ull << lane) - 1ull));
                const int64_t prc = tm.offQ + (int64_t)r * tm.ld + c, pcr = tm.offQ + (int64_t)c * tm.ld + r;
                o.rc[2 * e] = r;
                o.rc[2 * e + 1] = c;
                if (o.epos) {
                    o.epos[2 * e] = prc;
                    o.epos[2 * e + 1] = pcr;
                }
                if (o.abar) o.abar[e] = Abar[prc];
                if (o.m_rc) {
                    o.m_rc[2 * e] = M[prc];
                    o.m_rc[2 * e + 1] = M[pcr];
                }
            }
            pos += __popcll(bal);
        }
    }
}

// The edge structure of a batch is fixed: once k_edge_emit has recorded where every edge lives (epos), the values of a
// new run are one indexed read each.
__global__ __launch_bounds__(256) void k_gather_values(const int64_t* epos, int64_t E, const float* Abar, const float* M, float* abar,
                                                       float* m_rc) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    const int64_t a = epos[2 * e], b = epos[2 * e + 1];
    if (abar) abar[e] = Abar[a];
    if (m_rc) {
        m_rc[2 * e] = M[a];
        m_rc[2 * e + 1] = M[b];
    }
}


// ---------------------------------------------------------------------------------------------
// The seeded initial masks without the host walking their streams (round 5).  construct_edge_mask draws the n x n mask of a target with
// ONE normal_ call from a torch CPU generator seeded per target (explain.py:645-652; the seed protocol of the golden runs), and the
// edge-sparse kernels read that draw on the EDGES only.  ATen's normal_ of >= 16 values (normal_fill) is one 32-bit engine draw per value
// followed by a Box-Muller transform of the pairs (j, j + 8) of every 16 values (and, for a ragged stream, a redraw of the last 16 values
// from the 16 draws that follow the fill).  The transform cannot be restated on a device bit for bit (DESIGN: it is whatever libm the
// host's ATen build calls), but the ENGINE is plain integer arithmetic: mt19937, whose state after d draws is the seeded state after
// ceil(d / 624) block updates.  So the device walks every target's engine (k_mt_stream: one workgroup per target, the 624-word state in
// LDS, a block update = three data-parallel sweeps - words [0, 227) depend on the old block only, [227, 454) on those, [454, 624) on
// those) and writes the RAW state words of draws [0, n^2 (+ 16)) into the target's block of a scratch array (the Abar array: unused
// before the run); k_mt_gather_pairs then picks, for every directed edge entry, the two words of its Box-Muller pair; only those
// 16 bytes per entry cross PCIe, and the host lets ATen temper and transform exactly them (gnnx_host_transform_edge_words: the pair
// staging of gnnx_host_draw_edge_masks).  Per 16 384-target BA-House x100k batch the host walked 1.0e9 draws (0.4 core-seconds, what
// made eight ranks of a node host-bound on a 16-core quota); now it transforms 8 M pairs.
// ---------------------------------------------------------------------------------------------
constexpr int MT_N = 624, MT_M = 397;
__device__ __forceinline__ uint32_t mt_twist(uint32_t u, uint32_t v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
// stream[offQ[t] + d] = raw state word of draw d of target t's engine, d < n^2 (+ 16 when n^2 is not a multiple of 16: the draws of the
// redrawn tail); targets with fewer than 16 values are left to the host (ATen's scalar path).  seeds [T] int64 (at::mt19937(seed): the low 32 bits).
__global__ __launch_bounds__(256) void k_mt_stream(const TargetMeta* meta, const int64_t* seeds, uint32_t* stream) {
    __shared__ uint32_t st[2][MT_N];
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x;
    const long long nn = (long long)tm.n * tm.n;
    if (nn < 16) return;
    const long long total = nn + ((nn & 15) ? 16 : 0);
    if (tid == 0) {   // at::mt19937::init_with_uint32
        uint32_t x = (uint32_t)((unsigned long long)seeds[blockIdx.x] & 0xffffffffull);
        st[0][0] = x;
        for (int j = 1; j < MT_N; ++j) {
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j;
            st[0][j] = x;
        }
    }
    __syncthreads();
    uint32_t* out = stream + tm.offQ;
    int cur = 0;
    for (long long d0 = 0; d0 < total; d0 += MT_N) {
        const uint32_t* o = st[cur];
        uint32_t* nw = st[cur ^ 1];
        // next_state of MT19937RNGEngine.h (the host's mt_block_update), old block -> new block
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
        for (int k = tid; k < MT_N; k += 256)
            if (d0 + k < total) out[d0 + k] = nw[k];
    }
}

// words[e] = {w(j), w(j + 8)} of entry (r, c), then of entry (c, r), for every upper-triangle edge e of the batch (rc [E][2], eoff [T + 1]):
// the raw words of the Box-Muller pair that holds the entry's value - pair j = lane & 7 of the 16 values around position p = r n + c, or
// of the redrawn tail (draws [n^2, n^2 + 16)) for the last 16 positions of a ragged stream.  One workgroup per target.
__global__ __launch_bounds__(256) void k_mt_gather_pairs(const TargetMeta* meta, const int64_t* eoff, const int32_t* rc, const uint32_t* stream,
                                                         uint32_t* words) {
    const TargetMeta tm = meta[blockIdx.x];
    const long long n = tm.n, nn = n * n;
    if (nn < 16) return;
    const long long reg_end = (nn & 15) ? nn - 16 : nn;
    const uint32_t* src = stream + tm.offQ;
    for (long long e = eoff[blockIdx.x] + threadIdx.x; e < eoff[blockIdx.x + 1]; e += 256) {
        const long long r = rc[2 * e], c = rc[2 * e + 1];
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
            const long long p = dir ? c * n + r : r * n + c;
            const long long base = p < reg_end ? (p & ~15ll) : nn;
            const int lane = (int)(p < reg_end ? (p & 15) : p - (nn - 16));
            words[4 * e + 2 * dir] = src[base + (lane & 7)];
            words[4 * e + 2 * dir + 1] = src[base + (lane & 7) + 8];
        }
    }
}

}  // namespace gnnx

namespace gnnx {

// ---------------------------------------------------------------------------------------------
// Post-processing of the explanations on the device (explain_nodes_gnn_stats / explain_graphs, explain.py:306-351, with
// io_utils.denoise_graph, utils/io_utils.py:193-245), on the edge lists of k_gather_edges:
//   k_denoise     per target: the threshold that keeps the `threshold_num` heaviest undirected edges (ties kept, as the
//                 reference's `adj >= np.sort(adj[adj > 0])[-2 threshold_num]`), then the largest connected component of
//                 the kept edges over all n nodes (first one in node order among equals, as max(nx.connected_components));
//   k_auc_*       ROC-AUC of all targets' edge scores against the motif ground truth as exact pair counts
//                 (#{pos > neg}, #{pos == neg}): AUC = (gt + eq / 2) / (P N) - what sklearn's trapezoid rule evaluates.
// ---------------------------------------------------------------------------------------------
struct DenoiseArgs {
    const TargetMeta* meta;
    const int64_t* eoff;   // [T + 1]
    const int32_t* rc;     // [E][2]
    const float* vals;     // [E] masked adjacency on the upper-triangle edges
    int32_t threshold_num; // undirected edges to keep (the reference's threshold_num; it doubles it for the symmetric matrix)
    uint8_t* keep;         // [E] out: edge belongs to the denoised explanation
    float* threshold;      // [T] out (0 when the target has no positive edge)
    int32_t* stats;        // [T][3] out: nodes and edges of the kept component, its smallest node id
    int32_t* comp;         // scratch [R]: component label of every node
    int32_t* cnt;          // scratch [R]: nodes per label
};

// comp / cnt live in global memory and are updated with atomics (performed at L2): read them past this CU's L1
__device__ __forceinline__ int dn_load(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float dn_block_max(float v, float* s4) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
}
__device__ __forceinline__ int dn_block_sum(int v, int* s4) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

__global__ __launch_bounds__(256) void k_denoise(DenoiseArgs a) {
    __shared__ float sf[4];
    __shared__ int si[4];
    __shared__ int s_changed;
    const int t = blockIdx.x, tid = threadIdx.x;
    const TargetMeta tm = a.meta[t];
    const int64_t e0 = a.eoff[t];
    const int E = (int)(a.eoff[t + 1] - e0);
    const int32_t* rc = a.rc + 2 * e0;
    const float* v = a.vals + e0;
    uint8_t* keep = a.keep + e0;
    int32_t* comp = a.comp + tm.offR;
    int32_t* cnt = a.cnt + tm.offR;
    // ---- threshold: the k-th largest positive value with multiplicity (every undirected edge counts once) ----
    float cur = __builtin_inff(), thr = 0.0f;
    int remaining = a.threshold_num;
    bool any = false;
    for (int it = 0; it < a.threshold_num; ++it) {   // at most k distinct values are visited
        float m = 0.0f;
        for (int e = tid; e < E; e += 256) {
            const float x = v[e];
            if (x > 0.0f && x < cur) m = fmaxf(m, x);
        }
        m = dn_block_max(m, sf);
        if (m <= 0.0f) break;                       // fewer than k positive edges: keep them all (threshold = the smallest)
        any = true;
        thr = m;
        int c = 0;
        for (int e = tid; e < E; e += 256) c += (v[e] == m);
        c = dn_block_sum(c, si);
        if (c >= remaining) break;
        remaining -= c;
        cur = m;
    }
    if (tid == 0) a.threshold[t] = any ? thr : 0.0f;
    // ---- connected components of the kept edges by min-label propagation ----
    for (int i = tid; i < tm.n; i += 256) {
        comp[i] = i;
        cnt[i] = 0;
    }
    for (int e = tid; e < E; e += 256) keep[e] = (any && v[e] >= thr && v[e] > 0.0f) ? 1 : 0;
    __syncthreads();
    for (int round = 0; round < tm.n; ++round) {
        if (tid == 0) s_changed = 0;
        __syncthreads();
        for (int e = tid; e < E; e += 256)
            if (keep[e]) {
                const int r = rc[2 * e], c = rc[2 * e + 1];
                const int x = dn_load(&comp[r]), y = dn_load(&comp[c]);
                if (x != y) {
                    const int m = x < y ? x : y;
                    atomicMin(&comp[r], m);
                    atomicMin(&comp[c], m);
                    s_changed = 1;
                }
            }
        __syncthreads();
        const int ch = s_changed;
        __syncthreads();
        if (!ch) break;
    }
    // labels are not yet roots everywhere (a node keeps the smallest label it has seen): chase them to the fixed point
    for (int i = tid; i < tm.n; i += 256) {
        int l = dn_load(&comp[i]);
        for (int nx = dn_load(&comp[l]); nx != l; nx = dn_load(&comp[l])) l = nx;
        atomicMin(&comp[i], l);          // labels only ever decrease: concurrent chasers stay consistent
    }
    __syncthreads();
    for (int i = tid; i < tm.n; i += 256) atomicAdd(&cnt[dn_load(&comp[i])], 1);
    __syncthreads();
    // largest component; among equals the one with the smallest label (= smallest node id: the first nx would yield)
    int best = 0, bl = 0x7fffffff;
    for (int i = tid; i < tm.n; i += 256) {
        const int c = dn_load(&cnt[i]);
        if (c > best || (c == best && c > 0 && i < bl)) {
            best = c;
            bl = i;
        }
    }
    {
        __shared__ int sb[256], sl[256];
        sb[tid] = best;
        sl[tid] = bl;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (tid < s) {
                const int c = sb[tid + s], l = sl[tid + s];
                if (c > sb[tid] || (c == sb[tid] && l < sl[tid])) {
                    sb[tid] = c;
                    sl[tid] = l;
                }
            }
            __syncthreads();
        }
        best = sb[0];
        bl = sl[0];
    }
    int ne = 0;
    for (int e = tid; e < E; e += 256) {
        const bool k = keep[e] && dn_load(&comp[rc[2 * e]]) == bl;
        keep[e] = k ? 1 : 0;
        ne += k;
    }
    ne = dn_block_sum(ne, si);
    if (tid == 0) {
        a.stats[3 * t] = best;
        a.stats[3 * t + 1] = ne;
        a.stats[3 * t + 2] = bl;
    }
}

// positives (real == 1) compacted into pos[] (order irrelevant for counting); counts[0] = P
__global__ __launch_bounds__(256) void k_auc_compact(const float* vals, const uint8_t* real, int64_t E, float* pos, unsigned long long* counts) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < E && real[e]) pos[atomicAdd(&counts[0], 1ull)] = vals[e];
}
// every negative against every positive: counts[1] += #{pos > neg}, counts[2] += #{pos == neg}, counts[3] += negatives
__global__ __launch_bounds__(256) void k_auc_count(const float* vals, const uint8_t* real, int64_t E, const float* pos,
                                                   unsigned long long* counts) {
    __shared__ float sp[1024];
    __shared__ unsigned long long sred[3][4];
    const unsigned long long P = counts[0];
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool neg = e < E && !real[e];
    const float x = neg ? vals[e] : 0.0f;
    unsigned long long gt = 0, eq = 0;
    for (unsigned long long p0 = 0; p0 < P; p0 += 1024) {
        const int chunk = (int)((P - p0 < 1024ull) ? (P - p0) : 1024ull);
        __syncthreads();
        for (int k = threadIdx.x; k < chunk; k += 256) sp[k] = pos[p0 + k];
        __syncthreads();
        if (neg)
            for (int k = 0; k < chunk; ++k) {
                gt += (sp[k] > x);
                eq += (sp[k] == x);
            }
    }
    unsigned long long r[3] = {gt, eq, neg ? 1ull : 0ull};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        unsigned lo = (unsigned)r[j], hi = (unsigned)(r[j] >> 32);   // hi stays 0 for any realistic E; summed separately
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned lo2 = (unsigned)__shfl_xor((int)lo, o), hi2 = (unsigned)__shfl_xor((int)hi, o);
            const unsigned long long s = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)hi2 << 32) | lo2);
            lo = (unsigned)s;
            hi = (unsigned)(s >> 32);
        }
        if ((threadIdx.x & 63) == 0) sred[j][threadIdx.x >> 6] = ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(&counts[1 + threadIdx.x], sred[threadIdx.x][0] + sred[threadIdx.x][1] + sred[threadIdx.x][2] + sred[threadIdx.x][3]);
}

}  // namespace gnnx
