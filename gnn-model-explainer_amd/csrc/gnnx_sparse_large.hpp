// gnnx_sparse_large.hpp — the edge-sparse, one-workgroup-per-target optimisation for targets that are too large for
// k_sparse_resident's LDS (512 < n <= 4095 here; node mode).
//
// Same mathematics, row slots, hop pruning, lane mapping and helpers as k_sparse_resident (gnnx_sparse.hpp); what
// changes is where things live:
//   * LDS keeps what every gather chases through - the masked adjacency per directed entry, the sorted column lists,
//     rowptr - and the per-row scalars (norms, labels, g3) and weights;
//   * the row arrays X, U1, U2 (= dZ2), dZ1 are the caller's workspace arrays in HBM / L2 (stride 32 floats), written
//     and re-read by the same workgroup, i.e. through one CU's L1/L2 path (__syncthreads makes them visible);
//   * the mask entries on edges, their Adam moments and the edge weights are gathered once into compact per-edge planes
//     (coalesced; in an otherwise unused transposed workspace array, like the per-edge indices and the slot records)
//     and scattered back into the dense M at the end - random accesses into a 24 MB dense block cost 67 us per iteration
//     on the largest BA-House x100k target, the planes 10;
//   * rows of up to 1024 entries are split into slots of 64 (the BA-House x100k hubs), loops run over rows / edges
//     instead of one item per thread.
// Hop pruning is what makes this affordable: on the BA-House x100k sample a 2460-node sub-graph has a few hundred rows
// within two hops of its target, so the row phases touch a small fraction of the sub-graph and the rest only costs
// its edges' regulariser updates.
#pragma once
#include "gnnx_sparse.hpp"

namespace gnnx {

constexpr int SPL_THREADS = 512;            // 8 waves: two per SIMD leave 256 VGPRs per lane (the 1024-thread build spilled 120)
constexpr int SPL_GATHER_UNROLL = 4;        // entries in flight per lane: the rows come from L2, not LDS
constexpr int SPL_CHUNK = 64;               // entries per row slot
constexpr int SPL_N_MAX = 4095;             // node ids are packed in 12 bits
constexpr int SPL_POOL_FLOATS = 39168;      // 153 KB of LDS
__host__ __device__ constexpr int spl_stage_floats(int D) { return 16 * TILE * (D | 1); }  // a [32][D|1] dZ1 tile per wave

struct SlotRec { unsigned x, y, m0, m1; };  // see k_sparse_large: x = row | nsplit << 12 | wsplit << 17 | first << 22 | rem << 23 | inB << 28, y = e0 | len << 16, m = entries pointing at t or its neighbours

__host__ __device__ inline int sparse_slots_of_c(int deg, int chunk) { return deg <= chunk ? 1 : (deg + chunk - 1) / chunk; }

struct SparseLargeLayout {
    int oRowptr, oArt, oRn1, oRn2, oYhat, oG3, oW, oWp, oStage, oAb, oCol, total;
};
__host__ __device__ inline SparseLargeLayout sparse_large_layout(int ld, int nnz, int D, int H, int C) {
    SparseLargeLayout L;
    int o = 0;
    L.oRowptr = o; o += ld + 1;   // int; the degrees are counted straight into it
    L.oArt = o;    o += ld;       // Art .. G3 (5 ld floats) double as the setup's uint16 temporaries
    L.oRn1 = o;    o += ld;
    L.oRn2 = o;    o += ld;
    L.oYhat = o;   o += ld;
    L.oG3 = o;     o += ld;
    L.oW = o;      o += (D + 2 * H) * 33;
    L.oWp = o;     o += C * 96;
    L.oStage = o;  o += spl_stage_floats(D);
    L.oAb = o;     o += nnz;
    L.oCol = o;    o += (nnz + 1) / 2;
    L.total = o;
    return L;
}
// slots: row slots of 64 entries needed by the rows within two hops of the target (k_count_edges_large); they are
// processed 512 at a time, their records (both row sets) live in a workspace array of 16 ld entries
__host__ __device__ inline bool sparse_large_fits(int n, int ld, int nnz, int slots, int D, int H, int C) {
    return n <= SPL_N_MAX && nnz < 65536 && 7 * (nnz / 2) <= 32 * ld && slots >= 0 && 2 * slots + 64 <= 16 * ld && C <= RES_CMAX &&
           H >= 2 && 10 * ld >= 7 * ld + 2 * SPL_CHUNK + 16 && sparse_large_layout(ld, nnz, D, H, C).total <= SPL_POOL_FLOATS;
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

// csr_*: the targets' CSR structure, built once per plan by k_build_csr_large (scanning a 24 MB dense block with one
// workgroup takes 4 ms per pass - too much to repeat in every launch): rowptr at csr_off[2 t], columns at csr_off[2 t + 1]
template <int DQ, int HQ>
__global__ __launch_bounds__(SPL_THREADS) void k_sparse_large(Params p, const int32_t* targets, const float* adam_tab,
                                                              const int32_t* csr_rowptr, const unsigned short* csr_col,
                                                              const long long* csr_off) {
    constexpr int NT = SPL_THREADS, NW = NT / 64;
    __shared__ float pool[SPL_POOL_FLOATS];
    __shared__ SparseFixed sh;
    const int t = targets[blockIdx.x];
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, ld = tm.ld, tr = tm.t;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    constexpr bool EXACT = (DQ != 16);   // <5, 10>: exactly D = 10, H = O = 20 (compile-time widths); other shapes take <16, 16>
    const int D = EXACT ? 2 * DQ : p.D, H = EXACT ? 2 * HQ : p.H, O = EXACT ? 2 * HQ : p.O, C = p.C;
    const float* Ag = p.A + tm.offQ;
    float* Mg = p.M + tm.offQ;
    float* est = p.UT[2] + tm.offR * FS;  // per-edge planes [7][eup]: M_ij, M_ji, m_ij, m_ji, v_ij, v_ji, weight
    // row arrays in the caller's workspace (stride FS); dZ2 overwrites U2 row by row as in the resident kernel
    const float* gX = p.X + tm.offR * FS;
    float* gU1 = p.U[0] + tm.offR * FS;
    float* gU2 = p.U[1] + tm.offR * FS;
    float* gdZ1 = p.dZ[0] + tm.offR * FS;
    float* gGe = p.dZT[0] + tm.offR * FS;   // dL/dAbar per directed entry (row-side products), nnz <= 32 ld floats
    unsigned* eidx = reinterpret_cast<unsigned*>(p.UT[0] + tm.offR * FS);  // [eup][2]: i | j << 12 | near << 24 | near2 << 25, e_ij | e_ji << 16

    auto fail_nan = [&]() {
        const float qnan = __builtin_nanf("");
        for (int e = tid; e < ld * ld; e += NT) p.Abar[tm.offQ + e] = qnan;
        if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = qnan;
    };
    if (n > SPL_N_MAX || 6 * ld + 1 + (D + 2 * H) * 33 + C * 96 + spl_stage_floats(D) > SPL_POOL_FLOATS) {  // uniform
        fail_nan();
        return;
    }
    // ---------------- setup 1: rowptr from the plan's CSR (the nnz-independent part of the layout comes first) ----------------
    int* rowptr = reinterpret_cast<int*>(pool);
    const int32_t* grp = csr_rowptr + csr_off[2 * t];
    for (int r = tid; r <= ld; r += NT) rowptr[r] = grp[r];
    if (tid == 0) {
        sh.nnz = grp[ld];
        sh.bad = 0;
    }
    __syncthreads();
    const int nnz = sh.nnz;
    if (!sparse_large_fits(n, ld, nnz, 0, D, H, C)) {
        fail_nan();
        return;
    }
    const SparseLargeLayout L = sparse_large_layout(ld, nnz, D, H, C);
    float* sArt = pool + L.oArt;
    float* sRn1 = pool + L.oRn1;
    float* sRn2 = pool + L.oRn2;
    float* sYhat = pool + L.oYhat;
    float* sG3 = pool + L.oG3;
    float* sW1 = pool + L.oW;
    float* sW2 = sW1 + D * 33;
    float* sW3 = sW2 + H * 33;
    float* sWp = pool + L.oWp;
    const int sS = D | 1;
    float* stage = pool + L.oStage + wave * (TILE * sS);
    float* sAb = pool + L.oAb;
    unsigned short* scol = reinterpret_cast<unsigned short*>(pool + L.oCol);

    // ---------------- setup 2: sorted column lists from the plan's CSR ----------------
    {
        const unsigned short* gcol = csr_col + csr_off[2 * t + 1];
        for (int e = tid; e < nnz; e += NT) scol[e] = gcol[e];
    }
    __syncthreads();
    // ---------------- setup 3: uint16 temporaries in the Art .. G3 region ----------------
    unsigned short* u0 = reinterpret_cast<unsigned short*>(sArt);  // [ld] first upper entry (col > row) of the row
    unsigned short* upptr = u0 + ld;                                // [ld + 1] prefix of the upper counts
    unsigned short* level = upptr + ld + 1;                         // [ld] hop level 0..3
    unsigned short* slot_tab = level + ld;                          // per set: slot_start [ld + 1], order [ld], bucket [CHUNK + 1]
    for (int r = tid; r < ld; r += NT) {
        const int a = rowptr[r], b = rowptr[r + 1];
        const int f = lower_bound_u16(scol, a, b, r + 1);
        u0[r] = (unsigned short)f;
        upptr[r] = (unsigned short)(b - f);
        level[r] = (r == tr) ? 0 : 3;
    }
    __syncthreads();
    if (wave == 0) {
        const int total = wave_exclusive_scan_array(upptr, ld, lane);
        if (lane == 0) {
            upptr[ld] = (unsigned short)total;
            sh.eup = total;
        }
    }
    __syncthreads();
    for (int d = 1; d <= 2; ++d) {  // hop levels (see k_sparse_resident)
        for (int r = tid; r < n; r += NT)
            if (level[r] == d - 1)
                for (int e = rowptr[r]; e < rowptr[r + 1]; ++e)
                    if (level[scol[e]] > d) level[scol[e]] = (unsigned short)d;  // benign race
        __syncthreads();
    }
    constexpr int SETSZ_EXTRA = SPL_CHUNK + 2;
    if ((tid & 31) == 0 && (tid >> 5) < 2) {  // one thread per row set: A (level <= 2), B (level <= 1)
        const int set = tid >> 5;
        const int lvlmax = 2 - set;
        unsigned short* slot_start = slot_tab + set * (2 * ld + SETSZ_EXTRA);
        unsigned short* order = slot_start + ld + 1;
        unsigned short* bucket = order + ld;
        int pos = 0, pcount = 0, cnt = 0;
        for (int d = 0; d <= SPL_CHUNK; ++d) bucket[d] = 0;
        for (int rr = 0; rr < n; ++rr) {
            if (level[rr] > lvlmax) continue;
            ++cnt;
            const int d = rowptr[rr + 1] - rowptr[rr];
            if (d > SPL_CHUNK) {
                const int ns = sparse_slots_of_c(d, SPL_CHUNK);
                if (ns > SP_MAX_SPLIT) sh.bad = 1;
                pos = sparse_place(pos, ns);
                order[pcount] = (unsigned short)rr;
                slot_start[pcount] = (unsigned short)pos;
                pos += ns;
                ++pcount;
            } else {
                bucket[d]++;
            }
        }
        int run = pcount;
        for (int d = SPL_CHUNK; d >= 0; --d) {
            const int c = bucket[d];
            bucket[d] = (unsigned short)run;
            run += c;
        }
        for (int rr = 0; rr < n; ++rr) {
            if (level[rr] > lvlmax) continue;
            const int d = rowptr[rr + 1] - rowptr[rr];
            if (d <= SPL_CHUNK) order[bucket[d]++] = (unsigned short)rr;
        }
        for (int q = pcount; q < cnt; ++q) slot_start[q] = (unsigned short)(pos + (q - pcount));
        slot_start[cnt] = (unsigned short)(pos + (cnt - pcount));
        sh.set_rows[set] = cnt;
        sh.set_slots[set] = pos + (cnt - pcount);
        if (4 * (pos + (cnt - pcount)) + 128 > 32 * ld) sh.bad = 1;   // slot records: 4 words each in a [ld][32]-word array
    }
    __syncthreads();
    const int eup = sh.eup;
    // slot records -> workspace (set A first, then set B, each padded to whole waves): the phases walk them 512 at a time
    //   x = row | nsplit << 12 | wsplit << 17 | first << 22 | rem << 23 | inB << 28,   y = e0 | len << 16,   m0 / m1 = bit k: entry e0 + k
    //   points at t or a neighbour of t   (x = y = 0: empty slot; rem, inB, bmask: RowSlot)
    SlotRec* srec = reinterpret_cast<SlotRec*>(p.UT[1] + tm.offR * FS);
    const int slotsA = sh.bad ? 0 : sh.set_slots[0], slotsB = sh.bad ? 0 : sh.set_slots[1];
    const int padA = (slotsA + 31) & ~31, padB = (slotsB + 31) & ~31;
    for (int k = 0; k < 2; ++k) {
        const unsigned short* slot_start = slot_tab + k * (2 * ld + SETSZ_EXTRA);
        const unsigned short* order = slot_start + ld + 1;
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
                    if ((int)slot_start[mid] <= sl) lo = mid; else hi = mid;
                }
                const int row = order[lo];
                const int ra = rowptr[row], rb = rowptr[row + 1];
                const int ns = sparse_slots_of_c(rb - ra, SPL_CHUNK), kk = sl - (int)slot_start[lo];
                if (kk < ns) {
                    zrow = row;
                    ze0 = ra + kk * SPL_CHUNK;
                    zlen = ((ze0 + SPL_CHUNK < rb) ? ze0 + SPL_CHUNK : rb) - ze0;
                    zns = ns;
                    zrem = ns - kk;
                    zfirst = (kk == 0);
                    zinb = level[row] <= 1;
                    for (int k2 = 0; k2 < zlen; ++k2)
                        if (level[scol[ze0 + k2]] <= 1) (k2 < 32 ? zm0 : zm1) |= 1u << (k2 & 31);
                }
            }
            int wsplit = zfirst ? zns : 1;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int other = __shfl_xor(wsplit, o);
                wsplit = other > wsplit ? other : wsplit;
            }
            if (h == 0) {
                SlotRec rec;
                rec.x = (unsigned)zrow | ((unsigned)zns << 12) | ((unsigned)wsplit << 17) | ((unsigned)zfirst << 22) | ((unsigned)zrem << 23) |
                        ((unsigned)zinb << 28);
                rec.y = (unsigned)ze0 | ((unsigned)zlen << 16);
                rec.m0 = zm0;
                rec.m1 = zm1;
                srec[base + sl] = rec;
            }
        }
    }
    auto slot_of = [&](int set, int round) -> RowSlot {  // this lane's slot in the given round (512 slots per round)
        const int sl = round * (NT / 2) + wave * TILE + li;
        const int npad = set ? padB : padA;
        RowSlot z;
        SlotRec rec;
        rec.x = rec.y = rec.m0 = rec.m1 = 0u;
        if (sl < npad) rec = srec[(set ? padA : 0) + sl];
        z.row = rec.x & 4095u;
        z.nsplit = (rec.x >> 12) & 31u;
        z.wsplit = (rec.x >> 17) & 31u;
        z.first = (rec.x >> 22) & 1u;
        z.rem = (rec.x >> 23) & 31u;
        if (z.rem == 0) z.rem = 1;
        z.inB = (rec.x >> 28) & 1u;
        z.bmask = rec.m0;
        z.bmask_hi = rec.m1;
        z.e0 = rec.y & 0xffffu;
        z.e1 = z.e0 + (int)(rec.y >> 16);
        z.wave_active = round * (NT / 2) + wave * TILE < npad;
        if (z.nsplit == 0) z.nsplit = 1;
        if (z.wsplit == 0) z.wsplit = 1;
        return z;
    };
    const int roundsA = (padA + NT / 2 - 1) / (NT / 2), roundsB = (padB + NT / 2 - 1) / (NT / 2);
    float* gZraw = p.Zraw + tm.offR * FS;
    // per-edge indices -> global; Adam moments of the live entries start at zero; symmetry check
    {
        bool asym = (2 * eup != nnz);
        const int t0 = rowptr[tr], t1 = rowptr[tr + 1];
        for (int k = tid; k < eup; k += NT) {
            int lo = 0, hi = ld;  // largest row i with upptr[i] <= k
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((int)upptr[mid] <= k) lo = mid; else hi = mid;
            }
            const int i = lo;
            const int e = (int)u0[i] + (k - (int)upptr[i]);
            const int j = scol[e];
            const int em = lower_bound_u16(scol, rowptr[j], rowptr[j + 1], i);
            if (em >= rowptr[j + 1] || (int)scol[em] != i) asym = true;
            if (Ag[(size_t)j * ld + i] != Ag[(size_t)i * ld + j]) asym = true;
            const int pi = lower_bound_u16(scol, t0, t1, i), pj = lower_bound_u16(scol, t0, t1, j);
            const bool near = i == tr || j == tr || (pi < t1 && (int)scol[pi] == i) || (pj < t1 && (int)scol[pj] == j);
            const bool near2 = level[i] <= 2 || level[j] <= 2;  // else dZ1 is exactly zero on both endpoints: regularisers only
            eidx[2 * k] = (unsigned)i | ((unsigned)j << 12) | ((unsigned)near << 24) | ((unsigned)near2 << 25);
            eidx[2 * k + 1] = (unsigned)e | ((unsigned)(asym ? e : em) << 16);
            est[0 * eup + k] = Mg[(size_t)i * ld + j];
            est[1 * eup + k] = Mg[(size_t)j * ld + i];
            est[2 * eup + k] = 0.0f;
            est[3 * eup + k] = 0.0f;
            est[4 * eup + k] = 0.0f;
            est[5 * eup + k] = 0.0f;
            est[6 * eup + k] = Ag[(size_t)i * ld + j];
        }
        if (asym) sh.bad = 1;
    }
    __syncthreads();  // the uint16 temporaries are dead from here on
    if (sh.bad) {
        fail_nan();
        return;
    }
    // ---------------- row arrays (never-written rows must read as zero), model, labels ----------------
    for (int e = tid; e < n * FS; e += NT) {
        gU1[e] = 0.0f;
        gU2[e] = 0.0f;
        gdZ1[e] = 0.0f;
    }
    for (int e = tid; e < nnz; e += NT) gGe[e] = 0.0f;   // entries of rows beyond two hops are never written
    for (int e = tid; e < D * 32; e += NT) sW1[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + e];
    for (int e = tid; e < H * 32; e += NT) sW2[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 1024 + e];
    for (int e = tid; e < H * 32; e += NT) sW3[(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + 2048 + e];
    if (tid < 96) sh.bias[tid >> 5][tid & 31] = p.wts[WT_B + tid];
    for (int e = tid; e < C * 96; e += NT) sWp[e] = p.wts[WT_WP + e];
    if (tid < CMAX) sh.sbp[tid] = p.wts[WT_BP + tid];
    for (int r = tid; r < ld; r += NT) {
        sYhat[r] = p.yhat[tm.offR + r];
        sArt[r] = 0.0f;
    }
    if (tid < 32) {
        sh.fcur[tid] = 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
        sh.mf[tid] = 0.0f;
        sh.vf[tid] = 0.0f;
    }
    const float inv_n2 = 1.0f / ((float)n * (float)n);
    const int rt0 = rowptr[tr], rt1 = rowptr[tr + 1];
    // sigma(M) -> symmetrised masked adjacency, one float per directed entry
    for (int k = tid; k < eup; k += NT) {
        const unsigned en = eidx[2 * k + 1];
        const float a = est[6 * eup + k] * (0.5f * (sigmoidf_(est[0 * eup + k]) + sigmoidf_(est[1 * eup + k])));
        sAb[en & 0xffffu] = a;
        sAb[en >> 16] = a;
    }
    __syncthreads();

    for (int iter = 0; iter < p.num_iters; ++iter) {
        if (tid < 32) sh.phi[tid] = (tid < D) ? sigmoidf_(sh.fcur[tid]) : 0.0f;
        for (int e = rt0 + tid; e < rt1; e += NT) sArt[scol[e]] = sAb[e];  // Abar[t][.] as a dense row (zero elsewhere)
        __syncthreads();
        const float step_size = adam_tab[2 * iter], inv_bc2s = 1.0f / adam_tab[2 * iter + 1];

        // ======== layer 1 on the rows within two hops: Zraw = Abar . X (kept in registers), U1 ========
        for (int round = 0; round < roundsA; ++round) {
            const RowSlot SA = slot_of(0, round);
            if (!SA.wave_active) continue;  // uniform per wave
            const bool first = SA.first;
            const int r = first ? SA.row : 0;
            float acc[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) acc[q] = 0.0f;
            sparse_gather<false, DQ, SPL_GATHER_UNROLL>(sAb, scol, gX, FS, D, SA.e0, SA.e1, h, acc);
            sparse_combine<DQ>(acc, SA.rem, SA.wsplit);
#pragma unroll
            for (int q = 0; q < DQ; ++q) {
                if (first && 2 * q + h < D) gZraw[r * FS + 2 * q + h] = acc[q];  // for the feature-mask gradient
                acc[q] = (first && 2 * q + h < D) ? acc[q] * sh.phi[2 * q + h] : 0.0f;
            }
            sparse_forward_rowlocal<DQ>(acc, sW1, sh.bias[0], D, H, li, h, first, gU1 + r * FS, sRn1 + r);
        }
        __syncthreads();
        // ======== layer 2 on the target and its neighbours: U2 ========
        for (int round = 0; round < roundsB; ++round) {
            const RowSlot SB = slot_of(1, round);
            if (!SB.wave_active) continue;
            const bool first = SB.first;
            const int r = first ? SB.row : 0;
            float acc[HQ];
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
            sparse_gather<true, HQ, SPL_GATHER_UNROLL>(sAb, scol, gU1, FS, H, SB.e0, SB.e1, h, acc);
            sparse_combine<HQ>(acc, SB.rem, SB.wsplit);
#pragma unroll
            for (int q = 0; q < HQ; ++q) acc[q] = first ? acc[q] : 0.0f;
            sparse_forward_rowlocal<HQ>(acc, sW2, sh.bias[1], H, H, li, h, first, gU2 + r * FS, sRn2 + r);
        }
        __syncthreads();
        // ======== row t of layer 3, head, dE, dZ3[t] ========
        {
            float z = 0.0f;
            if (li < H)
                for (int e = rt0 + 2 * wave + h; e < rt1; e += 2 * NW) z = fmaf(sAb[e], relu_(gU2[(int)scol[e] * FS + li]), z);
            z += __shfl_xor(z, 32);
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
            const float art = sArt[r];
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
            gpart += __shfl_xor(gpart, 32);
            if (first && h == 0) sG3[r] = gpart;
            const f32x16 c16 = sparse_backward_rowlocal<HQ>(du, uu, first ? sRn2[r] : 1.0f, sW2, H, H, li, h);
            sparse_store_cols(c16, gU2 + r * FS, H, first, h);
        }
        __syncthreads();
        const float* gdZ2 = gU2;
        // ======== dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 on the rows within two hops; feature-mask gradient partials ========
        {
            float dfq[DQ];
#pragma unroll
            for (int q = 0; q < DQ; ++q) dfq[q] = 0.0f;
            for (int round = 0; round < roundsA; ++round) {
                const RowSlot SA = slot_of(0, round);
                if (!SA.wave_active) continue;
                const bool first = SA.first;
                const int r = first ? SA.row : 0;
                float acc[HQ], uu[HQ];
#pragma unroll
                for (int q = 0; q < HQ; ++q) acc[q] = 0.0f;
                // dZ2 is non-zero only on t and its neighbours: only this slot's entries that point there contribute (marked in the
                // slot record), not the whole 64-entry chunk
                for (int half = 0; half < 2; ++half)
                    for (unsigned m = half ? SA.bmask_hi : SA.bmask; m; m &= m - 1u) {
                        const int e = SA.e0 + 32 * half + __ffs((int)m) - 1;
                        const float a = sAb[e];
                        const float* br = gdZ2 + (int)scol[e] * FS + h;
#pragma unroll
                        for (int q = 0; q < HQ; ++q) acc[q] = fmaf(a, (2 * q + h < H) ? br[2 * q] : 0.0f, acc[q]);
                    }
                sparse_combine<HQ>(acc, SA.rem, SA.wsplit);
#pragma unroll
                for (int q = 0; q < HQ; ++q) {
                    const int c = 2 * q + h;
                    const float u = (first && c < H) ? gU1[r * FS + c] : 0.0f;
                    float dx = acc[q];
                    if (first && r == tr && c < H) dx += sh.dEs[c];
                    acc[q] = (u > 0.0f) ? dx : 0.0f;
                    uu[q] = u;
                }
                const f32x16 c16 = sparse_backward_rowlocal<HQ>(acc, uu, first ? sRn1[r] : 1.0f, sW1, D, H, li, h);
                sparse_store_cols(c16, gdZ1 + r * FS, D, first, h);
                sparse_store_cols(c16, stage + li * sS, D, true, h);  // same values through LDS for the other half-lane
                wave_sync();
#pragma unroll
                for (int q = 0; q < DQ; ++q)
                    if (first && 2 * q + h < D) dfq[q] = fmaf(stage[li * sS + 2 * q + h], gZraw[r * FS + 2 * q + h], dfq[q]);
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
                    for (int e = SA.e0 + h; e < SA.e1; e += 4) {
                        const bool two = e + 2 < SA.e1;
                        const int j0 = scol[e], j1 = scol[two ? e + 2 : e];
                        const float* x0 = gX + j0 * FS;
                        const float* x1 = gX + j1 * FS;
                        float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
#pragma unroll
                        for (int c = 0; c < 2 * DQ; c += 2) {
                            a0 = fmaf(dz[c], x0[c], a0);
                            a1 = fmaf(dz[c + 1], x0[c + 1], a1);
                            b0 = fmaf(dz[c], x1[c], b0);
                            b1 = fmaf(dz[c + 1], x1[c + 1], b1);
                        }
                        if (inB) {
                            const float* u0 = gU1 + j0 * FS;
                            const float* u1 = gU1 + j1 * FS;
#pragma unroll
                            for (int c = 0; c < 2 * HQ; c += 2) {
                                a0 = fmaf(d2[c], relu_(u0[c]), a0);
                                a1 = fmaf(d2[c + 1], relu_(u0[c + 1]), a1);
                                b0 = fmaf(d2[c], relu_(u1[c]), b0);
                                b1 = fmaf(d2[c + 1], relu_(u1[c + 1]), b1);
                            }
                        }
                        gGe[e] = a0 + a1;
                        if (two) gGe[e + 2] = b0 + b1;
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
                v += __shfl_xor(v, 16);
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
        // ======== per edge: G_ij + G_ji, regulariser gradients, Adam in place on both directed entries, next Abar ========
        const bool republish = iter + 1 < p.num_iters;  // the returned mask is the one of the LAST forward (explain.py:209-211)
        for (int k = tid; k < eup; k += NT) {
            const unsigned nd = eidx[2 * k], en = eidx[2 * k + 1];
            const int i = nd & 4095u, j = (nd >> 12) & 4095u;
            const bool near = (nd >> 24) & 1u, near2 = (nd >> 25) & 1u;
            (void)near;
            (void)near2;
            const float G0 = gGe[en & 0xffffu], G1 = gGe[en >> 16];   // row-side products of both directions (layer-1 backward)
            float G = G0 + G1;
            G += (i == tr) ? sG3[j] : 0.0f;
            G += (j == tr) ? sG3[i] : 0.0f;
            const float dy = sYhat[i] - sYhat[j];
            const float w = est[6 * eup + k];
            const float gc = (0.5f * G + p.c_lap * 0.5f * dy * dy * inv_n2) * w;
            float Mij = est[0 * eup + k], Mji = est[1 * eup + k], mij = est[2 * eup + k], mji = est[3 * eup + k],
                  vij = est[4 * eup + k], vji = est[5 * eup + k];
            {
                const float S = sigmoidf_(Mij);
                const float g = (gc + p.c_size - p.c_ent * Mij * inv_n2) * S * (1.0f - S);
                adam_update(Mij, mij, vij, g, p.beta1, p.beta2, p.eps, step_size, inv_bc2s);
            }
            {
                const float S = sigmoidf_(Mji);
                const float g = (gc + p.c_size - p.c_ent * Mji * inv_n2) * S * (1.0f - S);
                adam_update(Mji, mji, vji, g, p.beta1, p.beta2, p.eps, step_size, inv_bc2s);
            }
            est[0 * eup + k] = Mij;
            est[1 * eup + k] = Mji;
            est[2 * eup + k] = mij;
            est[3 * eup + k] = mji;
            est[4 * eup + k] = vij;
            est[5 * eup + k] = vji;
            if (republish) {  // nobody reads sAb any more in this iteration (the barrier above); sArt is rebuilt next iteration
                const float a = w * (0.5f * (sigmoidf_(Mij) + sigmoidf_(Mji)));
                sAb[en & 0xffffu] = a;
                sAb[en >> 16] = a;
            }
        }
        for (int r = tid; r < ld; r += NT) sArt[r] = 0.0f;
        __syncthreads();
        if (tid < D) {  // feature mask
            const float ph = sh.phi[tid];
            const float gf = (sh.dfp[tid] + p.c_feat_size / (float)D) * ph * (1.0f - ph);
            float fn = sh.fcur[tid], m = sh.mf[tid], v = sh.vf[tid];
            adam_update(fn, m, v, gf, p.beta1, p.beta2, p.eps, step_size, inv_bc2s);
            sh.fcur[tid] = fn;
            sh.mf[tid] = m;
            sh.vf[tid] = v;
        }
        __syncthreads();
    }
    // ---------------- results: dense Abar block (zero off the edges), M on the edges, feature mask ----------------
    {
        f32x4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int e = tid * 4; e < ld * ld; e += 4 * NT) *reinterpret_cast<f32x4*>(p.Abar + tm.offQ + e) = z4;
    }
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < eup; k += NT) {
        const unsigned nd = eidx[2 * k], en = eidx[2 * k + 1];
        const int i = nd & 4095u, j = (nd >> 12) & 4095u;
        const float a = sAb[en & 0xffffu];
        p.Abar[tm.offQ + (size_t)i * ld + j] = a;
        p.Abar[tm.offQ + (size_t)j * ld + i] = a;
        Mg[(size_t)i * ld + j] = est[0 * eup + k];
        Mg[(size_t)j * ld + i] = est[1 * eup + k];
    }
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < D) ? sh.fcur[tid] : 0.0f;
}

// gnnx_plan_analyze, every target with n <= SPL_N_MAX (also those of <= 512 rows that fit no resident class, e.g. more
// than 2048 edges): directed entries and the row slots (of SPL_CHUNK entries)
// needed by the rows within two hops of the target - the same levels and placement as k_sparse_large computes - and the
// same count for slots of SP_CHUNK entries (the 512-thread class of k_sparse_resident).
// out[3 t] = nnz, out[3 t + 1] = slots of 64 (-1: a row that cannot be placed), out[3 t + 2] = slots of 16 (-1 likewise);
// targets outside the range get -1 everywhere.
__global__ __launch_bounds__(256) void k_count_edges_large(const TargetMeta* meta, const float* A, int32_t* out) {
    __shared__ int deg[SPL_N_MAX + 1];
    __shared__ unsigned char level[SPL_N_MAX + 1];
    __shared__ int part[4];
    const TargetMeta tm = meta[blockIdx.x];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tm.n > SPL_N_MAX) {
        if (tid == 0) {
            out[3 * blockIdx.x] = -1;
            out[3 * blockIdx.x + 1] = -1;
            out[3 * blockIdx.x + 2] = -1;
        }
        return;
    }
    const float* Ag = A + tm.offQ;
    int cnt = 0;
    for (int r = wave; r < tm.n; r += 4) {
        int d = 0;
        for (int c0 = 0; c0 < tm.n; c0 += 64) {
            const int c = c0 + lane;
            const bool nz = (c < tm.n && c != r) ? (Ag[(size_t)r * tm.ld + c] != 0.0f) : false;
            d += __popcll(__ballot(nz));
        }
        cnt += d;
        if (lane == 0) deg[r] = d;
    }
    for (int r = tid; r < tm.n; r += 256) level[r] = (r == tm.t) ? 0 : 3;
    if (lane == 0) part[wave] = cnt;
    __syncthreads();
    for (int d = 1; d <= 2; ++d) {  // hop levels from the dense rows
        for (int r = wave; r < tm.n; r += 4) {
            if (level[r] != d - 1) continue;  // uniform per wave
            for (int c0 = 0; c0 < tm.n; c0 += 64) {
                const int c = c0 + lane;
                if (c < tm.n && c != r && Ag[(size_t)r * tm.ld + c] != 0.0f && level[c] > d) level[c] = (unsigned char)d;
            }
        }
        __syncthreads();
    }
    if (tid < 2) {  // thread 0: slots of SPL_CHUNK entries, thread 1: slots of SP_CHUNK entries
        const int chunk = tid ? SP_CHUNK : SPL_CHUNK;
        int pos = 0, singles = 0;
        bool placeable = true;
        for (int r = 0; r < tm.n; ++r) {
            if (level[r] > 2) continue;
            if (deg[r] > chunk) {
                const int ns = sparse_slots_of_c(deg[r], chunk);
                placeable &= ns <= SP_MAX_SPLIT;
                pos = sparse_place(pos, ns) + ns;
            } else {
                ++singles;
            }
        }
        if (tid == 0) out[3 * blockIdx.x] = part[0] + part[1] + part[2] + part[3];
        out[3 * blockIdx.x + 1 + tid] = placeable ? pos + singles : -1;
    }
}

// gnnx_plan_analyze: CSR (rowptr [ld + 1], ascending columns) of every target routed to k_sparse_large, from its block
// of the packed dense adjacency; one workgroup per target, rows by waves, 4 chunks of 64 columns in flight per wave.
__global__ __launch_bounds__(512) void k_build_csr_large(const TargetMeta* meta, const float* A, const int32_t* targets,
                                                         const long long* csr_off, int32_t* csr_rowptr, unsigned short* csr_col) {
    const int t = targets[blockIdx.x];
    const TargetMeta tm = meta[t];
    const int n = tm.n, ld = tm.ld;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = 8;
    const float* Ag = A + tm.offQ;
    int32_t* rowptr = csr_rowptr + csr_off[2 * t];
    unsigned short* col = csr_col + csr_off[2 * t + 1];
    __shared__ int srp[SPL_N_MAX + 34];
    for (int r = wave; r < ld; r += NW) {
        int cnt = 0;
        if (r < n)
            for (int c0 = 0; c0 < n; c0 += 256) {
                float a[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int c = c0 + 64 * u + lane;
                    a[u] = (c < n && c != r) ? Ag[(size_t)r * ld + c] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) cnt += __popcll(__ballot(a[u] != 0.0f));
            }
        if (lane == 0) srp[r] = cnt;
    }
    __syncthreads();
    if (wave == 0) {
        const int total = wave_exclusive_scan_array(srp, ld, lane);
        if (lane == 0) srp[ld] = total;
    }
    __syncthreads();
    for (int r = tid; r <= ld; r += 512) rowptr[r] = srp[r];
    for (int r = wave; r < n; r += NW) {
        int base = srp[r];
        for (int c0 = 0; c0 < n; c0 += 256) {
            float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 64 * u + lane;
                a[u] = (c < n && c != r) ? Ag[(size_t)r * ld + c] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool nz = a[u] != 0.0f;
                const unsigned long long bal = __ballot(nz);
                if (nz) col[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(c0 + 64 * u + lane);
                base += __popcll(bal);
            }
        }
    }
}

}  // namespace gnnx
