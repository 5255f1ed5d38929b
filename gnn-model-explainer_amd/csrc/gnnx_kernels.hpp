// gnnx_kernels.hpp — CDNA4 (gfx950) device code of the GNNExplainer mask-optimisation loop.
//
// One iteration of the reference loop (explainer/explain.py:137-146) for ALL targets of a batch is
//   k_conv<FWD1..3>   masked-adjacency contraction  Z = Abar . X_{l-1}  on MFMA (exact f32,
//                     v_mfma_f32_32x32x2_f32), split-K over the 4 waves of a workgroup, LDS reduction,
//                     fused epilogue  (.W + b, row L2-normalise)            models.py:58-80, 230-267
//   k_head            logits of the target row / max-pooled rows, softmax, -log p, dE    models.py:375,
//                     269-316; explain.py:709-714, 750-753
//   k_conv<BWD3>      row-local backward through the last layer's normalise -> dZ3
//   k_conv<BWD2,1>    dX = Abar . dZ_{l+1} (same MFMA contraction, Abar symmetric) + fused row-local
//                     backward (ReLU mask, normalise Jacobian, .W^T) -> dZ_l
//   k_conv<BWD0>      dX0 = Abar . dZ_1, reduced to the feature-mask gradient
//   k_mask<true>      fused, memory-bound: G = dL/dAbar tile on MFMA (K = D+2H, never materialised),
//                     + Laplacian, size and entropy regulariser gradients, sigmoid', Adam on (M, m, v),
//                     new sigma(M) symmetrised into the next Abar, loss partial sums wave-reduced.
//                     explain.py:665-678, 755-770, 780-793; utils/train_utils.py:9-10
// Math: SURVEY.md Appendix A; CPU spec: oracle/closed_form.py (tests only).
//
// Layout: see include/gnnx.h.  Every leading dimension is a multiple of 32 so a 32x32 MFMA tile never
// straddles a target; the symmetric Abar is read as Abar[k][i] (128-B coalesced segments) for the A
// operand, feature rows are the B operand.
#pragma once
#include <stdint.h>

namespace gnnx {

constexpr int TILE = 32;  // MFMA 32x32x2 tile edge
constexpr int FS = 32;    // floats per feature row
constexpr int CMAX = 32;  // max classes
constexpr int NLOSS = 8;

// offsets (floats) inside the packed, zero-padded model block
constexpr int WT_W = 0;                  // W_l   [32][32]  at WT_W + l*1024   (row = input k, col = output c)
constexpr int WT_B = 3 * 1024;           // b_l   [32]      at WT_B + l*32
constexpr int WT_WP = WT_B + 3 * 32;     // Wp    [32][96]  (class c, l*32 + j)
constexpr int WT_BP = WT_WP + CMAX * 96; // bp    [32]
constexpr int WT_TOTAL = WT_BP + CMAX;

struct TargetMeta {
    int32_t n;     // sub-graph nodes
    int32_t ld;    // round_up(n, 32)
    int32_t t;     // target row (node mode)
    int32_t y_gt;  // ground-truth label
    int64_t offQ;  // float offset of the ld x ld block in square arrays
    int64_t offR;  // row offset in row arrays
};

struct ConvTile { int32_t t, rb; };       // target, 32-row block
struct MaskTile { int32_t t, I, J; int32_t pad; };  // target, tile pair I <= J

struct Params {
    const TargetMeta* meta;
    const float* A;   // adjacency (symmetric, zero padded)
    float* M;         // edge-mask parameter
    float* mM;        // Adam first moment
    float* vM;        // Adam second moment
    float* Abar;      // masked adjacency of the current iterate
    const float* X;   // input features [R][32]
    const float* XT;  // per target column-major copy [32][ld]
    const float* yhat;  // predicted class ids as float [R]
    float* U[3];      // normalised pre-activations, row-major [R][32]
    float* UT[3];     // same, per target column-major [32][ld]
    float* rn[3];     // row norms [R]
    float* dZ[3];     // gradients w.r.t. the aggregated inputs, row-major [R][32]
    float* dZT[3];    // same, column-major
    float* dE;        // direct gradient of the concatenated embedding [T][3][32]
    int32_t* argrow;  // row that receives dE[l][c]  [T][3][32]
    float* df;        // feature-mask gradient partials, one row of 32 per 32-row block [R/32][32]
    float* f[2];      // feature-mask parameter, ping-pong by iteration parity [T][32]
    float* mf;
    float* vf;
    float* probs;     // softmax of the head [T][CMAX]
    float* loss;      // [T][num_iters][NLOSS] or null
    const float* wts; // packed model block
    int32_t D, H, O, C;
    int32_t graph_mode;
    int32_t num_iters;
    float lr, beta1, beta2, eps;
    float c_size, c_feat_size, c_ent, c_lap;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// row of the 32x32 MFMA accumulator held in register r of a lane in half h (lane>>5); column = lane&31
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

enum ConvMode { FWD1 = 0, FWD2 = 1, FWD3 = 2, BWD3 = 3, BWD2 = 4, BWD1 = 5, BWD0 = 6 };

// ---------------------------------------------------------------------------------------------
// Masked-adjacency contraction + fused row-local epilogue.  One workgroup (4 waves) per 32-row
// block of one target; the K range (all ld columns of Abar) is split over the 4 waves.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_conv(Params p, const ConvTile* tiles, int iter) {
    __shared__ float red[4 * TILE * 33];  // split-K partial tiles, then reused as Z / dY staging
    __shared__ float wl[32 * 33];         // layer weight, padded rows
    __shared__ float zs[TILE * 33];

    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = p.meta[tl.t];
    const int ld = tm.ld;
    const int row0 = tl.rb * TILE;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;

    constexpr int layer = (MODE == FWD1 || MODE == BWD1) ? 0 : (MODE == FWD2 || MODE == BWD2) ? 1 : 2;
    // stage the layer weight (BWD0 has no row-local part)
    if (MODE != BWD0) {
        const float* W = p.wts + WT_W + layer * 1024;
        for (int e = tid; e < 1024; e += 256) wl[(e >> 5) * 33 + (e & 31)] = W[e];
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    if (MODE != BWD3) {
        const float* Ab = p.Abar + tm.offQ;
        const float* Bsrc = (MODE == FWD1)   ? p.X
                            : (MODE == FWD2) ? p.U[0]
                            : (MODE == FWD3) ? p.U[1]
                            : (MODE == BWD2) ? p.dZ[2]
                            : (MODE == BWD1) ? p.dZ[1]
                                             : p.dZ[0];
        Bsrc += tm.offR * FS;
        float phi = 1.0f;
        if (MODE == FWD1) phi = (li < p.D) ? sigmoidf_(p.f[iter & 1][tl.t * FS + li]) : 0.0f;
        const int kchunk = ld >> 2;
        const int k0 = wave * kchunk;

        for (int s = 0; s < kchunk; s += 2) {
            const int k = k0 + s + h;
            float a = Ab[(size_t)k * ld + row0 + li];  // Abar[k][i] == Abar[i][k]
            float b = Bsrc[(size_t)k * FS + li];
            if (MODE == FWD1) b *= phi;
            if (MODE == FWD2 || MODE == FWD3) b = fmaxf(b, 0.0f);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // split-K reduction through LDS
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
    __syncthreads();
    const int row = tid >> 3;        // 0..31
    const int cg = (tid & 7) * 4;    // column group
    float z4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += red[(w * TILE + row) * 33 + cg + j];
        z4[j] = s;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) zs[row * 33 + cg + j] = z4[j];
    __syncthreads();

    const size_t grow = (size_t)tm.offR + row0 + row;  // global row in row arrays
    const int irow = row0 + row;                       // row inside the target

    if (MODE == FWD1 || MODE == FWD2 || MODE == FWD3) {
        const int din = (MODE == FWD1) ? p.D : p.H;
        const int dout = (MODE == FWD3) ? p.O : p.H;
        const float* bias = p.wts + WT_B + layer * 32;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = 0.0f;
        for (int k = 0; k < din; ++k) {
            const float z = zs[row * 33 + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf(z, wl[k * 33 + cg + j], y[j]);
        }
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = (cg + j < dout) ? y[j] + bias[cg + j] : 0.0f;
            ss = fmaf(y[j], y[j], ss);
        }
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 4);
        const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
        float* U = p.U[layer] + grow * FS;
        float* UT = p.UT[layer] + tm.offR * FS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u = y[j] / rnorm;
            U[cg + j] = u;
            UT[(size_t)(cg + j) * ld + irow] = u;
        }
        if ((tid & 7) == 0) p.rn[layer][grow] = rnorm;
    } else if (MODE == BWD0) {
        // dX0 rows -> feature-mask gradient: df[d] += sum_rows dX0[row][d] * X[row][d]
        const float* Xr = p.X + grow * FS;
        float part[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) part[j] = zs[row * 33 + cg + j] * Xr[cg + j];
        // reduce over the 32 rows: rows differ in tid>>3 -> lanes 8 apart inside a wave, waves via LDS
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[j] += __shfl_xor(part[j], 8);
            part[j] += __shfl_xor(part[j], 16);
            part[j] += __shfl_xor(part[j], 32);
        }
        __syncthreads();
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wave * 32 + cg + j] = part[j];
        }
        __syncthreads();
        if (tid < 32) {  // one partial per row block: summed in a fixed order by k_mask (deterministic)
            const float s = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
            p.df[((size_t)(tm.offR >> 5) + tl.rb) * FS + tid] = s;
        }
    } else {
        // BWD3 / BWD2 / BWD1: dX (+ direct part) -> dZ_layer
        const int dout = (layer == 2) ? p.O : p.H;  // width of U[layer] / dX
        const int din = (layer == 0) ? p.D : p.H;   // width of dZ[layer]
        const float* U = p.U[layer] + grow * FS;
        const float* dE = p.dE + (tl.t * 3 + layer) * FS;
        const int32_t* ar = p.argrow + (tl.t * 3 + layer) * FS;
        float du[4], u[4];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cg + j;
            float dx = (MODE == BWD3) ? 0.0f : zs[row * 33 + c];
            u[j] = U[c];
            if (c < dout && ar[c] == irow) dx += dE[c];
            if (layer < 2) dx = (u[j] > 0.0f) ? dx : 0.0f;
            du[j] = (c < dout) ? dx : 0.0f;
            s = fmaf(du[j], u[j], s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        const float rnorm = p.rn[layer][grow];
        __syncthreads();  // everyone is done reading zs
#pragma unroll
        for (int j = 0; j < 4; ++j) zs[row * 33 + cg + j] = (du[j] - u[j] * s) / rnorm;  // dY
        __syncthreads();
        float dz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dz[j] = 0.0f;
        for (int c = 0; c < dout; ++c) {
            const float dy = zs[row * 33 + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) dz[j] = fmaf(dy, wl[(cg + j) * 33 + c], dz[j]);
        }
        float* dZ = p.dZ[layer] + grow * FS;
        float* dZT = p.dZT[layer] + tm.offR * FS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = (cg + j < din) ? dz[j] : 0.0f;
            dZ[cg + j] = v;
            dZT[(size_t)(cg + j) * ld + irow] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Head: one workgroup (256 threads) per target.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(Params p, int iter) {
    __shared__ float e[96];
    __shared__ int erow[96];
    __shared__ float g[CMAX];
    __shared__ float wmax[4 * 96];
    __shared__ int warg[4 * 96];
    const int t = blockIdx.x;
    const TargetMeta tm = p.meta[t];
    const int tid = threadIdx.x;
    const int dims[3] = {p.H, p.H, p.O};

    if (!p.graph_mode) {
        if (tid < 96) {
            const int l = tid >> 5, c = tid & 31;
            float v = 0.0f;
            if (c < dims[l]) {
                v = p.U[l][(tm.offR + tm.t) * FS + c];
                if (l < 2) v = fmaxf(v, 0.0f);
            }
            e[tid] = v;
            erow[tid] = tm.t;
        }
    } else {
        // column-wise max over the n rows of every layer (padded rows of the reference included:
        // they are rows < n here; the engine's own padding rows n..ld-1 are excluded)
        const int wave = tid >> 6, lane = tid & 63;
        for (int col = wave; col < 96; col += 4) {
            const int l = col >> 5, c = col & 31;
            float best = -3.0e38f;
            int barg = 0;
            if (c < dims[l]) {
                const float* UT = p.UT[l] + tm.offR * FS + (size_t)c * tm.ld;
                for (int i = lane; i < tm.n; i += 64) {
                    float v = UT[i];
                    if (l < 2) v = fmaxf(v, 0.0f);
                    if (v > best) { best = v; barg = i; }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ob = __shfl_xor(best, o);
                    const int oa = __shfl_xor(barg, o);
                    if (ob > best || (ob == best && oa < barg)) { best = ob; barg = oa; }
                }
            } else {
                best = 0.0f;
            }
            if (lane == 0) { e[col] = best; erow[col] = barg; }
        }
    }
    __syncthreads();
    const float* Wp = p.wts + WT_WP;
    if (tid < 64) {  // wave 0: logits, softmax
        float z = -3.0e38f;
        if (tid < p.C) {
            float s = 0.0f;
            for (int q = 0; q < 96; ++q) s = fmaf(Wp[tid * 96 + q], e[q], s);
            z = s + p.wts[WT_BP + tid];
        }
        float mx = z;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float ex = (tid < p.C) ? expf(z - mx) : 0.0f;
        float sum = ex;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const float pr = ex / sum;
        if (tid < CMAX) {
            g[tid] = (tid < p.C) ? pr - ((tid == tm.y_gt) ? 1.0f : 0.0f) : 0.0f;
            p.probs[t * CMAX + tid] = (tid < p.C) ? pr : 0.0f;
        }
        if (p.loss && tid == tm.y_gt) p.loss[((size_t)t * p.num_iters + iter) * NLOSS + 0] = -logf(pr);
    }
    __syncthreads();
    if (tid < 96) {
        float s = 0.0f;
        for (int c = 0; c < p.C; ++c) s = fmaf(Wp[c * 96 + tid], g[c], s);
        p.dE[t * 96 + tid] = s;
        p.argrow[t * 96 + tid] = erow[tid];
    }
    if (p.loss && tid == 128) {
        float s = 0.0f;
        for (int d = 0; d < p.D; ++d) s += sigmoidf_(p.f[iter & 1][t * FS + d]);
        p.loss[((size_t)t * p.num_iters + iter) * NLOSS + 4] = p.c_feat_size * s / (float)p.D;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused mask kernel: one wave per tile pair {(I,J),(J,I)}, I <= J.
//   UPDATE=false : only Abar = A * sym(sigma(M)) (initial forward)
//   UPDATE=true  : gradient + Adam step on both tiles, then (WRITE_ABAR) the next Abar
// ---------------------------------------------------------------------------------------------
template <bool UPDATE, bool WRITE_ABAR>
__global__ __launch_bounds__(64) void k_mask(Params p, const MaskTile* tiles, int iter, float step_size, float bc2s) {
    __shared__ float tM[TILE * 33], tm1[TILE * 33], tv[TILE * 33], tA[TILE * 33];
    const MaskTile tl = tiles[blockIdx.x];
    const TargetMeta tm = p.meta[tl.t];
    const int ld = tm.ld, n = tm.n;
    const int I0 = tl.I * TILE, J0 = tl.J * TILE;
    const bool diag = (tl.I == tl.J);
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    const size_t q = tm.offQ;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    if (UPDATE) {
        const size_t ro = (size_t)tm.offR * FS;
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            const int d = (l == 0) ? p.D : p.H;
            const float* zT = p.dZT[l] + ro;
            const float* xT = (l == 0) ? p.XT + ro : p.UT[l - 1] + ro;
            for (int s = 0; s < d; s += 2) {
                const int k = s + h;  // k < 32 always; columns >= d hold zeros
                float phi = 1.0f;
                if (l == 0) phi = (k < p.D) ? sigmoidf_(p.f[iter & 1][tl.t * FS + k]) : 0.0f;
                const float zi = zT[(size_t)k * ld + I0 + li];
                const float zj = zT[(size_t)k * ld + J0 + li];
                float xi = xT[(size_t)k * ld + I0 + li];
                float xj = xT[(size_t)k * ld + J0 + li];
                if (l == 0) { xi *= phi; xj *= phi; } else { xi = fmaxf(xi, 0.0f); xj = fmaxf(xj, 0.0f); }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zi, xj, acc, 0, 0, 0);  // G[i][j]
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, zj, acc, 0, 0, 0);  // G[j][i]
            }
        }
    }

    // stage the partner tile (J,I) row-wise (coalesced) so it can be read transposed
    for (int rr = h; rr < TILE; rr += 2) {
        const size_t g = q + (size_t)(J0 + rr) * ld + I0 + li;
        tM[rr * 33 + li] = p.M[g];
        if (UPDATE) { tm1[rr * 33 + li] = p.mM[g]; tv[rr * 33 + li] = p.vM[g]; }
    }
    __syncthreads();

    const float inv_n2 = 1.0f / ((float)n * (float)n);
    float yj = 0.0f;
    const bool lapl = UPDATE && !p.graph_mode;
    if (lapl) yj = p.yhat[tm.offR + J0 + li];
    // step_size = lr / (1 - beta1^k), bc2s = sqrt(1 - beta2^k): evaluated in double on the host, as torch does
    float s_size = 0.0f, s_ent = 0.0f, s_lap = 0.0f;

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = acc_row(r, h), j = li;
        const int gi = I0 + i, gj = J0 + j;
        const bool valid = (gi < n) && (gj < n);
        const size_t own = q + (size_t)gi * ld + gj;
        const float Aij = p.A[own];
        const float offd = (gi != gj) ? 1.0f : 0.0f;
        float Mij = p.M[own];
        float Mji = tM[j * 33 + i];
        if (UPDATE) {
            float mij = p.mM[own], vij = p.vM[own];
            float mji = tm1[j * 33 + i], vji = tv[j * 33 + i];
            const float Sij = sigmoidf_(Mij), Sji = sigmoidf_(Mji);
            float Gs = 0.5f * acc[r];
            if (lapl) {
                const float yi = p.yhat[tm.offR + gi];
                const float dy = yi - yj;
                Gs += p.c_lap * 0.5f * dy * dy * inv_n2;
                if (p.loss && valid) {
                    const float ab = Aij * 0.5f * (Sij + Sji) * offd;
                    s_lap += ab * (yj * yj - yi * yj);
                    if (!diag) s_lap += ab * (yi * yi - yi * yj);
                }
            }
            if (p.loss && valid) {
                s_size += Sij;
                s_ent += -Sij * logf(Sij) - (1.0f - Sij) * logf(1.0f - Sij);
                if (!diag) {
                    s_size += Sji;
                    s_ent += -Sji * logf(Sji) - (1.0f - Sji) * logf(1.0f - Sji);
                }
            }
            const float gc = Gs * Aij * offd;
            const float gij = (gc + p.c_size + p.c_ent * (logf(1.0f - Sij) - logf(Sij)) * inv_n2) * Sij * (1.0f - Sij);
            const float gji = (gc + p.c_size + p.c_ent * (logf(1.0f - Sji) - logf(Sji)) * inv_n2) * Sji * (1.0f - Sji);
            // torch.optim.Adam, single-tensor form
            mij = mij + (gij - mij) * (1.0f - p.beta1);
            mji = mji + (gji - mji) * (1.0f - p.beta1);
            vij = vij * p.beta2 + (1.0f - p.beta2) * gij * gij;
            vji = vji * p.beta2 + (1.0f - p.beta2) * gji * gji;
            Mij = Mij - step_size * (mij / (sqrtf(vij) / bc2s + p.eps));
            Mji = Mji - step_size * (mji / (sqrtf(vji) / bc2s + p.eps));
            if (valid) {
                p.M[own] = Mij;
                p.mM[own] = mij;
                p.vM[own] = vij;
            }
            tM[j * 33 + i] = Mji;
            tm1[j * 33 + i] = mji;
            tv[j * 33 + i] = vji;
        }
        if (WRITE_ABAR) {
            const float ab = Aij * (0.5f * (sigmoidf_(Mij) + sigmoidf_(Mji))) * offd;
            p.Abar[own] = valid ? ab : 0.0f;
            tA[j * 33 + i] = valid ? ab : 0.0f;
        }
    }
    __syncthreads();
    if (!diag) {
        for (int rr = h; rr < TILE; rr += 2) {
            const size_t g = q + (size_t)(J0 + rr) * ld + I0 + li;
            const bool valid = (J0 + rr < n) && (I0 + li < n);
            if (UPDATE && valid) {
                p.M[g] = tM[rr * 33 + li];
                p.mM[g] = tm1[rr * 33 + li];
                p.vM[g] = tv[rr * 33 + li];
            }
            if (WRITE_ABAR) p.Abar[g] = tA[rr * 33 + li];
        }
    }
    if (UPDATE && p.loss) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s_size += __shfl_xor(s_size, o);
            s_ent += __shfl_xor(s_ent, o);
            s_lap += __shfl_xor(s_lap, o);
        }
        if (lane == 0) {
            float* L = p.loss + ((size_t)tl.t * p.num_iters + iter) * NLOSS;
            atomicAdd(&L[1], p.c_size * s_size);
            atomicAdd(&L[2], p.c_lap * s_lap * inv_n2);
            atomicAdd(&L[3], p.c_ent * s_ent * inv_n2);
        }
    }
    // feature-mask Adam step: once per target, by the wave that owns tile (0,0)
    if (UPDATE && tl.I == 0 && tl.J == 0 && lane < p.D) {
        const int o = tl.t * FS + lane;
        const float fcur = p.f[iter & 1][o];
        const float ph = sigmoidf_(fcur);
        float dsum = 0.0f;
        for (int rb = 0; rb < (ld >> 5); ++rb) dsum += p.df[((size_t)(tm.offR >> 5) + rb) * FS + lane];
        const float gf = (dsum + p.c_feat_size / (float)p.D) * ph * (1.0f - ph);
        float m = p.mf[o], v = p.vf[o];
        m = m + (gf - m) * (1.0f - p.beta1);
        v = v * p.beta2 + (1.0f - p.beta2) * gf * gf;
        p.mf[o] = m;
        p.vf[o] = v;
        p.f[(iter + 1) & 1][o] = fcur - step_size * (m / (sqrtf(v) / bc2s + p.eps));
    }
}

// per target: column-major copy of X, zero the Adam state of the feature mask, initial feature mask
__global__ __launch_bounds__(256) void k_prep(Params p, const float* f_init) {
    const int t = blockIdx.x;
    const TargetMeta tm = p.meta[t];
    const float* X = p.X + tm.offR * FS;
    float* XT = const_cast<float*>(p.XT) + tm.offR * FS;
    for (int e = threadIdx.x; e < tm.ld * FS; e += blockDim.x) {
        const int c = e / tm.ld, i = e - c * tm.ld;
        XT[e] = X[(size_t)i * FS + c];
    }
    if (threadIdx.x < FS) {
        const int o = t * FS + threadIdx.x;
        const float f0 = f_init ? f_init[o] : 0.0f;
        p.f[0][o] = f0;
        p.f[1][o] = f0;
        p.mf[o] = 0.0f;
        p.vf[o] = 0.0f;
    }
}

}  // namespace gnnx
