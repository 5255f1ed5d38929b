// gnnx_resident.hpp — on-chip-resident mask optimisation for small targets (NB = ld/32 <= 3 row blocks, i.e.
// n <= 96), node mode.
//
// One workgroup (4 waves) per target runs ALL iterations of explainer/explain.py:137-146 in one launch:
//   * the edge mask M and its Adam moments live in registers: for every tile pair {(I,J),(J,I)}, I <= J, thread
//     (i, c4) owns the entries (I0+i, J0+c4..+3) and, off the diagonal, their mirror entries (J0+c4.., I0+i) - the
//     ownership rule of the streaming k_mask, so the output is bitwise symmetric and deterministic;
//   * the masked adjacency (upper tiles), the input features and every intermediate of the 3-layer GCN
//     forward/backward (Zraw, U1, U2, dZ2, dZ1, row norms, head vectors) live in LDS;
//   * the contractions Abar.X run on v_mfma_f32_32x32x2_f32 with BOTH operands read from LDS (per 32-row block:
//     K = 32 NB split over the 4 waves, partial tiles reduced through LDS);
//   * phases are separated by __syncthreads() instead of kernel boundaries; HBM is touched only to load the
//     target at the start and to store M, Abar and the feature mask at the end.
// Same mathematics, same algebraic shortcuts and same per-entry ownership as the streaming kernels of
// gnnx_kernels.hpp; the streaming path remains the general one (any n, graph mode, loss logging).
#pragma once
#include "gnnx_kernels.hpp"

namespace gnnx {

constexpr int RES_CMAX = 8;   // classes the resident path takes
constexpr int RES_NBMAX = 3;  // row blocks the resident path takes (LDS: ~145 KB of the CU's 160 KB at NB = 3)

__host__ __device__ constexpr int res_pairs(int nb) { return nb * (nb + 1) / 2; }
// index of the upper tile (I <= J) in the order (0,0) (0,1) .. (0,NB-1) (1,1) ..
__host__ __device__ constexpr int res_pair_index(int nb, int I, int J) { return I * nb - I * (I - 1) / 2 + (J - I); }

template <int NB>
struct ResidentShared {
    static constexpr int LD = NB * TILE;
    float sA[res_pairs(NB)][TILE * 33];  // masked adjacency, upper tiles [I<=J][i][j] (the matrix is symmetric)
    float sX[LD * 33];                   // input features
    float sZraw[LD * 33];                // Abar . X
    float sU1[LD * 33], sU2[LD * 33];
    float sdZ2[LD * 33], sdZ1[LD * 33];
    float wl[3][32 * 33];      // layer weights W1, W2, W3
    float red[4 * TILE * 33];  // split-K partial tiles / partial G tiles
    float zs[TILE * 33];
    float sS[TILE * 33];       // sigma exchange on diagonal tiles
    float sWp[RES_CMAX * 96];  // prediction head rows
    float sbp[CMAX];
    float rn1[LD], rn2[LD], yhat[LD], g3[LD];
    float phi[32], fcur[32], mf[32], vf[32], bias[3][32];
    float z3[32], y3[32], dz3[32], e[96], g[CMAX], dEs[96], dfp[32], col32[32];
    float sr3;
};

// element (r, c) of the symmetric masked adjacency from its upper tiles
template <int NB>
__device__ __forceinline__ float res_abar(const ResidentShared<NB>& sh, int r, int c) {
    const int rb = r >> 5, cb = c >> 5;
    return (rb <= cb) ? sh.sA[res_pair_index(NB, rb, cb)][(r & 31) * 33 + (c & 31)]
                      : sh.sA[res_pair_index(NB, cb, rb)][(c & 31) * 33 + (r & 31)];
}

// split-K contraction for the 32 rows of block I: Z = Abar[I-block, :] . B, B given by a functor of (row k, column).
template <int NB, class BFn>
__device__ __forceinline__ void resident_contract(ResidentShared<NB>& sh, int I, BFn bval, float (&z4)[4]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int row = tid >> 3, cg = (tid & 7) * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        // A operand A[rho = li][kappa = k] = Abar[I0+li][J0+k]: the stored tile if I <= J, its transpose otherwise
        const bool upper = I <= J;
        const float* T = sh.sA[upper ? res_pair_index(NB, I, J) : res_pair_index(NB, J, I)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // wave w owns k in [8w, 8w+8) of every tile: 4 MFMA steps of 2 k values
            const int k = 8 * wave + 2 * u + h;
            const float a = upper ? T[li * 33 + k] : T[k * 33 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bval(J * TILE + k, li), acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sh.red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
        z4[j] = (sh.red[(0 * TILE + row) * 33 + cg + j] + sh.red[(1 * TILE + row) * 33 + cg + j]) +
                (sh.red[(2 * TILE + row) * 33 + cg + j] + sh.red[(3 * TILE + row) * 33 + cg + j]);
}

// forward row-local epilogue for the rows of block I: Y = Z W + b, U = Y / max(|Y|, 1e-12)
template <int NB>
__device__ __forceinline__ void resident_fwd_epilogue(ResidentShared<NB>& sh, const float (&z4)[4], int I, int layer, int din,
                                                      int dout, float* sU, float* srn) {
    const int tid = threadIdx.x, row = tid >> 3, cg = (tid & 7) * 4;
    const int gr = I * TILE + row;
#pragma unroll
    for (int j = 0; j < 4; ++j) sh.zs[row * 33 + cg + j] = z4[j];
    __syncthreads();
    float y[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int k = 0; k < din; ++k) {
        const float z = sh.zs[row * 33 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = fmaf(z, sh.wl[layer][k * 33 + cg + j], y[j]);
    }
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        y[j] = (cg + j < dout) ? y[j] + sh.bias[layer][cg + j] : 0.0f;
        ss = fmaf(y[j], y[j], ss);
    }
    ss += __shfl_xor(ss, 1);
    ss += __shfl_xor(ss, 2);
    ss += __shfl_xor(ss, 4);
    const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int j = 0; j < 4; ++j) sU[gr * 33 + cg + j] = y[j] / rnorm;
    if ((tid & 7) == 0) srn[gr] = rnorm;
}

// sum over the 32 rows the workgroup is working on of a per-thread 4-vector (thread (row, cg)) -> sh.col32[0..31]
template <int NB>
__device__ __forceinline__ void resident_colsum(ResidentShared<NB>& sh, float (&part)[4]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, cg = (tid & 7) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[j] += __shfl_xor(part[j], 8);
        part[j] += __shfl_xor(part[j], 16);
        part[j] += __shfl_xor(part[j], 32);
    }
    __syncthreads();
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh.red[wave * 32 + cg + j] = part[j];
    }
    __syncthreads();
    if (tid < 32) sh.col32[tid] = sh.red[tid] + sh.red[32 + tid] + sh.red[64 + tid] + sh.red[96 + tid];
    __syncthreads();
}

template <int NB>
__global__ __launch_bounds__(256) void k_resident(Params p, const int32_t* targets, const float* adam_tab) {
    constexpr int P = res_pairs(NB);
    constexpr int LD = NB * TILE;
    __shared__ ResidentShared<NB> sh;
    const int t = targets[blockIdx.x];
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, tr = tm.t;  // tm.ld == LD
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int i = tid >> 3, c4 = (tid & 7) * 4;

    // ---- load the target once: own entries (I0+i, J0+c4..) and mirror entries (J0+c4.., I0+i) of every tile pair ----
    f32x4 Mo[P], mo[P], vo[P], Ao[P], Mp[P], mp[P], vp[P];
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) {
            const int pi = res_pair_index(NB, I, J);
            const size_t own = tm.offQ + (size_t)(I * TILE + i) * LD + J * TILE + c4;
            Mo[pi] = *reinterpret_cast<const f32x4*>(p.M + own);
            Ao[pi] = *reinterpret_cast<const f32x4*>(p.A + own);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const size_t mir = tm.offQ + (size_t)(J * TILE + c4 + e) * LD + I * TILE + i;
                mo[pi][e] = p.m_in ? p.m_in[own + e] : 0.0f;   // gnnx_run_resume: Adam moments (zeros otherwise)
                vo[pi][e] = p.v_in ? p.v_in[own + e] : 0.0f;
                mp[pi][e] = (p.m_in && I != J) ? p.m_in[mir] : 0.0f;
                vp[pi][e] = (p.v_in && I != J) ? p.v_in[mir] : 0.0f;
                Mp[pi][e] = (I != J) ? p.M[mir] : 0.0f;
            }
        }
#pragma unroll
    for (int I = 0; I < NB; ++I) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(p.X + (tm.offR + I * TILE + i) * FS + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sh.sX[(I * TILE + i) * 33 + c4 + e] = x4[e];
    }
    for (int l = 0; l < 3; ++l)
        for (int e = tid; e < 1024; e += 256) sh.wl[l][(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + l * 1024 + e];
    if (tid < 96) sh.bias[tid >> 5][tid & 31] = p.wts[WT_B + tid];
    for (int e = tid; e < p.C * 96; e += 256) sh.sWp[e] = p.wts[WT_WP + e];
    if (tid < CMAX) sh.sbp[tid] = p.wts[WT_BP + tid];
    if (tid < LD) sh.yhat[tid] = p.yhat[tm.offR + tid];
    if (tid < 32) {
        const float* fs = p.fs_in ? p.fs_in + (size_t)t * 3 * FS + tid : nullptr;   // gnnx_run_resume
        sh.fcur[tid] = (fs && tid < p.D) ? fs[0] : 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
        sh.mf[tid] = (fs && tid < p.D) ? fs[FS] : 0.0f;
        sh.vf[tid] = (fs && tid < p.D) ? fs[2 * FS] : 0.0f;
    }
    const float inv_n2 = 1.0f / ((float)n * (float)n);

    // sigma(M) -> symmetrised masked adjacency tiles in LDS (at the start and after every update)
    auto publish_abar = [&]() {
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
                const int pi = res_pair_index(NB, I, J);
                f32x4 S;
#pragma unroll
                for (int e = 0; e < 4; ++e) S[e] = sigmoidf_(Mo[pi][e]);
                if (I == J) {  // diagonal tile: the mirror entry belongs to another thread -> exchange sigma through LDS
                    if (I > 0) __syncthreads();  // the previous diagonal tile's sigmas have been consumed
#pragma unroll
                    for (int e = 0; e < 4; ++e) sh.sS[i * 33 + c4 + e] = S[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int j = c4 + e;
                        sh.sA[pi][i * 33 + j] = (i != j) ? Ao[pi][e] * (0.5f * (S[e] + sh.sS[j * 33 + i])) : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sh.sA[pi][i * 33 + c4 + e] = Ao[pi][e] * (0.5f * (S[e] + sigmoidf_(Mp[pi][e])));
                }
            }
        __syncthreads();
    };
    publish_abar();

    for (int iter = 0; iter < p.num_iters; ++iter) {
        // Adam scalars: the same host-computed values (double, then float) the streaming kernels receive as
        // arguments: adam_tab[2k] = lr / (1 - beta1^(k+1)), adam_tab[2k+1] = sqrt(1 - beta2^(k+1))
        if (tid < 32) {
            sh.phi[tid] = (tid < p.D) ? sigmoidf_(sh.fcur[tid]) : 0.0f;
            sh.z3[tid] = 0.0f;
            sh.dfp[tid] = 0.0f;
        }
        __syncthreads();
        const float step_size = adam_tab[2 * iter], bc2s = adam_tab[2 * iter + 1];

        float z4[4];
        // ---- layer 1: Zraw = Abar . X ; U1 ----
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            resident_contract<NB>(sh, I, [&](int k, int c) { return sh.sX[k * 33 + c]; }, z4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sh.sZraw[(I * TILE + i) * 33 + c4 + j] = z4[j];
                z4[j] *= sh.phi[c4 + j];
            }
            resident_fwd_epilogue<NB>(sh, z4, I, 0, p.D, p.H, sh.sU1, sh.rn1);
            __syncthreads();
        }
        // ---- layer 2: U2 ----
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            resident_contract<NB>(sh, I, [&](int k, int c) { return fmaxf(sh.sU1[k * 33 + c], 0.0f); }, z4);
            resident_fwd_epilogue<NB>(sh, z4, I, 1, p.H, p.H, sh.sU2, sh.rn2);
            __syncthreads();
        }
        // ---- row t of Abar . relu(U2) (the only row of layer 3 the reference reads, explain.py:713) ----
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int gr = I * TILE + i;
            const float atr = res_abar<NB>(sh, tr, gr);
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] = atr * fmaxf(sh.sU2[gr * 33 + c4 + j], 0.0f);
            resident_colsum<NB>(sh, part);
            if (tid < 32) sh.z3[tid] += sh.col32[tid];
        }
        __syncthreads();
        // ---- layer 3 (row t only), head, dE, dZ3[t] ----
        if (tid < 64) {
            const int c = tid & 31;
            float y = 0.0f;
            if (c < p.O) {
#pragma unroll 4
                for (int k = 0; k < p.H; ++k) y = fmaf(sh.z3[k], sh.wl[2][k * 33 + c], y);
                y += sh.bias[2][c];
            }
            float ss = (tid < 32) ? y * y : 0.0f;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
            const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
            if (tid < 32) {
                const float u = y / rnorm;
                sh.y3[tid] = u;
                sh.e[64 + tid] = u;
                sh.e[tid] = (tid < p.H) ? fmaxf(sh.sU1[tr * 33 + tid], 0.0f) : 0.0f;
                sh.e[32 + tid] = (tid < p.H) ? fmaxf(sh.sU2[tr * 33 + tid], 0.0f) : 0.0f;
            }
            if (tid == 0) sh.sr3 = rnorm;
        }
        __syncthreads();
        {   // softmax head (explain.py:713-714, 750-753): g = p - onehot(y_gt), dE = Wp^T g
            if (tid < 64) {
                // logits: class c = lane / 8 (C <= 8), the 96-term dot split over 8 lanes, then moved to lane c
                const int c = tid >> 3, part = tid & 7;
                float s = 0.0f;
                if (c < p.C)
                    for (int q = part * 12; q < part * 12 + 12; ++q) s = fmaf(sh.sWp[c * 96 + q], sh.e[q], s);
                s += __shfl_xor(s, 1);
                s += __shfl_xor(s, 2);
                s += __shfl_xor(s, 4);
                const float zc = __shfl(s, (tid & 7) * 8);  // lane l < 8 receives the sum of class l
                float z = (tid < p.C) ? zc + sh.sbp[tid] : -3.0e38f;
                float mx = z;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
                const float ex = (tid < p.C) ? expf(z - mx) : 0.0f;
                float sum = ex;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
                if (tid < CMAX) sh.g[tid] = (tid < p.C) ? ex / sum - ((tid == tm.y_gt) ? 1.0f : 0.0f) : 0.0f;
            }
            __syncthreads();
            if (tid < 96) {
                float s = 0.0f;
#pragma unroll 4
                for (int c = 0; c < p.C; ++c) s = fmaf(sh.sWp[c * 96 + tid], sh.g[c], s);
                sh.dEs[tid] = s;
            }
            __syncthreads();
        }
        if (tid < 64) {
            const int c = tid & 31;
            const float du = (tid < 32 && c < p.O) ? sh.dEs[64 + c] : 0.0f;
            const float u = (tid < 32) ? sh.y3[c] : 0.0f;
            float s = du * u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if (tid < 32) sh.zs[c] = (du - u * s) / sh.sr3;  // dY3[t]
        }
        __syncthreads();
        if (tid < 32) {
            float v = 0.0f;
            if (tid < p.H) {
#pragma unroll 4
                for (int c = 0; c < p.O; ++c) v = fmaf(sh.zs[c], sh.wl[2][tid * 33 + c], v);
            }
            sh.dz3[tid] = v;
        }
        __syncthreads();
        // ---- dZ2 (rank-1: dX2[r] = Abar[r][t] dZ3[t] + dE2 on row t) and g3 ----
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int gr = I * TILE + i;
            const float ait = res_abar<NB>(sh, tr, gr);
            float du[4], u[4], dz[4];
            float gpart = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c4 + j;
                u[j] = sh.sU2[gr * 33 + c];
                gpart = fmaf(sh.dz3[c], fmaxf(u[j], 0.0f), gpart);
                float dx = ait * sh.dz3[c];
                if (gr == tr) dx += sh.dEs[32 + c];
                dx = (u[j] > 0.0f) ? dx : 0.0f;
                du[j] = (c < p.H) ? dx : 0.0f;
            }
            gpart += __shfl_xor(gpart, 1);
            gpart += __shfl_xor(gpart, 2);
            gpart += __shfl_xor(gpart, 4);
            if ((tid & 7) == 0) sh.g3[gr] = (gr < n) ? gpart : 0.0f;
            rowlocal_backward(du, u, sh.rn2[gr], p.H, i, c4, sh.zs, sh.wl[1], dz);
#pragma unroll
            for (int j = 0; j < 4; ++j) sh.sdZ2[gr * 33 + c4 + j] = (c4 + j < p.H) ? dz[j] : 0.0f;
        }
        __syncthreads();
        // ---- dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 ; feature-mask gradient ----
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int gr = I * TILE + i;
            resident_contract<NB>(sh, I, [&](int k, int c) { return sh.sdZ2[k * 33 + c]; }, z4);
            float du[4], u[4], dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c4 + j;
                u[j] = sh.sU1[gr * 33 + c];
                float dx = z4[j];
                if (gr == tr && c < p.H) dx += sh.dEs[c];
                dx = (u[j] > 0.0f) ? dx : 0.0f;
                du[j] = (c < p.H) ? dx : 0.0f;
            }
            rowlocal_backward(du, u, sh.rn1[gr], p.H, i, c4, sh.zs, sh.wl[0], dz);
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = (c4 + j < p.D) ? dz[j] : 0.0f;
                sh.sdZ1[gr * 33 + c4 + j] = dz[j];
                part[j] = dz[j] * sh.sZraw[gr * 33 + c4 + j];
            }
            resident_colsum<NB>(sh, part);
            if (tid < 32) sh.dfp[tid] += sh.col32[tid];
        }
        __syncthreads();
        // ---- per tile pair: G = dL/dAbar (+ transpose) on MFMA (K = D + H split over the waves; layer 3 is the
        //      rank-2 g3 term), then the gradient and Adam on the register-resident entries ----
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = I; J < NB; ++J) {
                const int pi = res_pair_index(NB, I, J);
                const int I0 = I * TILE, J0 = J * TILE;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int s = wave + 4 * u;  // k-step (columns 2s, 2s+1)
                    const int k = 2 * s + h;
                    if (2 * s < p.D) {
                        const float zi = sh.sdZ1[(I0 + li) * 33 + k], zj = sh.sdZ1[(J0 + li) * 33 + k];
                        const float xi = sh.sX[(I0 + li) * 33 + k] * sh.phi[k], xj = sh.sX[(J0 + li) * 33 + k] * sh.phi[k];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zi, xj, acc, 0, 0, 0);  // G[i][j]
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, zj, acc, 0, 0, 0);  // G[j][i]
                    }
                    if (2 * s < p.H) {
                        const float zi = sh.sdZ2[(I0 + li) * 33 + k], zj = sh.sdZ2[(J0 + li) * 33 + k];
                        const float xi = fmaxf(sh.sU1[(I0 + li) * 33 + k], 0.0f), xj = fmaxf(sh.sU1[(J0 + li) * 33 + k], 0.0f);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zi, xj, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, zj, acc, 0, 0, 0);
                    }
                }
                if (pi > 0) __syncthreads();  // the previous pair's partial sums have been read
#pragma unroll
                for (int r = 0; r < 16; ++r) sh.red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
                __syncthreads();
                const int gi = I0 + i;
                const float yi = sh.yhat[gi], g3i = sh.g3[gi];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = c4 + e, gj = J0 + j;
                    float Gsum = (sh.red[(0 * TILE + i) * 33 + j] + sh.red[(1 * TILE + i) * 33 + j]) +
                                 (sh.red[(2 * TILE + i) * 33 + j] + sh.red[(3 * TILE + i) * 33 + j]);
                    Gsum += (gi == tr) ? sh.g3[gj] : 0.0f;
                    Gsum += (gj == tr) ? g3i : 0.0f;
                    const float dy = yi - sh.yhat[gj];
                    const float Gs = 0.5f * Gsum + p.c_lap * 0.5f * dy * dy * inv_n2;
                    const float gc = (gi != gj) ? Gs * Ao[pi][e] : 0.0f;
                    {
                        const float S = sigmoidf_(Mo[pi][e]);
                        const float gij = (gc + p.c_size - p.c_ent * Mo[pi][e] * inv_n2) * S * (1.0f - S);
                        float Mn = Mo[pi][e], mn = mo[pi][e], vn = vo[pi][e];
                        adam_update(Mn, mn, vn, gij, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
                        Mo[pi][e] = Mn;
                        mo[pi][e] = mn;
                        vo[pi][e] = vn;
                    }
                    if (I != J) {  // the mirror entry (gj, gi) belongs to this thread too
                        const float S = sigmoidf_(Mp[pi][e]);
                        const float gji = (gc + p.c_size - p.c_ent * Mp[pi][e] * inv_n2) * S * (1.0f - S);
                        float Mn = Mp[pi][e], mn = mp[pi][e], vn = vp[pi][e];
                        adam_update(Mn, mn, vn, gji, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
                        Mp[pi][e] = Mn;
                        mp[pi][e] = mn;
                        vp[pi][e] = vn;
                    }
                }
            }
        if (tid < p.D) {  // feature mask
            const float ph = sh.phi[tid];
            const float gf = (sh.dfp[tid] + p.c_feat_size / (float)p.D) * ph * (1.0f - ph);
            float fn = sh.fcur[tid], m = sh.mf[tid], v = sh.vf[tid];
            adam_update(fn, m, v, gf, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            sh.fcur[tid] = fn;
            sh.mf[tid] = m;
            sh.vf[tid] = v;
        }
        __syncthreads();
        if (iter + 1 < p.num_iters) publish_abar();  // the returned mask is the one of the LAST forward (explain.py:209-211)
    }
    // ---- store the results: M (own + mirror entries), Abar of the last forward (both orientations), feature mask ----
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = I; J < NB; ++J) {
            const int pi = res_pair_index(NB, I, J);
            const size_t own = tm.offQ + (size_t)(I * TILE + i) * LD + J * TILE + c4;
            *reinterpret_cast<f32x4*>(p.M + own) = Mo[pi];
            if (p.m_out) *reinterpret_cast<f32x4*>(p.m_out + own) = mo[pi];
            if (p.v_out) *reinterpret_cast<f32x4*>(p.v_out + own) = vo[pi];
            f32x4 ab;
#pragma unroll
            for (int e = 0; e < 4; ++e) ab[e] = sh.sA[pi][i * 33 + c4 + e];
            *reinterpret_cast<f32x4*>(p.Abar + own) = ab;
            if (I != J) {
                const size_t mirror_row = tm.offQ + (size_t)(J * TILE + i) * LD + I * TILE + c4;  // row i of tile (J, I)
                f32x4 abT;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    abT[e] = sh.sA[pi][(c4 + e) * 33 + i];
                    const size_t mir = tm.offQ + (size_t)(J * TILE + c4 + e) * LD + I * TILE + i;
                    p.M[mir] = Mp[pi][e];
                    if (p.m_out) p.m_out[mir] = mp[pi][e];
                    if (p.v_out) p.v_out[mir] = vp[pi][e];
                }
                *reinterpret_cast<f32x4*>(p.Abar + mirror_row) = abT;
            }
        }
    // final feature mask into the slot the streaming path would have used (copied out by the host afterwards)
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < p.D) ? sh.fcur[tid] : 0.0f;
    if (p.fs_out && tid < FS) {
        float* fs = p.fs_out + (size_t)t * 3 * FS + tid;
        fs[0] = (tid < p.D) ? sh.fcur[tid] : 0.0f;
        fs[FS] = (tid < p.D) ? sh.mf[tid] : 0.0f;
        fs[2 * FS] = (tid < p.D) ? sh.vf[tid] : 0.0f;
    }
}

}  // namespace gnnx
