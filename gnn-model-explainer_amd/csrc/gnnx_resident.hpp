// gnnx_resident.hpp — on-chip-resident mask optimisation for single-tile targets (n <= 32), node mode.
//
// One workgroup (4 waves) per target runs ALL iterations of explainer/explain.py:137-146 in one launch:
//   * the edge mask M and its Adam moments live in registers (thread (i, c4) owns entries (i, c4..c4+3)),
//   * the masked adjacency, the input features and every intermediate of the 3-layer GCN forward/backward
//     (Zraw, U1, U2, dZ2, dZ1, row norms, head vectors) live in LDS,
//   * the contractions Abar.X run on v_mfma_f32_32x32x2_f32 with both operands read from LDS
//     (K = 32 split over the 4 waves, partial tiles reduced through LDS),
//   * phases are separated by __syncthreads() instead of kernel boundaries; HBM is touched only to load the
//     target at the start and to store M, Abar and the feature mask at the end.
// Same mathematics, same algebraic shortcuts and same per-entry ownership rules as the streaming kernels of
// gnnx_kernels.hpp; the streaming path remains the general one (any n, graph mode, loss logging).
#pragma once
#include "gnnx_kernels.hpp"

namespace gnnx {

constexpr int RES_CMAX = 8;

struct ResidentShared {
    float sA[TILE * 33];     // masked adjacency (symmetric), [i][j]
    float sX[TILE * 33];     // input features
    float sZraw[TILE * 33];  // Abar . X
    float sU1[TILE * 33], sU2[TILE * 33];
    float sdZ2[TILE * 33], sdZ1[TILE * 33];
    float wl[3][32 * 33];    // layer weights W1, W2, W3
    float red[4 * TILE * 33];  // split-K partial tiles / partial G tiles
    float zs[TILE * 33];
    float sS[TILE * 33];     // sigma exchange
    float sWp[RES_CMAX * 96];  // prediction head rows (the resident path takes C <= RES_CMAX)
    float sbp[CMAX];
    float rn1[TILE], rn2[TILE], yhat[TILE], g3[TILE], arow[TILE];
    float phi[32], fcur[32], mf[32], vf[32], bias[3][32];
    float z3[32], y3[32], dz3[32], e[96], g[CMAX], dEs[96], dfp[32];
    float sr3;
};

// split-K contraction of the resident tile: Z = Abar . B for the 32 rows, B given by a functor of (k, column).
template <class BFn>
__device__ __forceinline__ void resident_contract(ResidentShared& sh, BFn bval, float (&z4)[4]) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int row = tid >> 3, cg = (tid & 7) * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // wave w owns k in [8w, 8w+8): 4 MFMA steps of 2 k values
        const int k = 8 * wave + 2 * u + h;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sh.sA[li * 33 + k], bval(k, li), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sh.red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j)
        z4[j] = (sh.red[(0 * TILE + row) * 33 + cg + j] + sh.red[(1 * TILE + row) * 33 + cg + j]) +
                (sh.red[(2 * TILE + row) * 33 + cg + j] + sh.red[(3 * TILE + row) * 33 + cg + j]);
}

// forward row-local epilogue: Y = Z W + b, U = Y / max(|Y|, 1e-12); returns this thread's 4 outputs
__device__ __forceinline__ void resident_fwd_epilogue(ResidentShared& sh, const float (&z4)[4], int layer, int din, int dout,
                                                      float* sU, float* srn, float (&u4)[4]) {
    const int tid = threadIdx.x, row = tid >> 3, cg = (tid & 7) * 4;
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
    for (int j = 0; j < 4; ++j) {
        u4[j] = y[j] / rnorm;
        sU[row * 33 + cg + j] = u4[j];
    }
    if ((tid & 7) == 0) srn[row] = rnorm;
}

// sum over the 32 rows of a per-thread 4-vector (thread (row, cg)): result in sh.red[0..31] after the call
__device__ __forceinline__ void resident_colsum(ResidentShared& sh, float (&part)[4], float* out32) {
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
    if (tid < 32) out32[tid] = sh.red[tid] + sh.red[32 + tid] + sh.red[64 + tid] + sh.red[96 + tid];
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_resident32(Params p, const int32_t* targets, const float* adam_tab) {
    __shared__ ResidentShared sh;
    const int t = targets[blockIdx.x];
    const TargetMeta tm = p.meta[t];
    const int n = tm.n, tr = tm.t;  // ld == 32
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int i = tid >> 3, c4 = (tid & 7) * 4;
    const size_t own = tm.offQ + (size_t)i * TILE + c4;

    // ---- load the target once ----
    f32x4 M4 = *reinterpret_cast<const f32x4*>(p.M + own);
    const f32x4 A4 = *reinterpret_cast<const f32x4*>(p.A + own);
    f32x4 m4 = {0.0f, 0.0f, 0.0f, 0.0f}, v4 = m4;
    {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(p.X + (tm.offR + i) * FS + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sh.sX[i * 33 + c4 + e] = x4[e];
        for (int l = 0; l < 3; ++l)
            for (int e = tid; e < 1024; e += 256) sh.wl[l][(e >> 5) * 33 + (e & 31)] = p.wts[WT_W + l * 1024 + e];
        if (tid < 96) sh.bias[tid >> 5][tid & 31] = p.wts[WT_B + tid];
        for (int e = tid; e < p.C * 96; e += 256) sh.sWp[e] = p.wts[WT_WP + e];
        if (tid < CMAX) sh.sbp[tid] = p.wts[WT_BP + tid];
        if (tid < 32) {
            sh.yhat[tid] = p.yhat[tm.offR + tid];
            sh.fcur[tid] = 0.0f;  // construct_feat_mask: constant 0 (explain.py:639-641)
            sh.mf[tid] = 0.0f;
            sh.vf[tid] = 0.0f;
        }
    }
    const float inv_n2 = 1.0f / ((float)n * (float)n);
    const float yi = p.yhat[tm.offR + i];
    const f32x4 yj4 = *reinterpret_cast<const f32x4*>(p.yhat + tm.offR + c4);

    // sigma(M) -> symmetrised masked adjacency in LDS (also used after every update)
    auto publish_abar = [&]() {
        f32x4 S;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            S[e] = sigmoidf_(M4[e]);
            sh.sS[i * 33 + c4 + e] = S[e];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = c4 + e;
            sh.sA[i * 33 + j] = (i != j) ? A4[e] * (0.5f * (S[e] + sh.sS[j * 33 + i])) : 0.0f;
        }
        __syncthreads();
    };
    publish_abar();

    for (int iter = 0; iter < p.num_iters; ++iter) {
        // Adam scalars: the same host-computed values (double, then float) the streaming kernels receive as
        // arguments: adam_tab[2k] = lr / (1 - beta1^(k+1)), adam_tab[2k+1] = sqrt(1 - beta2^(k+1))
        if (tid < 32) sh.phi[tid] = (tid < p.D) ? sigmoidf_(sh.fcur[tid]) : 0.0f;
        __syncthreads();
        const float step_size = adam_tab[2 * iter], inv_bc2s = 1.0f / adam_tab[2 * iter + 1];

        // ---- layer 1: Zraw = Abar . X ; U1 ----
        float z4[4], u4[4];
        resident_contract(sh, [&](int k, int c) { return sh.sX[k * 33 + c]; }, z4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sh.sZraw[i * 33 + c4 + j] = z4[j];
            z4[j] *= sh.phi[c4 + j];
        }
        resident_fwd_epilogue(sh, z4, 0, p.D, p.H, sh.sU1, sh.rn1, u4);
        __syncthreads();
        // ---- layer 2: U2, and row t of Abar . relu(U2) ----
        resident_contract(sh, [&](int k, int c) { return fmaxf(sh.sU1[k * 33 + c], 0.0f); }, z4);
        resident_fwd_epilogue(sh, z4, 1, p.H, p.H, sh.sU2, sh.rn2, u4);
        {
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) part[j] = sh.sA[tr * 33 + i] * fmaxf(u4[j], 0.0f);
            resident_colsum(sh, part, sh.z3);
        }
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
            if (tid < p.H)
#pragma unroll 4
                for (int c = 0; c < p.O; ++c) v = fmaf(sh.zs[c], sh.wl[2][tid * 33 + c], v);
            sh.dz3[tid] = v;
        }
        __syncthreads();
        // ---- dZ2 (rank-1: dX2[i] = Abar[i][t] dZ3[t] + dE2 on row t) and g3 ----
        {
            const float ait = sh.sA[tr * 33 + i];
            float du[4], u[4], dz[4];
            float gpart = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c4 + j;
                u[j] = sh.sU2[i * 33 + c];
                gpart = fmaf(sh.dz3[c], fmaxf(u[j], 0.0f), gpart);
                float dx = ait * sh.dz3[c];
                if (i == tr) dx += sh.dEs[32 + c];
                dx = (u[j] > 0.0f) ? dx : 0.0f;
                du[j] = (c < p.H) ? dx : 0.0f;
            }
            gpart += __shfl_xor(gpart, 1);
            gpart += __shfl_xor(gpart, 2);
            gpart += __shfl_xor(gpart, 4);
            if ((tid & 7) == 0) sh.g3[i] = (i < n) ? gpart : 0.0f;
            rowlocal_backward(du, u, sh.rn2[i], p.H, i, c4, sh.zs, sh.wl[1], dz);
#pragma unroll
            for (int j = 0; j < 4; ++j) sh.sdZ2[i * 33 + c4 + j] = (c4 + j < p.H) ? dz[j] : 0.0f;
        }
        __syncthreads();
        // ---- dX1 = Abar . dZ2 (+ dE1 on row t) -> dZ1 ; feature-mask gradient ----
        resident_contract(sh, [&](int k, int c) { return sh.sdZ2[k * 33 + c]; }, z4);
        {
            float du[4], u[4], dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c4 + j;
                u[j] = sh.sU1[i * 33 + c];
                float dx = z4[j];
                if (i == tr && c < p.H) dx += sh.dEs[c];
                dx = (u[j] > 0.0f) ? dx : 0.0f;
                du[j] = (c < p.H) ? dx : 0.0f;
            }
            rowlocal_backward(du, u, sh.rn1[i], p.H, i, c4, sh.zs, sh.wl[0], dz);
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = (c4 + j < p.D) ? dz[j] : 0.0f;
                sh.sdZ1[i * 33 + c4 + j] = dz[j];
                part[j] = dz[j] * sh.sZraw[i * 33 + c4 + j];
            }
            resident_colsum(sh, part, sh.dfp);
        }
        // ---- G tile = dL/dAbar (+ transpose) on MFMA, K = D + H split over the waves; layer 3 is the rank-2 g3 term
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = wave + 4 * u;  // k-step (columns 2s, 2s+1)
                const int k = 2 * s + h;
                if (2 * s < p.D) {
                    const float z = sh.sdZ1[li * 33 + k];
                    const float x = sh.sX[li * 33 + k] * sh.phi[k];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z, x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, z, acc, 0, 0, 0);
                }
                if (2 * s < p.H) {
                    const float z = sh.sdZ2[li * 33 + k];
                    const float x = fmaxf(sh.sU1[li * 33 + k], 0.0f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(z, x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, z, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sh.red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
        }
        __syncthreads();
        // ---- gradient + Adam on the register-resident mask ----
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = c4 + e;
            float Gsum = (sh.red[(0 * TILE + i) * 33 + j] + sh.red[(1 * TILE + i) * 33 + j]) +
                         (sh.red[(2 * TILE + i) * 33 + j] + sh.red[(3 * TILE + i) * 33 + j]);
            Gsum += (i == tr) ? sh.g3[j] : 0.0f;
            Gsum += (j == tr) ? sh.g3[i] : 0.0f;
            const float dy = yi - yj4[e];
            const float Gs = 0.5f * Gsum + p.c_lap * 0.5f * dy * dy * inv_n2;
            const float gc = (i != j) ? Gs * A4[e] : 0.0f;
            const float S = sigmoidf_(M4[e]);
            const float gij = (gc + p.c_size - p.c_ent * M4[e] * inv_n2) * S * (1.0f - S);
            float Mn = M4[e], mn = m4[e], vn = v4[e];
            adam_update(Mn, mn, vn, gij, p.beta1, p.beta2, p.eps, step_size, inv_bc2s);
            M4[e] = Mn;
            m4[e] = mn;
            v4[e] = vn;
        }
        if (tid < p.D) {  // feature mask
            const float ph = sh.phi[tid];
            const float gf = (sh.dfp[tid] + p.c_feat_size / (float)p.D) * ph * (1.0f - ph);
            float fn = sh.fcur[tid], m = sh.mf[tid], v = sh.vf[tid];
            adam_update(fn, m, v, gf, p.beta1, p.beta2, p.eps, step_size, inv_bc2s);
            sh.fcur[tid] = fn;
            sh.mf[tid] = m;
            sh.vf[tid] = v;
        }
        __syncthreads();
        if (iter + 1 < p.num_iters) publish_abar();  // the returned mask is the one of the LAST forward (explain.py:209-211)
    }
    // ---- store the results ----
    *reinterpret_cast<f32x4*>(p.M + own) = M4;
    f32x4 ab;
#pragma unroll
    for (int e = 0; e < 4; ++e) ab[e] = sh.sA[i * 33 + c4 + e];
    *reinterpret_cast<f32x4*>(p.Abar + own) = ab;
    // final feature mask into the slot the streaming path would have used (copied out by the host afterwards)
    if (tid < FS) p.f[p.num_iters & 1][t * FS + tid] = (tid < p.D) ? sh.fcur[tid] : 0.0f;
}

}  // namespace gnnx
