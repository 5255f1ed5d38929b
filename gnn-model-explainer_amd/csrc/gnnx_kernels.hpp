// gnnx_kernels.hpp — CDNA4 (gfx950) device code of the GNNExplainer mask-optimisation loop.
//
// One iteration of the reference loop (explainer/explain.py:137-146) for ALL targets of a batch:
//
//  node mode (GcnEncoderNode, 5 launches)            graph mode (GcnEncoderGraph, 8 launches)
//   k_conv<FWD1>   Zraw = Abar.X ; U1                 k_conv<FWD1>, k_conv<FWD2>, k_conv<FWD3>
//   k_conv<FWD2>   U2                                 k_head          max-pool head, dE, argmax rows
//   k_node_head    row t of layer 3, head, dE,        k_conv<BWD3>    row-local -> dZ3
//                  dZ3[t], dZ2 (rank-1), g3           k_conv<BWD2>    Abar.dZ3 -> dZ2
//   k_conv<BWD1>   Abar.dZ2 -> dZ1, df partials       k_conv<BWD1>    Abar.dZ2 -> dZ1, df partials
//   k_mask<..>     fused G-tile + regularisers + Adam k_mask<..>
//
//   k_conv   masked-adjacency contraction on MFMA (exact f32, v_mfma_f32_32x32x2_f32): one workgroup per
//            32-row block, K split over its 4 waves, operand loads issued 16 k-steps deep, LDS split-K
//            reduction, fused row-local epilogue (.W + b, L2-normalise / its Jacobian, ReLU mask, .W^T).
//            models.py:58-80, 230-267
//   k_mask   one wave per tile pair {(I,J),(J,I)}: G = dL/dAbar tile on MFMA (never materialised), Laplacian
//            + size + entropy gradients, sigmoid', Adam on (M, m, v), the next symmetrised masked adjacency,
//            loss partial sums wave-reduced.  explain.py:665-678, 755-770, 780-793; utils/train_utils.py:9-10
//
// Exact algebraic shortcuts used in node mode: only row t of the last layer is ever read
// (explain.py:713), so dZ3 has one non-zero row, Abar.dZ3 is rank-1 and the layer-3 part of G is rank-2;
// and colsum((Abar.dZ1) * X) == colsum(dZ1 * (Abar.X)) because Abar is symmetric, so the feature-mask
// gradient needs no fourth contraction.
// Math: SURVEY.md Appendix A; CPU spec: oracle/closed_form.py (tests only).
#pragma once
#include <stdint.h>

namespace gnnx {

constexpr int TILE = 32;  // MFMA 32x32x2 tile edge
constexpr int FS = 32;    // floats per feature row
constexpr int CMAX = 32;  // max classes
constexpr int NLOSS = 16;   // per (target, iteration): pred, size, lap, ent, feat_size, mask density (after the step), 2 spare, class probabilities [LOGP .. LOGP + 8)
constexpr int LOGD = 5, LOGP = 8, LOGPN = 8;

// offsets (floats) inside the packed, zero-padded model block
constexpr int WT_W = 0;                   // W_l   [32][32]  at WT_W + l*1024   (row = input k, col = output c)
constexpr int WT_B = 3 * 1024;            // b_l   [32]      at WT_B + l*32
constexpr int WT_WP = WT_B + 3 * 32;      // Wp    [32][96]  (class c, l*32 + j)
constexpr int WT_BP = WT_WP + CMAX * 96;  // bp    [32]
constexpr int WT_TOTAL = WT_BP + CMAX;

struct TargetMeta {
    int32_t n;     // sub-graph nodes
    int32_t ld;    // round_up(n, 32)
    int32_t t;     // target row (node mode)
    int32_t y_gt;  // ground-truth label
    int64_t offQ;  // float offset of the ld x ld block in square arrays
    int64_t offR;  // row offset in row arrays
};

// tile-table entries carry a copy of the target's meta: one dependent load per workgroup instead of two
struct ConvTile { int32_t t, rb; TargetMeta tm; };               // target, 32-row block
struct MaskTile { int32_t t, I, J; int32_t pad; TargetMeta tm; };  // target, tile pair I <= J
// Work unit of the masked-adjacency contraction: nrb in {1, 2, 4} adjacent 32-row blocks of one target from row block rb, over
// the K range [kb, ke).  Rows of large targets are cut into nks such slices (this is slice ks); each slice leaves its partial
// tile of row block rb + j in slab part + j * nks + ks and k_conv_reduce (the next launch: no in-kernel fences) sums the
// slabs in slice order and runs the row-local epilogue.  nks = 1: the unit runs the epilogue itself.
struct ConvUnit { int32_t t, rb, nrb, kb, ke, ks, nks, part; TargetMeta tm; };
// a row block whose slices meet in slabs part .. part + nks - 1
struct ConvJoin { int32_t t, rb, nks, part; TargetMeta tm; };

struct Params {
    const TargetMeta* meta;
    const float* A;     // adjacency (symmetric, zero padded)
    float* M;           // edge-mask parameter
    float* mM;          // Adam first moment
    float* vM;          // Adam second moment
    float* Abar;        // masked adjacency of the current iterate
    const float* X;     // input features [R][32]
    const float* XT;    // per target column-major copy [32][ld]
    const float* yhat;  // predicted class ids as float [R]
    float* Zraw;        // Abar . X (before the feature mask) [R][32]
    float* U[3];        // normalised pre-activations, row-major [R][32]
    float* UT[3];       // same, per target column-major [32][ld]
    float* rn[3];       // row norms [R]
    float* dZ[3];       // gradients w.r.t. the aggregated inputs, row-major [R][32]
    float* dZT[3];      // same, column-major
    float* g3;          // node mode: layer-3 part of row t of G, one float per row [R]
    float* cpart;       // partial tiles of the K slices of k_conv, 32 x 32 floats each
    float* z3p;         // node mode: per row block partial of row t of Abar . relu(U2)  [R/32][32]
    float* dE;          // direct gradient of the concatenated embedding [T][3][32]
    int32_t* argrow;    // row that receives dE[l][c]  [T][3][32]
    float* df;          // feature-mask gradient partials, one row of 32 per 32-row block [R/32][32]
    float* f[2];        // feature-mask parameter, ping-pong by iteration parity [T][32]
    float* mf;
    float* vf;
    float* probs;       // softmax of the head [T][CMAX]
    float* loss;        // [T][num_iters][NLOSS] or null
    const float* wts;   // packed model block
    // --bn (apply_bn, models.py:222-228, 241-253): after the ReLU of the two hidden layers every node's activation row is
    // standardised over its features (a fresh BatchNorm1d(num_nodes) in training mode: biased variance, eps 1e-5, weight 1,
    // bias 0).  Xn[l] / XnT[l] = the standardised activations of layer l (what the next layer, the head and the G product
    // consume instead of relu(U[l])), bnr[l] = 1 / std per row.  Null unless bn.
    float* Xn[2];
    float* XnT[2];
    float* bnr[2];
    // resume (gnnx_run_resume): Adam state to start from / to hand back; null = zeros / not wanted
    const float* m_in;   // exp_avg of M, packed square layout
    const float* v_in;   // exp_avg_sq of M
    const float* fs_in;  // [T][3][32]: feat_mask, exp_avg, exp_avg_sq
    float* m_out;
    float* v_out;
    float* fs_out;
    // decision trace (gnnx_set_trace; the LOG form of the sparse resident kernel): which side of every ReLU gate that reaches the loss
    // each iteration's forward took, and the rows the graph-mode max-pools picked.  Null = not recorded.
    uint32_t* trace_gates;  // [num_iters][R][2]: bit c of word (iter, row, l) = U_{l+1}[row][c] > 0 (models.py:241, 251); rows outside the layer's row set: 0
    int32_t* trace_pool;    // graph mode [T][num_iters][96]: arg-max row of pooled column (layer, c) (models.py:283, 291, 300)
    int64_t trace_rows;     // R
    int32_t bn;
    int32_t D, H, O, C;
    int32_t graph_mode;
    int32_t num_iters;
    int32_t opt;        // 0 Adam, 1 SGD with momentum, 2 RMSprop, 3 Adagrad (gnnx_hyper.opt)
    int32_t edge_only;  // gnnx_hyper.edge_results_only: the edge-sparse kernels write Abar on the edges only
    float lr, beta2, eps;   // beta2: Adam's second-moment decay, or RMSprop's alpha
    float omb1, omb2;   // (float)(1 - beta1), (float)(1 - beta2) with the subtraction in DOUBLE, as torch passes them to lerp_ / addcmul_
    float c_size, c_feat_size, c_ent, c_lap;
};

// Room beside a resident workgroup (round 5).  A workgroup of the edge-sparse resident kernels used to take ALL of its compute unit: 162 KB of
// the 160 KB + of LDS and 2 waves x 256 registers per lane and SIMD.  The short kernels of a pipelined job's prepare stage (k-hop, packing, edge
// counts, edge lists of the NEXT batches) then waited for a compute unit to drain - 5 to 30 times their own duration (profiles/r05_kernel_stats_
// syn1.csv), and three preparing threads could not keep the optimisations fed.  Now the mixed resident kernel is capped at GNNX_MIXED_NUM_VGPR
// registers and its pool leaves 5 KB of LDS, and the service kernels are capped at GNNX_SERVICE_NUM_VGPR: their workgroups find a place BESIDE
// a resident workgroup (whose waves issue 22 % of the time).  The attribute counts HALF of gfx950's unified register file (112 -> 224 per lane).
#if defined(__HIPCC__)
#define GNNX_NUM_VGPR_ATTR(n) __attribute__((amdgpu_num_vgpr(n)))
#else
#define GNNX_NUM_VGPR_ATTR(n)      // (the CPU emulator of tests/emu compiles the same sources)
#endif
#ifndef GNNX_SERVICE_NUM_VGPR
#define GNNX_SERVICE_NUM_VGPR 16   // 32 registers per lane: two service waves fit the 64 a resident pair of waves leaves on a SIMD
#endif
#if GNNX_SERVICE_NUM_VGPR
#define GNNX_SERVICE_ATTR GNNX_NUM_VGPR_ATTR(GNNX_SERVICE_NUM_VGPR)
#else
#define GNNX_SERVICE_ATTR
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Transcendentals and divisions of the update path: hardware forms (v_rcp_f32, v_sqrt_f32, v_exp_f32 on x log2(e): 1-4 ulp).
// -DGNNX_IEEE_MATH builds the IEEE forms instead (correctly rounded division / square root, 1-ulp expf, the Adam quotient in
// torch's operation order).  Measured in round 3 with the windowed parity test against the reference's own optimiser state
// (tests/test_windowed_parity.py, tools/gpu_r3c.sh): IEEE vs hardware forms = 2418 vs 2414 of 2420 syn1 windows, 2212 vs 2210 of
// 2213 syn4, 8923 vs 8917 of 8963 syn5, 3710 vs 3719 of 3803 config-4 windows within 1e-5 - no measurable difference (what the
// remaining windows amplify is summation order, not these ulps) - for a syn1 batch of 4.32 instead of 3.89 ms: every one of these
// operations sits on the per-iteration latency chain.  So the hardware forms stay the default.
#ifndef GNNX_IEEE_MATH
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float exp_(float x) { return __expf(x); }
#else
__device__ __forceinline__ float rcp_(float x) { return 1.0f / x; }
__device__ __forceinline__ float sqrt_(float x) { return sqrtf(x); }
__device__ __forceinline__ float exp_(float x) { return expf(x); }
#endif
__device__ __forceinline__ float sigmoidf_(float x) { return rcp_(1.0f + exp_(-x)); }

// From here on the value of a register is unknown to the optimiser (no instruction is emitted): expressions derived from it are formed where
// they are used instead of being kept in registers across a whole loop.  (tests/emu/include/hip/hip_runtime.h defines it for the host compiler.)
#ifndef GNNX_OPAQUE
#define GNNX_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
// v + (v of lane ^ 32) / v + (v of lane ^ 16) in every lane.  gfx950's lane swaps (v_permlane32_swap: the upper half of the first operand <->
// the lower half of the second; v_permlane16_swap: odd 16-lane rows of the first <-> even rows of the second) on two copies of v leave
// {lo, lo} / {hi, hi} (resp. {r0, r0, r2, r2} / {r1, r1, r3, r3}) in the pair: their sum is the shuffle form's, operand for operand (the addition
// commutes: bit-identical in all 64 lanes, tools/micro/permlane_swap.hip) - a VALU instruction instead of a ds_bpermute round trip through the
// LDS crossbar: 25.1 ns instead of 38.5 / 36.8 ns per dependent step (profiles/r05_permlane_swap.txt), and no lgkmcnt wait shared with the
// loads in flight.  Inline assembly: with this compiler (ROCm 7.2) the sum of the builtin's two results comes out as r[0] + r[0].  The hazard
// recogniser puts TWO wait states (`s_nop 1`) between the VALU write of an operand (the `b = a` copy) and the lane swap that reads it - checked
// on the ISA hipcc emits for __builtin_amdgcn_permlane32_swap on gfx950 - so the inline form writes out the same two (round 5 had `s_nop 0`, one
// short: ADVICE r5).  The CPU emulator of tests/emu runs swap_halves_sum below: the same two-copy exchange, lane for lane, on the emulator's
// shuffles - so the CPU suite exercises the pairing the instruction performs; tests/test_gpu_lane_sums.py pins the instruction itself.
#if defined(__HIPCC__)
__device__ __forceinline__ float xor32_sum(float v) {
    int a = __builtin_bit_cast(int, v), b = a;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xor16_sum(float v) {
    int a = __builtin_bit_cast(int, v), b = a;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
#else
// v_permlane{32,16}_swap_b32 a, b on the emulator: groups of W lanes; the ODD groups of a <-> the EVEN groups of b (W = 32: a's upper half <->
// b's lower half).  Lane l of an even group: a' = a[l], b' = a[l + W] (taken from a's odd group);  lane l of an odd group: a' = b[l - W], b' = b[l].
template <int W>
__device__ __forceinline__ float swap_halves_sum(float v) {
    const float a = v, b = v;
    const int lane = (int)(threadIdx.x & 63u);
    const bool odd = (lane / W) & 1;
    const float a_from_b = __shfl_xor(b, W);   // what an odd-group lane of a receives: b of the even group below it
    const float b_from_a = __shfl_xor(a, W);   // what an even-group lane of b receives: a of the odd group above it
    const float a2 = odd ? a_from_b : a;
    const float b2 = odd ? b : b_from_a;
    return a2 + b2;
}
__device__ __forceinline__ float xor32_sum(float v) { return swap_halves_sum<32>(v); }
__device__ __forceinline__ float xor16_sum(float v) { return swap_halves_sum<16>(v); }
#endif

// row of the 32x32 MFMA accumulator held in register r of a lane in half h (lane>>5); column = lane&31
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// torch.optim.Adam, single-tensor form (torch/optim/adam.py _single_tensor_adam, what utils/train_utils.py:9-10 builds):
//   exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2);
//   denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2^k)).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr / (1 - beta1^k))
// The scalars are Python floats (doubles) there and reach the fp32 kernels as (float)(double expression): omb1 = (float)(1 - 0.9),
// omb2 = (float)(1 - 0.999) - NOT 1.0f - (float)0.999, which is 1.3e-5 smaller and made every step 6e-6 too long (found by the
// windowed parity test against the reference's own optimiser state, tests/test_windowed_parity.py).
// step_size = lr / (1 - beta1^k), bc2s = sqrt(1 - beta2^k), both evaluated in double on the host.
// ADAM_ONLY: the caller has already branched on the optimiser around a whole loop of updates - inside an unrolled loop the uniform
// test below would put every update in its own basic block, and the independent sqrt -> rcp chains of a thread's 8 mask entries
// would run one after the other instead of interleaved (measured: the edge phase of the sparse resident kernel 0.76 -> 1.28 us).
// HAVE_R: the caller passes rbc2 = 1.0f / bc2s, formed ONCE per iteration (and made opaque, GNNX_OPAQUE): left to itself the compiler sinks the
// IEEE division - ten instructions - into every predicated per-edge block that uses it (the ISA of the sparse resident kernel's edge phase
// had one per owned edge and iteration); the same value, so the same results.
template <bool ADAM_ONLY = false, bool HAVE_R = false>
__device__ __forceinline__ void adam_update(float& theta, float& m, float& v, float g, float omb1, float beta2, float omb2,
                                            float eps, float step_size, float bc2s, int opt = 0, float rbc2 = 0.0f) {
    if (ADAM_ONLY || opt == 0) {   // Adam (uniform branch: the optimiser is a property of the whole job)
        m = m + (g - m) * omb1;
        v = v * beta2 + omb2 * g * g;
#ifndef GNNX_IEEE_MATH
        theta = theta - step_size * (m * rcp_(sqrt_(v) * (HAVE_R ? rbc2 : 1.0f / bc2s) + eps));   // (1 / bc2s: uniform)
#else
        theta = theta + (-step_size * m) / (sqrtf(v) / bc2s + eps);   // addcdiv_: self + value * t1 / t2, denom = sqrt(v) / bc2s + eps
#endif
        return;
    }
    // The other optimisers utils/train_utils.py:11-16 can build (torch defaults), same per-entry state slots (m, v), the scalars of
    // iteration k from the host table: step_size = lr_k (after the scheduler), bc2s = momentum factor (SGD: 0 at the first step).
    //   opt 1  SGD(momentum=0.95): buf = g at the first step, else buf * 0.95 + g; p += -lr * buf               (torch/optim/sgd.py)
    //   opt 2  RMSprop(alpha=0.99, eps): sq = sq * alpha + (1 - alpha) g g; p += (-lr * g) / (sqrt(sq) + eps)   (rmsprop.py)
    //   opt 3  Adagrad(eps=1e-10): sum += g g; p += (-lr * g) / (sqrt(sum) + eps)                               (adagrad.py)
    // (hardware reciprocal / square root like the Adam path: after inlining the compiler may evaluate both sides of this uniform
    // branch and select - with IEEE divisions here the edge phase of the sparse resident kernel took 1.28 instead of 0.76 us)
    if (opt == 1) {
        m = m * bc2s + g;
        theta = theta + (-step_size) * m;
    } else {
        v = (opt == 2) ? v * beta2 + omb2 * g * g : v + g * g;
        theta = theta + (-step_size * g) * rcp_(sqrt_(v) + eps);
    }
}

enum ConvMode { FWD1 = 0, FWD2 = 1, FWD3 = 2, BWD3 = 3, BWD2 = 4, BWD1 = 5 };

// row-local backward of one GraphConv layer for the 8 lanes that share a row:
// du -> dY = (dU - U (dU.U)) / r, staged in zs, then dZ[k] = sum_c dY[c] W[k][c]
__device__ __forceinline__ void rowlocal_backward(const float (&du)[4], const float (&u)[4], float rnorm, int dout,
                                                  int row, int cg, float* zs, const float* wl, float (&dz)[4]) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s = fmaf(du[j], u[j], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    __syncthreads();  // previous readers of zs are done
#pragma unroll
    for (int j = 0; j < 4; ++j) zs[row * 33 + cg + j] = (du[j] - u[j] * s) / rnorm;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) dz[j] = 0.0f;
#pragma unroll 4
    for (int c = 0; c < dout; ++c) {
        const float dy = zs[row * 33 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) dz[j] = fmaf(dy, wl[(cg + j) * 33 + c], dz[j]);
    }
}

constexpr float BN_EPS = 1e-5f;
// sum over the 8 lanes (4 columns each) that share a row in the epilogues
__device__ __forceinline__ float row8_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}
// forward of apply_bn for one row: a[j] (columns cg + j < dout, others 0) -> x_hat[j]; returns 1 / std
__device__ __forceinline__ float bn_forward_row(float (&a)[4], int cg, int dout) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (cg + j < dout) ? a[j] : 0.0f;
    const float mu = row8_sum(s) / (float)dout;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = (cg + j < dout) ? a[j] - mu : 0.0f;
        q = fmaf(a[j], a[j], q);
    }
    const float rs = 1.0f / sqrtf(row8_sum(q) / (float)dout + BN_EPS);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] *= rs;
    return rs;
}
// backward of apply_bn for one row: dx[j] = dL/dx_hat -> dL/da = rs (dx - mean(dx) - x_hat mean(dx x_hat))
__device__ __forceinline__ void bn_backward_row(float (&dx)[4], const float (&xh)[4], float rs, int cg, int dout) {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = (cg + j < dout) ? dx[j] : 0.0f;
        s1 += d;
        s2 = fmaf(d, xh[j], s2);
    }
    const float m1 = row8_sum(s1) / (float)dout, m2 = row8_sum(s2) / (float)dout;
#pragma unroll
    for (int j = 0; j < 4; ++j) dx[j] = (cg + j < dout) ? rs * (dx[j] - m1 - xh[j] * m2) : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Masked-adjacency contraction + fused row-local epilogue: one workgroup (4 waves) per 32-row block; the K
// range (all ld columns of Abar) is split over the 4 waves and reduced through LDS.  (A 128-row / 16-B-per-lane
// variant was measured in round 1 and was slower - 175 vs 152 us per launch on the BA-House x100k sample - because
// its per-wave MFMA chains are 4x longer while the prefetch depth stays the same; see DESIGN.md.)
// ---------------------------------------------------------------------------------------------
struct ConvShared {
    float red[4 * TILE * 33];  // split-K partial tiles
    float wl[32 * 33];         // layer weight, padded rows
    float zs[TILE * 33];       // reduced Z rows, then dY staging
    float phis[32];
    float part[4 * 32];  // per-wave partials of the row-set reductions (red holds the tiles of the other row blocks)
};

// Fused row-local epilogue for 32 rows: thread (row = tid/8, cg = 4 (tid%8)) holds z4 = the reduced contraction of
// target row `irow`.  `slot` = index of this 32-row set among the target's row sets (for the per-set partials).
template <int MODE>
__device__ __forceinline__ void conv_epilogue(const Params& p, const ConvTile& tl, const TargetMeta& tm, ConvShared& sh,
                                              float (&z4)[4], int irow, int slot, const f32x4& pre_a, const f32x4& pre_b,
                                              const f32x4& pre_c, const int (&pre_ar)[4], float pre_rn, const f32x4& pre_x, float pre_rs) {
    constexpr int layer = (MODE == FWD1 || MODE == BWD1) ? 0 : (MODE == FWD2 || MODE == BWD2) ? 1 : 2;
    constexpr bool IS_FWD = (MODE == FWD1 || MODE == FWD2 || MODE == FWD3);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row = tid >> 3, cg = (tid & 7) * 4;
    const int ld = tm.ld;
    const size_t grow = (size_t)tm.offR + irow;
    if (IS_FWD) {
        if (MODE == FWD1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p.Zraw[grow * FS + cg + j] = z4[j];
                z4[j] *= sh.phis[cg + j];  // Abar.(X * phi) == (Abar.X) * phi, phi is a per-column scale
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sh.zs[row * 33 + cg + j] = z4[j];
        __syncthreads();
        const int din = (MODE == FWD1) ? p.D : p.H;
        const int dout = (MODE == FWD3) ? p.O : p.H;
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = 0.0f;
#pragma unroll 4
        for (int k = 0; k < din; ++k) {
            const float z = sh.zs[row * 33 + k];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = fmaf(z, sh.wl[k * 33 + cg + j], y[j]);
        }
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = (cg + j < dout) ? y[j] + pre_a[j] : 0.0f;
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
        float xh[4];  // the layer's output activation: relu(U), standardised per row with --bn
#pragma unroll
        for (int j = 0; j < 4; ++j) xh[j] = fmaxf(y[j] / rnorm, 0.0f);
        if (MODE != FWD3 && p.bn) {
            const float rs = bn_forward_row(xh, cg, dout);
            float* Xn = p.Xn[layer] + grow * FS;
            float* XnT = p.XnT[layer] + tm.offR * FS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Xn[cg + j] = xh[j];
                XnT[(size_t)(cg + j) * ld + irow] = xh[j];
            }
            if ((tid & 7) == 0) p.bnr[layer][grow] = rs;
        }
        if (MODE == FWD2 && !p.graph_mode) {
            // node mode reads only row t of layer 3 (explain.py:713): this row set's share of
            // Z3[t] = sum_k Abar[t][k] X2[k] is reduced here so the head never walks all n rows
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[j] = pre_rn * xh[j];
                part[j] += __shfl_xor(part[j], 8);
                part[j] += __shfl_xor(part[j], 16);
                part[j] += __shfl_xor(part[j], 32);
            }
            __syncthreads();
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sh.part[wave * 32 + cg + j] = part[j];
            }
            __syncthreads();
            if (tid < 32)
                p.z3p[((size_t)(tm.offR >> 5) + slot) * FS + tid] =
                    sh.part[tid] + sh.part[32 + tid] + sh.part[64 + tid] + sh.part[96 + tid];
        }
    } else {
        // BWD3 / BWD2 / BWD1: dX (+ direct part) -> dZ_layer
        const int dout = (layer == 2) ? p.O : p.H;  // width of U[layer] / dX
        const int din = (layer == 0) ? p.D : p.H;   // width of dZ[layer]
        float du[4], u[4], dz[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = cg + j;
            float dx = (MODE == BWD3) ? 0.0f : z4[j];
            u[j] = pre_a[j];
            if (c < dout && pre_ar[j] == irow) dx += pre_b[j];
            du[j] = (c < dout) ? dx : 0.0f;
        }
        if (layer < 2) {
            if (p.bn) {  // through the row-wise standardisation first (dx is the gradient w.r.t. the standardised activation)
                const float xh4[4] = {pre_x[0], pre_x[1], pre_x[2], pre_x[3]};
                bn_backward_row(du, xh4, pre_rs, cg, dout);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) du[j] = (u[j] > 0.0f) ? du[j] : 0.0f;
        }
        rowlocal_backward(du, u, pre_rn, dout, row, cg, sh.zs, sh.wl, dz);
        float* dZ = p.dZ[layer] + grow * FS;
        float* dZT = p.dZT[layer] + tm.offR * FS;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dz[j] = (cg + j < din) ? dz[j] : 0.0f;
            dZ[cg + j] = dz[j];
            dZT[(size_t)(cg + j) * ld + irow] = dz[j];
        }
        if (MODE == BWD1) {
            // feature-mask gradient: colsum((Abar.dZ1) * X) == colsum(dZ1 * (Abar.X)), Abar symmetric
            float part[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                part[j] = dz[j] * pre_c[j];
                part[j] += __shfl_xor(part[j], 8);  // the 8 rows of this wave
                part[j] += __shfl_xor(part[j], 16);
                part[j] += __shfl_xor(part[j], 32);
            }
            __syncthreads();
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sh.part[wave * 32 + cg + j] = part[j];
            }
            __syncthreads();
            if (tid < 32)  // one partial per row set, summed in a fixed order by k_mask (deterministic)
                p.df[((size_t)(tm.offR >> 5) + slot) * FS + tid] =
                    sh.part[tid] + sh.part[32 + tid] + sh.part[64 + tid] + sh.part[96 + tid];
        }
    }
}

// operands of the epilogue for target row irow, columns cg..cg+3
template <int MODE>
__device__ __forceinline__ void conv_epilogue_operands(const Params& p, const ConvTile& tl, const TargetMeta& tm, int irow,
                                                       int cg, f32x4& pre_a, f32x4& pre_b, f32x4& pre_c, int (&pre_ar)[4],
                                                       float& pre_rn, f32x4& pre_x, float& pre_rs) {
    constexpr int layer = (MODE == FWD1 || MODE == BWD1) ? 0 : (MODE == FWD2 || MODE == BWD2) ? 1 : 2;
    constexpr bool IS_FWD = (MODE == FWD1 || MODE == FWD2 || MODE == FWD3);
    const size_t grow = (size_t)tm.offR + irow;
    if (IS_FWD) {
        pre_a = *reinterpret_cast<const f32x4*>(p.wts + WT_B + layer * 32 + cg);  // bias
        if (MODE == FWD2 && !p.graph_mode) pre_rn = p.Abar[tm.offQ + (size_t)tm.t * tm.ld + irow];  // Abar[t][row]
    } else {
        pre_a = *reinterpret_cast<const f32x4*>(p.U[layer] + grow * FS + cg);          // U
        pre_b = *reinterpret_cast<const f32x4*>(p.dE + (tl.t * 3 + layer) * FS + cg);  // direct gradient
#pragma unroll
        for (int j = 0; j < 4; ++j) pre_ar[j] = p.argrow[(tl.t * 3 + layer) * FS + cg + j];
        pre_rn = p.rn[layer][grow];
        if (MODE == BWD1) pre_c = *reinterpret_cast<const f32x4*>(p.Zraw + grow * FS + cg);
        if (layer < 2 && p.bn) {
            pre_x = *reinterpret_cast<const f32x4*>(p.Xn[layer] + grow * FS + cg);
            pre_rs = p.bnr[layer][grow];
        }
    }
}

template <int MODE>
__device__ __forceinline__ void conv_stage_weights(const Params& p, const ConvTile& tl, ConvShared& sh, int iter) {
    constexpr int layer = (MODE == FWD1 || MODE == BWD1) ? 0 : (MODE == FWD2 || MODE == BWD2) ? 1 : 2;
    const int tid = threadIdx.x;
    const float* W = p.wts + WT_W + layer * 1024;
    for (int e = tid; e < 1024; e += 256) sh.wl[(e >> 5) * 33 + (e & 31)] = W[e];
    if (MODE == FWD1 && tid < 32) sh.phis[tid] = (tid < p.D) ? sigmoidf_(p.f[iter & 1][tl.t * FS + tid]) : 0.0f;
}

template <int MODE>
__device__ __forceinline__ const float* conv_b_source(const Params& p) {
    if (p.bn && (MODE == FWD2 || MODE == FWD3)) return p.Xn[MODE == FWD2 ? 0 : 1];  // standardised activations, no ReLU on top
    return (MODE == FWD1) ? p.X : (MODE == FWD2) ? p.U[0] : (MODE == FWD3) ? p.U[1] : (MODE == BWD2) ? p.dZ[2] : p.dZ[1];
}

template <int MODE>
__global__ __launch_bounds__(256) void k_conv(Params p, const ConvUnit* units, int iter) {
    __shared__ ConvShared sh;
    const ConvUnit cu = units[blockIdx.x];
    if (MODE == BWD3 && cu.ks) return;  // no contraction in BWD3: the first slice does the row blocks
    const TargetMeta tm = cu.tm;
    const int ld = tm.ld;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    // nrb in {1, 2, 4} adjacent row blocks per workgroup, the K range cut into 4 / nrb slices: wave w contracts row block
    // rb + w % nrb over slice w / nrb.  Waves of one slice read ADJACENT 128-B segments of the same rows of Abar (512 B
    // contiguous for nrb = 4) and the same rows of the B operand.
    const int nrb = cu.nrb, ksplit = 4 / nrb;
    const int wj = wave % nrb, ws = wave / nrb;
    const int row0 = (cu.rb + wj) * TILE;
    const ConvTile tl = {cu.t, cu.rb, cu.tm};
    conv_stage_weights<MODE>(p, tl, sh, iter);
    const int row = tid >> 3, cg = (tid & 7) * 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (MODE != BWD3) {
        const float* Bsrc = conv_b_source<MODE>(p) + tm.offR * FS + li;
        const bool relu_b = (MODE == FWD2 || MODE == FWD3) && !p.bn;
        const int kchunk = (cu.ke - cu.kb) / ksplit;  // per wave; a multiple of 8
        if (ld < 256) {
            // small targets (latency-bound; nrb = 1): lane (i = li, half h) reads 16 B of row row0+i,
            // Abar[row0+i][k0 + 8u + 4h .. +3].  The MFMA k index is a free permutation as long as A and B agree,
            // so step e of a batch uses k = k0 + 8u + 4h + e: one 16-B load feeds 4 MFMAs.  Measured on syn1:
            // 9.5-10.8 us per launch vs 11.0-11.5 us for the 4-B form below.
            const float* Ab = p.Abar + tm.offQ + (size_t)(row0 + li) * ld + 4 * h;
            const int k0 = cu.kb + ws * kchunk;
            constexpr int NB = 4;  // batches of 8 k values in flight per wave
            f32x4 a[NB];
            float b[NB][4];
            auto load = [&](int slot, int kk) {
                a[slot] = *reinterpret_cast<const f32x4*>(Ab + k0 + kk);
#pragma unroll
                for (int e = 0; e < 4; ++e) b[slot][e] = Bsrc[(size_t)(k0 + kk + 4 * h + e) * FS];
            };
#pragma unroll
            for (int sl = 0; sl < NB; ++sl)
                if (8 * sl < kchunk) load(sl, 8 * sl);
            for (int kk = 0; kk < kchunk; kk += 8 * NB) {
#pragma unroll
                for (int sl = 0; sl < NB; ++sl) {
                    if (kk + 8 * sl < kchunk) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float bb = b[sl][e];
                            if (relu_b) bb = fmaxf(bb, 0.0f);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl][e], bb, acc, 0, 0, 0);
                        }
                        if (kk + 8 * (sl + NB) < kchunk) load(sl, kk + 8 * (sl + NB));
                    }
                }
            }
        } else {
            // large targets (bandwidth-bound): Abar is symmetric, so the A operand is read as Abar[k][row0 + li]:
            // 128-B fully coalesced segments straight into the MFMA A layout.  The row form above touches 32
            // cache lines per wave instruction and was 12 % slower here (BA-House x100k: 152-162 vs 171-181 us).
            const float* Ab = p.Abar + tm.offQ + row0 + li;
            const int k0 = cu.kb + ws * kchunk + h;
            // A ring of RING k pairs in flight per wave: step j feeds pair j to the MFMA and at once re-issues that slot's
            // loads for pair j + RING, so the loads stay RING deep WHILE the MFMAs run.  (Round 2 ran batches - 16 MFMAs, then
            // the 32 loads of the batch after next - with predicated loads that compiled to a branch and a wait per load.)
            // Straight-line code: a load past the end of the slice reads the slice's last pair again (an L1 hit) and the
            // select that zeroes it sits at the USE.  Measured on the BA-House x100k streaming set (tools/probe_conv.py,
            // one box): batches 119 us per launch; ring 32 (4 waves per SIMD) 90; rings 8 / 12 / 16 / 24 (5 waves) 82-85.
            // tools/micro/stream_pattern.hip puts the ceiling of "one 32x32x2 MFMA + one B row pair per A row pair" at
            // 3.6 TB/s of Abar against 6.4 for the bare stream - the MFMA step and the B loads cost 2.3 and 1.4 TB/s.
#ifndef GNNX_CONV_RING  // (measurement knob, tools/probe_conv.py)
#define GNNX_CONV_RING 16
#endif
            constexpr int RING = GNNX_CONV_RING;
            float ra[RING], rb[RING];
            const int klast = kchunk - 2;
            auto fetch = [&](int slot, int ks) {  // ks = k offset of the pair inside the slice (even); kchunk is a multiple of 8
                const int k = k0 + (ks < klast ? ks : klast);
                ra[slot] = Ab[(size_t)k * ld];
                rb[slot] = Bsrc[(size_t)k * FS];
            };
#pragma unroll
            for (int u = 0; u < RING; ++u) fetch(u, 2 * u);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the loads next to their uses: 3-5 in flight)
            for (int s0 = 0; s0 < kchunk; s0 += 2 * RING) {
#pragma unroll
                for (int u = 0; u < RING; ++u) {
                    const bool on = s0 + 2 * u < kchunk;  // (the select sits at the USE: next to the load it would wait for it)
                    float bb = on ? rb[u] : 0.0f;
                    if (relu_b) bb = fmaxf(bb, 0.0f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(on ? ra[u] : 0.0f, bb, acc, 0, 0, 0);
                    fetch(u, s0 + 2 * RING + 2 * u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
    // the wave tiles meet in LDS: tile w = row block w % nrb, K slice w / nrb
#pragma unroll
    for (int r = 0; r < 16; ++r) sh.red[(wave * TILE + acc_row(r, h)) * 33 + li] = acc[r];
    for (int j = 0; j < nrb; ++j) {
        __syncthreads();  // tiles complete / the previous row block's epilogue is done with zs
        const int irow = (cu.rb + j) * TILE + row;
        float z4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float s = 0.0f;
            for (int w = j; w < 4; w += nrb) s += sh.red[(w * TILE + row) * 33 + cg + q];
            z4[q] = s;
        }
        if (MODE != BWD3 && cu.nks > 1) {  // one K slice of several: the partial tile goes to its slab
            *reinterpret_cast<f32x4*>(p.cpart + ((size_t)cu.part + j * cu.nks + cu.ks) * (TILE * FS) + row * FS + cg) =
                f32x4{z4[0], z4[1], z4[2], z4[3]};
            continue;
        }
        f32x4 pre_a = {0.0f, 0.0f, 0.0f, 0.0f}, pre_b = pre_a, pre_c = pre_a;
        int pre_ar[4] = {-1, -1, -1, -1};
        float pre_rn = 1.0f, pre_rs = 1.0f;
        f32x4 pre_x = pre_a;
        conv_epilogue_operands<MODE>(p, tl, tm, irow, cg, pre_a, pre_b, pre_c, pre_ar, pre_rn, pre_x, pre_rs);
        conv_epilogue<MODE>(p, tl, tm, sh, z4, irow, cu.rb + j, pre_a, pre_b, pre_c, pre_ar, pre_rn, pre_x, pre_rs);
    }
}

// The row blocks whose K range was cut: sum the slabs in slice order (deterministic), then the same epilogue.
template <int MODE>
__global__ __launch_bounds__(256) void k_conv_reduce(Params p, const ConvJoin* joins, int iter) {
    __shared__ ConvShared sh;
    const ConvJoin jn = joins[blockIdx.x];
    const TargetMeta tm = jn.tm;
    const ConvTile tl = {jn.t, jn.rb, jn.tm};
    const int tid = threadIdx.x;
    const int row = tid >> 3, cg = (tid & 7) * 4;
    const int irow = jn.rb * TILE + row;
    conv_stage_weights<MODE>(p, tl, sh, iter);
    f32x4 pre_a = {0.0f, 0.0f, 0.0f, 0.0f}, pre_b = pre_a, pre_c = pre_a;
    int pre_ar[4] = {-1, -1, -1, -1};
    float pre_rn = 1.0f, pre_rs = 1.0f;
    f32x4 pre_x = pre_a;
    conv_epilogue_operands<MODE>(p, tl, tm, irow, cg, pre_a, pre_b, pre_c, pre_ar, pre_rn, pre_x, pre_rs);
    const f32x4* slab = reinterpret_cast<const f32x4*>(p.cpart + (size_t)jn.part * (TILE * FS) + row * FS + cg);
    f32x4 acc = slab[0];
    for (int q = 1; q < jn.nks; ++q) {
        const f32x4 v = slab[(size_t)q * (TILE * FS / 4)];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
    }
    float z4[4] = {acc[0], acc[1], acc[2], acc[3]};
    __syncthreads();  // weights staged
    conv_epilogue<MODE>(p, tl, tm, sh, z4, irow, jn.rb, pre_a, pre_b, pre_c, pre_ar, pre_rn, pre_x, pre_rs);
}

// softmax head shared by both modes: e[96] (concatenated embedding) -> probs, g = p - onehot, dE = Wp^T g.
// Must be called by all 256 threads of the workgroup.
// sWp: the prediction head (C rows of 96 + bias) staged in LDS by stage_head_weights().
__device__ __forceinline__ void stage_head_weights(const Params& p, float* sWp) {
    for (int e = threadIdx.x; e < p.C * 96; e += blockDim.x) sWp[e] = p.wts[WT_WP + e];
    if (threadIdx.x < CMAX) sWp[CMAX * 96 + threadIdx.x] = p.wts[WT_BP + threadIdx.x];
}

__device__ __forceinline__ void head_softmax(const Params& p, const TargetMeta& tm, int t, int iter, bool write,
                                             const float* sWp, const float* e, float* g, float* dEs) {
    const int tid = threadIdx.x;
    const float* Wp = sWp;
    if (tid < 64) {  // wave 0: logits, softmax
        float z = -3.0e38f;
        if (tid < p.C) {
            float s = 0.0f;
            for (int q = 0; q < 96; ++q) s = fmaf(Wp[tid * 96 + q], e[q], s);
            z = s + sWp[CMAX * 96 + tid];
        }
        float mx = z;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float ex = (tid < p.C) ? expf(z - mx) : 0.0f;
        float sum = ex;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
        const float pr = ex / sum;
        if (tid < CMAX) {
            g[tid] = (tid < p.C) ? pr - ((tid == tm.y_gt) ? 1.0f : 0.0f) : 0.0f;
            if (write) p.probs[t * CMAX + tid] = (tid < p.C) ? pr : 0.0f;
        }
        if (write && p.loss && tid == tm.y_gt) p.loss[((size_t)t * p.num_iters + iter) * NLOSS + 0] = -logf(pr);
        if (write && p.loss && tid < p.C && tid < LOGPN) p.loss[((size_t)t * p.num_iters + iter) * NLOSS + LOGP + tid] = pr;   // explain.py:710-714, 157-158
    }
    __syncthreads();
    if (tid < 96) {
        float s = 0.0f;
        for (int c = 0; c < p.C; ++c) s = fmaf(Wp[c * 96 + tid], g[c], s);
        dEs[tid] = s;
    }
    if (write && p.loss && tid == 128) {
        float s = 0.0f;
        for (int d = 0; d < p.D; ++d) s += sigmoidf_(p.f[iter & 1][t * FS + d]);
        p.loss[((size_t)t * p.num_iters + iter) * NLOSS + 4] = p.c_feat_size * s / (float)p.D;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Graph-mode head: one workgroup per graph.  Column-wise max over the n rows of every layer.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head(Params p, int iter) {
    __shared__ float e[96];
    __shared__ int erow[96];
    __shared__ float g[CMAX];
    __shared__ float dEs[96];
    __shared__ float sWp[CMAX * 96 + CMAX];
    const int t = blockIdx.x;
    const TargetMeta tm = p.meta[t];
    const int tid = threadIdx.x;
    const int dims[3] = {p.H, p.H, p.O};
    stage_head_weights(p, sWp);
    // padded rows of the reference are rows < n here; the engine's own padding rows n..ld-1 are excluded
    const int wave = tid >> 6, lane = tid & 63;
    for (int col = wave; col < 96; col += 4) {
        const int l = col >> 5, c = col & 31;
        float best = -3.0e38f;
        int barg = 0;
        if (c < dims[l]) {
            const bool std_act = p.bn && l < 2;   // --bn: the pooled activation is the standardised one
            const float* UT = (std_act ? p.XnT[l] : p.UT[l]) + tm.offR * FS + (size_t)c * tm.ld;
            for (int i = lane; i < tm.n; i += 64) {
                float v = UT[i];
                if (l < 2 && !std_act) v = fmaxf(v, 0.0f);
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
    __syncthreads();
    head_softmax(p, tm, t, iter, true, sWp, e, g, dEs);
    if (tid < 96) {
        p.dE[t * 96 + tid] = dEs[tid];
        p.argrow[t * 96 + tid] = erow[tid];
    }
}

// ---------------------------------------------------------------------------------------------
// Node-mode head: one workgroup per 32-row block of a target.  Only row t of the last layer is consumed
// by the reference (explain.py:713), so every workgroup (redundantly, it is a length-n mat-vec) computes
// that row with Abar[t,:], the head, dE and dZ3[t]; then - because Abar.dZ3 is rank-1 - its own 32 rows
// of dZ2 and of g3, the layer-3 row of G.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_node_head(Params p, const ConvTile* tiles, int iter) {
    __shared__ float part[8 * 32];
    __shared__ float e[96], g[CMAX], dEs[96];
    __shared__ float y3[32], dz3[32];
    __shared__ float wl[32 * 33], zs[TILE * 33];
    __shared__ float sWp[CMAX * 96 + CMAX];
    __shared__ float sr3;
    const ConvTile tl = tiles[blockIdx.x];
    const int t = tl.t;
    const TargetMeta tm = tl.tm;
    const int tid = threadIdx.x, ld = tm.ld, n = tm.n, tr = tm.t;
    const float* Ab = p.Abar + tm.offQ;
    const float* U1 = p.U[0] + tm.offR * FS;
    const float* U2 = p.U[1] + tm.offR * FS;
    const float* W3 = p.wts + WT_W + 2 * 1024;
    const float* W2 = p.wts + WT_W + 1 * 1024;
    stage_head_weights(p, sWp);
    // operands of the later phases, loaded up-front (their latency hides under the mat-vec below)
    const int row = tid >> 3, cg = (tid & 7) * 4;
    const int i = tl.rb * TILE + row;
    // row t of the two hidden activations: relu(U), or the standardised rows with --bn (no ReLU on top)
    const float* X1a = p.bn ? p.Xn[0] + tm.offR * FS : U1;
    const float* X2a = p.bn ? p.Xn[1] + tm.offR * FS : U2;
    const float u1t = (tid < 32) ? X1a[(size_t)tr * FS + tid] : 0.0f;
    const float u2t = (tid < 32) ? X2a[(size_t)tr * FS + tid] : 0.0f;
    const float b3 = (tid < 32) ? p.wts[WT_B + 2 * 32 + tid] : 0.0f;
    const float ait = Ab[(size_t)tr * ld + i];  // Abar[t][i] == Abar[i][t]
    const f32x4 u2i = *reinterpret_cast<const f32x4*>(U2 + (size_t)i * FS + cg);
    const float rn2i = p.rn[1][tm.offR + i];
    f32x4 x2i = u2i;      // --bn: standardised row i of layer 2 and its 1 / std
    float rs2i = 1.0f;
    if (p.bn) {
        x2i = *reinterpret_cast<const f32x4*>(p.Xn[1] + (tm.offR + i) * FS + cg);
        rs2i = p.bnr[1][tm.offR + i];
    }

    // Z3[t][c] = sum_k Abar[t][k] relu(U2[k][c]): the per-row-block partials were reduced by k_conv<FWD2>;
    // sum them in a fixed order (8 slices x 32 columns)
    {
        const int c = tid & 31, sl = tid >> 5;
        float s = 0.0f;
        for (int rb = sl; rb < (ld >> 5); rb += 8) s += p.z3p[((size_t)(tm.offR >> 5) + rb) * FS + c];
        part[sl * 32 + c] = s;
        for (int e2 = tid; e2 < 1024; e2 += 256) wl[(e2 >> 5) * 33 + (e2 & 31)] = W3[e2];
    }
    __syncthreads();
    if (tid < 32) {
        float z = 0.0f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) z += part[sl * 32 + tid];
        zs[tid] = z;
    }
    __syncthreads();
    if (tid < 64) {  // Y3 = Z3 W3 + b3, normalise (wave 0, columns in lanes 0..31)
        const int c = tid & 31;
        const float b3c = __shfl(b3, c);  // all 64 lanes take part in the shuffle
        float y = 0.0f;
        if (c < p.O) {
#pragma unroll 4
            for (int k = 0; k < p.H; ++k) y = fmaf(zs[k], wl[k * 33 + c], y);
            y += b3c;
        }
        float ss = (tid < 32) ? y * y : 0.0f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
        const float rnorm = fmaxf(sqrtf(ss), 1e-12f);
        if (tid < 32) {
            const float u = y / rnorm;
            y3[tid] = u;
            e[64 + tid] = u;
            e[tid] = (tid < p.H) ? (p.bn ? u1t : fmaxf(u1t, 0.0f)) : 0.0f;
            e[32 + tid] = (tid < p.H) ? (p.bn ? u2t : fmaxf(u2t, 0.0f)) : 0.0f;
        }
        if (tid == 0) sr3 = rnorm;
    }
    __syncthreads();
    head_softmax(p, tm, t, iter, tl.rb == 0, sWp, e, g, dEs);
    // dZ3[t] : backward through the last layer's normalisation and W3 (row t only)
    if (tid < 64) {
        const int c = tid & 31;
        const float du = (tid < 32 && c < p.O) ? dEs[64 + c] : 0.0f;
        const float u = (tid < 32) ? y3[c] : 0.0f;
        float s = du * u;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        if (tid < 32) zs[c] = (du - u * s) / sr3;  // dY3[t]
    }
    __syncthreads();
    if (tid < 32) {
        float v = 0.0f;
        if (tid < p.H)
#pragma unroll 4
            for (int c = 0; c < p.O; ++c) v = fmaf(zs[c], wl[tid * 33 + c], v);
        dz3[tid] = v;
    }
    __syncthreads();
    for (int e2 = tid; e2 < 1024; e2 += 256) wl[(e2 >> 5) * 33 + (e2 & 31)] = W2[e2];
    // this block's 32 rows: dX2[i] = Abar[i][t] dZ3[t] (+ dE2 on row t), row-local backward of layer 2 -> dZ2;
    // g3[i] = dZ3[t] . relu(U2[i]) (layer-3 part of row t of dL/dAbar)
    float du[4], u[4], dz[4];
    float gpart = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = cg + j;
        u[j] = u2i[j];
        gpart = fmaf(dz3[c], p.bn ? x2i[j] : fmaxf(u[j], 0.0f), gpart);  // dz3[c] == 0 for c >= H
        float dx = ait * dz3[c];
        if (i == tr) dx += dEs[32 + c];
        du[j] = (c < p.H) ? dx : 0.0f;
    }
    if (p.bn) {
        const float xh4[4] = {x2i[0], x2i[1], x2i[2], x2i[3]};
        bn_backward_row(du, xh4, rs2i, cg, p.H);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) du[j] = (u[j] > 0.0f) ? du[j] : 0.0f;
    gpart += __shfl_xor(gpart, 1);
    gpart += __shfl_xor(gpart, 2);
    gpart += __shfl_xor(gpart, 4);
    if ((tid & 7) == 0) p.g3[tm.offR + i] = (i < n) ? gpart : 0.0f;
    rowlocal_backward(du, u, rn2i, p.H, row, cg, zs, wl, dz);
    float* dZ = p.dZ[1] + tm.offR * FS;
    float* dZT = p.dZT[1] + tm.offR * FS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = (cg + j < p.H) ? dz[j] : 0.0f;
        dZ[(size_t)i * FS + cg + j] = v;
        dZT[(size_t)(cg + j) * ld + i] = v;
    }
    if (tl.rb == 0 && tid < 96) {
        p.dE[t * 96 + tid] = dEs[tid];
        p.argrow[t * 96 + tid] = tr;
    }
}

// mask activation (explain.py:667-670), its derivative and d(entropy)/dS of explain.py:769 for both settings of mask_act
template <bool RELU>
__device__ __forceinline__ float mask_act(float m) { return RELU ? ((m <= 0.0f) ? 0.0f : m) : sigmoidf_(m); }  // NaN stays NaN, as torch.relu
template <bool RELU>
__device__ __forceinline__ float mask_dact(float m, float s) { return RELU ? ((m > 0.0f) ? 1.0f : 0.0f) : s * (1.0f - s); }
// sigmoid: log(1 - S) - log(S) == -M exactly; ReLU: the logs themselves (NaN for S > 1, -inf * 0 for S == 0 - as torch)
template <bool RELU>
__device__ __forceinline__ float mask_dent(float m, float s) { return RELU ? (logf(1.0f - s) - logf(s)) : -m; }

// ---------------------------------------------------------------------------------------------
// Fused mask kernel: one workgroup (4 waves) per tile pair {(I,J),(J,I)}, I <= J.
//   * thread (i = tid/8, c4 = 4 (tid%8)) owns the 4 entries (i, c4..c4+3) of tile (I,J) AND their mirror
//     entries (c4.., i) of tile (J,I); every mask entry is updated by exactly one thread (diagonal tiles
//     exchange sigma through LDS) => bitwise symmetric, batch-invariant, deterministic output;
//   * every global access to the n x n state (M, m, v, A, Abar) is 16 B per lane; the mirror tile is loaded
//     row-wise and transposed through LDS;
//   * the K = D + 2H product of the G tile (dL/dAbar, never materialised) is split over the 4 waves on
//     v_mfma_f32_32x32x2_f32, the 4 partial tiles are summed through LDS in a fixed order;
//   * all global loads are issued before any dependent work (one DRAM round trip per workgroup);
//   * padding entries (>= n) are ordinary non-edges (A = 0 there): they are updated like any other entry,
//     which keeps the inner loop free of predication; they never influence a real entry and are not returned.
//   UPDATE=false : only Abar = A * sym(sigma(M)) (initial forward)
//   UPDATE=true  : gradient + Adam step, then (WRITE_ABAR) the next Abar
//   NODE         : layer-3 part of G is the rank-2 term built from g3 (node mode)
//   LOSS         : also accumulate the size / entropy / Laplacian loss terms (logging)
// ---------------------------------------------------------------------------------------------
//   RELU         : mask_act == "ReLU" (explain.py:669-670, 757-760): relu(M) instead of sigmoid(M) in the masked adjacency, the
//                  size term and the entropy term - whose log(1 - relu(M)) is NaN for every entry > 1, exactly as in the
//                  reference (its loss is NaN from the first epoch on a N(1, .) initialised mask)
template <bool UPDATE, bool WRITE_ABAR, bool NODE, bool LOSS, bool RELU = false>
__global__ __launch_bounds__(256, 4) void k_mask(Params p, const MaskTile* tiles, int iter, float step_size, float bc2s) {
    constexpr int LS = 33;  // LDS row stride
    __shared__ float sGp[4 * TILE * LS];                             // per-wave partial G tiles, [w][i][j]
    __shared__ float sPM[TILE * LS], sPm[TILE * LS], sPv[TILE * LS];  // mirror tile (J,I), natural orientation [j][i]
    __shared__ float sS[TILE * LS];                                  // sigma exchange, then Abar of the mirror tile
    const MaskTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int ld = tm.ld, n = tm.n;
    const int I0 = tl.I * TILE, J0 = tl.J * TILE;
    const bool diag = (tl.I == tl.J);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int i = tid >> 3, c4 = (tid & 7) * 4;  // row-major layout: row i, columns c4 .. c4+3
    const int gi = I0 + i;
    const size_t own = tm.offQ + (size_t)gi * ld + J0 + c4;        // my 4 entries of tile (I,J)
    const size_t par = tm.offQ + (size_t)(J0 + i) * ld + I0 + c4;  // my row segment of tile (J,I) (staging only)

    // ---- every global load is issued here, back to back, and NOTHING is computed on a loaded value before the last one is out:
    // an operation next to its load makes the compiler wait for it right there (round 2 had sigmoid(f[k]) * x and relu(x) inside
    // the load loop: s_waitcnt vmcnt(0) after every k step, ~8 serialised L2 round trips per workgroup - the kernel ran at the
    // rate of that chain, 4.1-4.6 TB/s, not of HBM).  The G operands (L2-resident rows, they return first) go out before the
    // seven HBM streams of the two tiles.
    float zi[3][4], zj[3][4], xi[3][4], xj[3][4], fk[4];
    f32x4 yj4 = {0.0f, 0.0f, 0.0f, 0.0f}, g3j4 = {0.0f, 0.0f, 0.0f, 0.0f};
    float yi = 0.0f, g3i = 0.0f;
    // G operands: k-step s (columns 2s, 2s+1) of a layer goes to wave s % 4 -> at most 4 steps per wave per layer
    constexpr int NL = NODE ? 2 : 3;
    if (UPDATE) {
        const size_t ro = (size_t)tm.offR * FS;
        const float* fcur = p.f[iter & 1] + tl.t * FS;
#pragma unroll
        for (int u = 0; u < 4; ++u) fk[u] = fcur[2 * (wave + 4 * u) + h];  // (first: whatever the compiler hoists onto them waits for nothing else)
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int d = (l == 0) ? p.D : p.H;
            const float* zT = p.dZT[l] + ro;
            const float* xT = (l == 0) ? p.XT + ro : (p.bn ? p.XnT[l - 1] : p.UT[l - 1]) + ro;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = 2 * (wave + 4 * u) + h;  // < 32; columns >= d hold zeros
                const bool on = 2 * (wave + 4 * u) < d;
                float a = 0.0f, b = 0.0f, c = 0.0f, e = 0.0f;
                if (on) {
                    a = zT[(size_t)k * ld + I0 + li];
                    b = zT[(size_t)k * ld + J0 + li];
                    c = xT[(size_t)k * ld + I0 + li];
                    e = xT[(size_t)k * ld + J0 + li];
                }
                zi[l][u] = a; zj[l][u] = b; xi[l][u] = c; xj[l][u] = e;
            }
        }
        if (!p.graph_mode) {
            yj4 = *reinterpret_cast<const f32x4*>(p.yhat + tm.offR + J0 + c4);
            yi = p.yhat[tm.offR + gi];
        }
        if (NODE) {
            g3j4 = *reinterpret_cast<const f32x4*>(p.g3 + tm.offR + J0 + c4);
            g3i = p.g3[tm.offR + gi];
        }
    }
    f32x4 Mo = *reinterpret_cast<const f32x4*>(p.M + own);
    const f32x4 Ao = *reinterpret_cast<const f32x4*>(p.A + own);
    const f32x4 Mp = *reinterpret_cast<const f32x4*>(p.M + par);
    f32x4 mo, vo, mp, vp;
    if (UPDATE) {
        mo = *reinterpret_cast<const f32x4*>(p.mM + own);
        vo = *reinterpret_cast<const f32x4*>(p.vM + own);
        if (!diag) {
            mp = *reinterpret_cast<const f32x4*>(p.mM + par);
            vp = *reinterpret_cast<const f32x4*>(p.vM + par);
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // (keeps the loads above together, ahead of everything below)
    if (UPDATE) {
        // the G tiles first: their operands are back long before the HBM streams
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int d = (l == 0) ? p.D : p.H;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (2 * (wave + 4 * u) < d) {
                    float c = xi[l][u], e = xj[l][u];
                    if (l == 0) {  // the masked features: X * sigma(feat_mask)
                        const int k = 2 * (wave + 4 * u) + h;
                        const float phi = (k < p.D) ? sigmoidf_(fk[u]) : 0.0f;
                        c *= phi;
                        e *= phi;
                    } else if (!p.bn) {   // --bn: the standardised activations are the layer inputs as they are
                        c = fmaxf(c, 0.0f);
                        e = fmaxf(e, 0.0f);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(zi[l][u], e, acc, 0, 0, 0);  // G[i][j]
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c, zj[l][u], acc, 0, 0, 0);  // G[j][i]
                }
            }
        }
        // accumulator (C layout: row acc_row(r,h), column li) -> this wave's partial tile in LDS
#pragma unroll
        for (int r = 0; r < 16; ++r) sGp[(wave * TILE + acc_row(r, h)) * LS + li] = acc[r];
    }
    // mirror tile -> LDS in its natural orientation [j][i]; read back transposed below
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sPM[i * LS + c4 + e] = Mp[e];
        if (UPDATE && !diag) { sPm[i * LS + c4 + e] = mp[e]; sPv[i * LS + c4 + e] = vp[e]; }
    }
    __syncthreads();

    const float inv_n2 = 1.0f / ((float)n * (float)n);
    const bool lapl = UPDATE && !p.graph_mode;
    float s_size = 0.0f, s_ent = 0.0f, s_lap = 0.0f;
    f32x4 Sown;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = c4 + e, gj = J0 + j;
        float Mij = Mo[e];
        if (UPDATE) {
            const float Aij = Ao[e];
            const float offd = (gi != gj) ? 1.0f : 0.0f;
            float Gsum = (sGp[(0 * TILE + i) * LS + j] + sGp[(1 * TILE + i) * LS + j]) +
                         (sGp[(2 * TILE + i) * LS + j] + sGp[(3 * TILE + i) * LS + j]);
            const float yj = yj4[e];
            if (NODE) {
                Gsum += (gi == tm.t) ? g3j4[e] : 0.0f;
                Gsum += (gj == tm.t) ? g3i : 0.0f;
            }
            float Gs = 0.5f * Gsum;
            if (lapl) {
                const float dy = yi - yj;
                Gs += p.c_lap * 0.5f * dy * dy * inv_n2;
            }
            const float gc = Gs * Aij * offd;
            const float Sij = mask_act<RELU>(Mij);
            const float Mji_old = sPM[j * LS + i];
            const bool valid = LOSS && (gi < n) && (gj < n);
            float Sji_old = 0.0f;
            if (LOSS) Sji_old = sigmoidf_(Mji_old);
            if (!diag) {  // this thread also owns the mirror entry (j,i)
                float Mji = Mji_old, mji = sPm[j * LS + i], vji = sPv[j * LS + i];
                const float Sji = mask_act<RELU>(Mji);
                // d(entropy)/dS = log(1-S) - log(S) = -M exactly (S = sigma(M)): no logs on the update path
                float gji = (gc + p.c_size + p.c_ent * mask_dent<RELU>(Mji, Sji) * inv_n2) * mask_dact<RELU>(Mji, Sji);
                if (RELU && !(gi < n && gj < n)) gji = 0.0f;  // padding entries (M = 0) do not exist in the reference: relu'(0) * log(0) is NaN
                adam_update(Mji, mji, vji, gji, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
                sPM[j * LS + i] = Mji;
                sPm[j * LS + i] = mji;
                sPv[j * LS + i] = vji;
                sS[j * LS + i] = mask_act<RELU>(Mji);
                if (valid) {
                    s_size += Sji;
                    s_ent += -Sji * logf(Sji) - (1.0f - Sji) * logf(1.0f - Sji);
                }
            }
            if (valid) {
                s_size += Sij;
                s_ent += -Sij * logf(Sij) - (1.0f - Sij) * logf(1.0f - Sij);
                if (lapl) {
                    const float ab = Aij * 0.5f * (Sij + Sji_old) * offd;
                    s_lap += ab * (yj * yj - yi * yj);
                    if (!diag) s_lap += ab * (yi * yi - yi * yj);
                }
            }
            float gij = (gc + p.c_size + p.c_ent * mask_dent<RELU>(Mij, Sij) * inv_n2) * mask_dact<RELU>(Mij, Sij);
            if (RELU && !(gi < n && gj < n)) gij = 0.0f;
            float mij = mo[e], vij = vo[e];
            adam_update(Mij, mij, vij, gij, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
            Mo[e] = Mij;
            mo[e] = mij;
            vo[e] = vij;
        } else if (!diag) {
            sS[j * LS + i] = mask_act<RELU>(sPM[j * LS + i]);
        }
        Sown[e] = mask_act<RELU>(Mij);
        if (diag) sS[i * LS + j] = Sown[e];  // diagonal tile: publish, the mirror thread reads it transposed
    }
    if (UPDATE) {
        *reinterpret_cast<f32x4*>(p.M + own) = Mo;
        *reinterpret_cast<f32x4*>(p.mM + own) = mo;
        *reinterpret_cast<f32x4*>(p.vM + own) = vo;
    }
    __syncthreads();
    float s_den = 0.0f, s_adj = 0.0f;
    if (UPDATE && LOSS) {   // ExplainModule.mask_density (explain.py:680-683) after optimizer.step() (:142-148): the masked adjacency of the UPDATED mask
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = c4 + e;
            const float ab = Ao[e] * (0.5f * (Sown[e] + sS[j * LS + i]));
            const float both = diag ? 1.0f : 2.0f;       // an off-diagonal tile's thread also stands for the mirror entry
            s_den += (gi != J0 + j) ? both * ab : 0.0f;
            s_adj += both * Ao[e];
        }
    }
    if (WRITE_ABAR) {
        f32x4 ab4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = c4 + e;
            const float Sother = sS[j * LS + i];
            const float ab = Ao[e] * (0.5f * (Sown[e] + Sother));
            // the reference MULTIPLIES by (1 - I) (explain.py:678): a NaN on the diagonal stays NaN (only reachable with ReLU)
            ab4[e] = RELU ? ab * ((gi != J0 + j) ? 1.0f : 0.0f) : ((gi != J0 + j) ? ab : 0.0f);  // A = 0 on padding
        }
        *reinterpret_cast<f32x4*>(p.Abar + own) = ab4;
        if (!diag) {  // same values for (j,i): stage for the row-wise store of the mirror tile
#pragma unroll
            for (int e = 0; e < 4; ++e) sS[(c4 + e) * LS + i] = ab4[e];
        }
        __syncthreads();
    }
    if (!diag) {
        const int a = i * LS + c4;
        if (UPDATE) {
            f32x4 x, y, z;
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = sPM[a + e]; y[e] = sPm[a + e]; z[e] = sPv[a + e]; }
            *reinterpret_cast<f32x4*>(p.M + par) = x;
            *reinterpret_cast<f32x4*>(p.mM + par) = y;
            *reinterpret_cast<f32x4*>(p.vM + par) = z;
        }
        if (WRITE_ABAR) {
            f32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = sS[a + e];
            *reinterpret_cast<f32x4*>(p.Abar + par) = w;
        }
    }
    if (UPDATE && LOSS) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            s_size += __shfl_xor(s_size, o);
            s_ent += __shfl_xor(s_ent, o);
            s_lap += __shfl_xor(s_lap, o);
            s_den += __shfl_xor(s_den, o);
            s_adj += __shfl_xor(s_adj, o);
        }
        if (lane == 0) {  // logging only: float atomics, summation order not fixed
            float* L = p.loss + ((size_t)tl.t * p.num_iters + iter) * NLOSS;
            atomicAdd(&L[LOGD + 1], s_den);   // numerator / denominator of the density: the host forms the quotient (engine.py) - the tiles of a
            atomicAdd(&L[LOGD + 2], s_adj);   // target finish in no particular order
            atomicAdd(&L[1], p.c_size * s_size);
            atomicAdd(&L[2], p.c_lap * s_lap * inv_n2);
            atomicAdd(&L[3], p.c_ent * s_ent * inv_n2);
        }
    }
    // feature-mask Adam step: once per target, by the workgroup that owns tile (0,0)
    if (UPDATE && tl.I == 0 && tl.J == 0 && tid < p.D) {
        const int o = tl.t * FS + tid;
        const float fcur = p.f[iter & 1][o];
        const float ph = sigmoidf_(fcur);
        float dsum = 0.0f;
        for (int rb = 0; rb < (ld >> 5); ++rb) dsum += p.df[((size_t)(tm.offR >> 5) + rb) * FS + tid];
        const float gf = (dsum + p.c_feat_size / (float)p.D) * ph * (1.0f - ph);
        float fnew = fcur, m = p.mf[o], v = p.vf[o];
        adam_update(fnew, m, v, gf, p.omb1, p.beta2, p.omb2, p.eps, step_size, bc2s, p.opt);
        p.mf[o] = m;
        p.vf[o] = v;
        p.f[(iter + 1) & 1][o] = fnew;
    }
}

// ---------------------------------------------------------------------------------------------
// Gradient baseline (`model="grad"`, explain.py:125-133 with adj_feat_grad :717-738): ONE forward + backward of the
// encoder on the UNMASKED sub-graph with loss -log softmax(.)[predicted label], then
//     out = sigmoid(|dL/dA| + |dL/dA|^T) * A .
// The forward / backward launches are the streaming kernels with Abar := A and phi := 1; this kernel forms the two
// tiles G[i][j] = sum_l dZ_l[i] . X_{l-1}[j] and G[j][i] of a tile pair on MFMA (K = D + 2H split over the 4 waves, two
// accumulators because the absolute values are taken before the sum), adds the rank-1 layer-3 part (row t: g3) and
// writes tile (I,J) and its mirror.  One workgroup per tile pair, as k_mask.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grad_edges(Params p, const MaskTile* tiles, float* out) {
    constexpr int LS = 33;
    __shared__ float sGa[4 * TILE * LS], sGb[4 * TILE * LS];
    __shared__ float sS[TILE * LS];
    const MaskTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int ld = tm.ld;
    const int I0 = tl.I * TILE, J0 = tl.J * TILE;
    const bool diag = (tl.I == tl.J);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int i = tid >> 3, c4 = (tid & 7) * 4;
    const int gi = I0 + i;
    const size_t own = tm.offQ + (size_t)gi * ld + J0 + c4;
    const size_t par = tm.offQ + (size_t)(J0 + i) * ld + I0 + c4;
    const f32x4 Ao = *reinterpret_cast<const f32x4*>(p.A + own);
    const f32x4 g3j4 = *reinterpret_cast<const f32x4*>(p.g3 + tm.offR + J0 + c4);
    const float g3i = p.g3[tm.offR + gi];
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) accA[r] = accB[r] = 0.0f;
    const size_t ro = (size_t)tm.offR * FS;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const int d = (l == 0) ? p.D : p.H;
        const float* zT = p.dZT[l] + ro;
        const float* xT = (l == 0) ? p.XT + ro : p.UT[l - 1] + ro;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (2 * (wave + 4 * u) < d) {
                const int k = 2 * (wave + 4 * u) + h;  // columns >= d hold zeros
                const float zi = zT[(size_t)k * ld + I0 + li], zj = zT[(size_t)k * ld + J0 + li];
                float xi = xT[(size_t)k * ld + I0 + li], xj = xT[(size_t)k * ld + J0 + li];
                if (l == 1) {
                    xi = fmaxf(xi, 0.0f);
                    xj = fmaxf(xj, 0.0f);
                }
                accA = __builtin_amdgcn_mfma_f32_32x32x2f32(zi, xj, accA, 0, 0, 0);  // G[I0+i][J0+j]
                accB = __builtin_amdgcn_mfma_f32_32x32x2f32(xi, zj, accB, 0, 0, 0);  // G[J0+j][I0+i]
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        sGa[(wave * TILE + acc_row(r, h)) * LS + li] = accA[r];
        sGb[(wave * TILE + acc_row(r, h)) * LS + li] = accB[r];
    }
    __syncthreads();
    f32x4 o4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = c4 + e, gj = J0 + j;
        float ga = (sGa[(0 * TILE + i) * LS + j] + sGa[(1 * TILE + i) * LS + j]) + (sGa[(2 * TILE + i) * LS + j] + sGa[(3 * TILE + i) * LS + j]);
        float gb = (sGb[(0 * TILE + i) * LS + j] + sGb[(1 * TILE + i) * LS + j]) + (sGb[(2 * TILE + i) * LS + j] + sGb[(3 * TILE + i) * LS + j]);
        ga += (gi == tm.t) ? g3j4[e] : 0.0f;   // layer 3: dZ3 is non-zero on row t only, G3[t][j] = dZ3[t] . relu(U2[j]) = g3[j]
        gb += (gj == tm.t) ? g3i : 0.0f;
        o4[e] = sigmoidf_(fabsf(ga) + fabsf(gb)) * Ao[e];
        if (!diag) sS[j * LS + i] = o4[e];
    }
    *reinterpret_cast<f32x4*>(out + own) = o4;
    if (!diag) {
        __syncthreads();
        f32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = sS[i * LS + c4 + e];
        *reinterpret_cast<f32x4*>(out + par) = w;
    }
}

// per target: column-major copy of X, zero the Adam state of the feature mask, initial feature mask
__global__ __launch_bounds__(256) void k_prep(Params p, const float* f_init, const int32_t* ids) {
    const int t = ids ? ids[blockIdx.x] : (int)blockIdx.x;
    const TargetMeta tm = p.meta[t];
    const float* X = p.X + tm.offR * FS;
    float* XT = const_cast<float*>(p.XT) + tm.offR * FS;
    for (int e = threadIdx.x; e < tm.ld * FS; e += blockDim.x) {
        const int c = e / tm.ld, i = e - c * tm.ld;
        XT[e] = X[(size_t)i * FS + c];
    }
    if (threadIdx.x < FS) {
        const int o = t * FS + threadIdx.x;
        const float* fs = p.fs_in ? p.fs_in + (size_t)t * 3 * FS + threadIdx.x : nullptr;
        const float f0 = f_init ? f_init[o] : fs ? fs[0] : 0.0f;
        p.f[0][o] = f0;
        p.f[1][o] = f0;
        p.mf[o] = fs ? fs[FS] : 0.0f;
        p.vf[o] = fs ? fs[2 * FS] : 0.0f;
    }
}

// streaming path, end of a run: the feature-mask state after num_iters steps -> fs_out [T][3][32]  (gnnx_run_resume)
__global__ __launch_bounds__(64) void k_export_fstate(Params p, const int32_t* ids) {
    const int t = ids ? ids[blockIdx.x] : (int)blockIdx.x;
    if (threadIdx.x < FS) {
        const int o = t * FS + threadIdx.x;
        float* fs = p.fs_out + (size_t)t * 3 * FS + threadIdx.x;
        fs[0] = p.f[p.num_iters & 1][o];
        fs[FS] = p.mf[o];
        fs[2 * FS] = p.vf[o];
    }
}

// ---------------------------------------------------------------------------------------------
// Device-side sub-graph packing (replaces the host's dense slicing adj[nb][:, nb], feat[nb] of
// explainer/explain.py:492-501 for a whole batch): one workgroup per 32-row block.  Row i of target t is node
// u = nb[i]; each CSR neighbour v of u is located in the (ascending) neighbour list by binary search and its
// weight written to A[i][j].  A, X, yhat must be zero-filled by the caller (padding stays zero).
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const int64_t* indptr;   // CSR of the full graph [N+1]
    const int32_t* indices;  // [nnz]
    const float* weights;    // [nnz] or null (all ones)
    const float* feat;       // [N][feat_stride]
    int32_t feat_stride;
    const float* pred_label; // [N] predicted class id as float, or null (graph-free use)
    const int32_t* nb;       // concatenated neighbour lists
    const int64_t* nb_off;   // [T+1]
    float* A;
    float* X;
    float* yhat;
    int32_t D;
    // by-products for the analysis that follows the packing (gnnx_pack_csr_analyze), both [R] or null: every row's off-diagonal
    // non-zeros (what k_row_degrees counts) and its upper-triangle non-zeros (k_edge_rowcount) - the packing places every entry of the
    // row anyway, so two launches and two passes over A fall away.  (The CSR holds every (u, v) once - engine.device_graph sums duplicates; a
    // hand-built CSR with sorted rows may repeat an entry: it is counted once, as it occupies one cell of A.)
    int32_t* rowdeg;
    int32_t* rowcnt;
};

__global__ __launch_bounds__(256) void k_pack(PackArgs a, const ConvTile* tiles) {
    const ConvTile tl = tiles[blockIdx.x];
    const TargetMeta tm = tl.tm;
    const int row = threadIdx.x >> 3, part = threadIdx.x & 7;
    const int i = tl.rb * TILE + row;
    // the block's 32 rows of A (ld wide, contiguous), of X and of yhat start as zeros (padding rows and columns stay zero): done
    // here instead of three memsets over the whole batch - three launches fewer for a pipeline whose short kernels queue for CUs
    {
        f32x4* Ab = reinterpret_cast<f32x4*>(a.A + tm.offQ + (size_t)tl.rb * TILE * tm.ld);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int e = threadIdx.x; e < TILE * tm.ld / 4; e += 256) Ab[e] = z;
        reinterpret_cast<f32x4*>(a.X + ((size_t)tm.offR + tl.rb * TILE) * FS)[threadIdx.x] = z;   // 32 rows x 32 floats = 256 x 16 B
        if (a.yhat && threadIdx.x < TILE) a.yhat[tm.offR + tl.rb * TILE + threadIdx.x] = 0.0f;
    }
    __syncthreads();
    const bool live = i < tm.n;   // (all eight lanes of a row agree; padding rows stay zero and report zero counts)
    const int32_t* nb = a.nb + a.nb_off[tl.t];
    const int u = live ? nb[i] : 0;
    int deg = 0, up = 0;
    if (live) {
        float* Arow = a.A + tm.offQ + (size_t)i * tm.ld;
        for (int64_t e = a.indptr[u] + part; e < a.indptr[u + 1]; e += 8) {
            const int v = a.indices[e];
            int lo = 0, hi = tm.n - 1;
            while (lo < hi) {  // lower bound of v in the ascending neighbour list
                const int mid = (lo + hi) >> 1;
                if (nb[mid] < v) lo = mid + 1; else hi = mid;
            }
            if (nb[lo] == v) {
                const float w = a.weights ? a.weights[e] : 1.0f;
                Arow[lo] = w;
                // A repeated (u, v) lands in ONE cell of A, so it must count once: in a row with sorted indices (the precondition of
                // gnnx_pack_csr; engine.device_graph canonicalises and sums duplicates) a repeat is adjacent to its first occurrence.
                const bool rep = e > a.indptr[u] && a.indices[e - 1] == v;
                deg += (!rep && w != 0.0f && lo != i);
                up += (!rep && w != 0.0f && lo > i);
            }
        }
    }
    if (a.rowdeg || a.rowcnt) {   // the row's eight lanes are adjacent lanes of one wave
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            deg += __shfl_xor(deg, o);
            up += __shfl_xor(up, o);
        }
        if (part == 0 && a.rowdeg) a.rowdeg[tm.offR + i] = deg;
        if (part == 0 && a.rowcnt) a.rowcnt[tm.offR + i] = up;
    }
    if (!live) return;
    float* Xrow = a.X + ((size_t)tm.offR + i) * FS;
    for (int c = part; c < a.D; c += 8) Xrow[c] = a.feat[(size_t)u * a.feat_stride + c];
    if (part == 0 && a.pred_label) a.yhat[tm.offR + i] = a.pred_label[u];
}

}  // namespace gnnx
