// Measurement tool (GPU box): how fast can 4-wave workgroups stream a matrix the way k_conv reads Abar?
//   mode 0  "strided": wave-load = two 128-B row segments, rows ld*4 bytes apart (the row-major ld x ld block k_conv reads)
//   mode 1  "panel":   the same bytes laid out per 32-column panel, [panel][k][32]: every wave reads a contiguous stream
//   mode 2  "strided, 512-B segments": 4 adjacent panels per workgroup, one per wave, whole K each (adjacent 128-B segments)
// Same loads in flight per wave in all modes (2 batches of 16 k pairs), a dependent use of every batch, no MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 stream_pattern.hip -o stream_pattern ; run: ./stream_pattern [ld=4992] [copies=3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(const float* __restrict__ A, float* __restrict__ out, int ld, int nb) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int mat = blockIdx.x / (MODE == 2 ? nb / 4 : nb), unit = blockIdx.x % (MODE == 2 ? nb / 4 : nb);
    const float* base = A + (size_t)mat * ld * ld;
    int rb, k0, kchunk;
    if (MODE == 2) { rb = unit * 4 + wave; k0 = h; kchunk = ld; }
    else { rb = unit; kchunk = ld >> 2; k0 = wave * kchunk + h; }
    constexpr int KB = 16;
    float a0[KB], a1[KB];
    auto addr = [&](int k) -> const float* {
        return MODE == 1 ? base + ((size_t)rb * ld + k) * 32 + li : base + (size_t)k * ld + rb * 32 + li;
    };
    auto loadb = [&](float (&a)[KB], int s0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const bool on = (s0 + 2 * u) < kchunk;
            a[u] = on ? *addr(k0 + s0 + 2 * u) : 0.0f;
        }
    };
    float acc = 0.0f;
    auto use = [&](const float (&a)[KB]) {
#pragma unroll
        for (int u = 0; u < KB; ++u) acc = fmaf(a[u], 1.0001f, acc);
    };
    loadb(a0, 0);
    for (int s0 = 0; s0 < kchunk; s0 += 4 * KB) {
        const bool more1 = s0 + 2 * KB < kchunk;
        if (more1) loadb(a1, s0 + 2 * KB);
        use(a0);
        if (s0 + 4 * KB < kchunk) loadb(a0, s0 + 4 * KB);
        if (more1) use(a1);
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int MODE>
static float run(const float* A, float* out, int ld, int copies, int reps) {
    const int nb = ld / 32;
    const int grid = copies * (MODE == 2 ? nb / 4 : nb);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_stream<MODE>, dim3(grid), dim3(256), 0, 0, A, out, ld, nb);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream<MODE>, dim3(grid), dim3(256), 0, 0, A, out, ld, nb);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}


// k_conv's K loop step by step: FEAT bit 0 = the MFMA per k pair, bit 1 = the B operand loads (L2-resident rows), bit 2 = 25 KB of LDS
// per workgroup (k_conv's footprint), bit 3 = the split-K tile exchange through LDS at the end.
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int FEAT>
__global__ __launch_bounds__(256) void k_like(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int ld, int nb) {
    __shared__ float red[(FEAT & 12) ? 4 * 32 * 33 + 2 * 32 * 33 : 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, h = lane >> 5;
    const int mat = blockIdx.x / nb, rb = blockIdx.x % nb;
    const float* Ab = A + (size_t)mat * ld * ld + rb * 32 + li;
    const float* Bs = B + (size_t)mat * ld * 32 + li;
    const int kchunk = ld >> 2, k0 = wave * kchunk + h, klast = kchunk - 2;
    constexpr int KB = 16;
    float a0[KB], a1[KB], b0[KB], b1[KB];
    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    float accs = 0.0f;
    auto loadb = [&](float (&a)[KB], float (&b)[KB], int s0) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            const int ks = s0 + 2 * u;
            const int k = k0 + (ks < klast ? ks : klast);
            const float av = Ab[(size_t)k * ld];
            const float bv = (FEAT & 2) ? Bs[(size_t)k * 32] : 1.0f;
            a[u] = ks < kchunk ? av : 0.0f;
            b[u] = ks < kchunk ? bv : 0.0f;
        }
    };
    auto use = [&](const float (&a)[KB], const float (&b)[KB]) {
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            if ((FEAT & 16) && (u & 1)) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc2, 0, 0, 0);
            else if (FEAT & 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            else accs = fmaf(a[u], b[u], accs);
        }
    };
    loadb(a0, b0, 0);
    loadb(a1, b1, 2 * KB);
    for (int s0 = 0; s0 < kchunk; s0 += 4 * KB) {
        use(a0, b0);
        loadb(a0, b0, s0 + 4 * KB);
        use(a1, b1);
        loadb(a1, b1, s0 + 6 * KB);
    }
    float r = accs;
#pragma unroll
    for (int q = 0; q < 16; ++q) r += acc[q] + acc2[q];
    if (FEAT & 8) {
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 33 + li] = acc[q];
        __syncthreads();
        r = red[tid] + red[tid + 1056] + red[tid + 2112] + red[tid + 3168];
    }
    if ((FEAT & 4) && r == 77.7f) red[tid] = r;
    if (r == 123.456f) out[blockIdx.x] = r + ((FEAT & 12) ? red[(tid * 7) & 31] : 0.0f);
}

template <int FEAT>
static float run_like(const float* A, const float* B, float* out, int ld, int copies, int reps) {
    const int nb = ld / 32, grid = copies * nb;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_like<FEAT>, dim3(grid), dim3(256), 0, 0, A, B, out, ld, nb);
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_like<FEAT>, dim3(grid), dim3(256), 0, 0, A, B, out, ld, nb);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char** argv) {
    const int ld = argc > 1 ? atoi(argv[1]) : 4992, copies = argc > 2 ? atoi(argv[2]) : 3;
    const size_t n = (size_t)copies * ld * ld;
    float *A, *out;
    CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(A, 0, n * 4));
    const double gb = n * 4 / 1e9;
    const float t0 = run<0>(A, out, ld, copies, 10), t1 = run<1>(A, out, ld, copies, 10), t2 = run<2>(A, out, ld, copies, 10);
    printf("ld %d x %d matrices (%.0f MB): strided 128-B segments %.1f us = %.2f TB/s | panel layout %.1f us = %.2f TB/s | 4 adjacent panels per workgroup %.1f us = %.2f TB/s\n",
           ld, copies, gb * 1e3, t0 * 1e3, gb / t0, t1 * 1e3, gb / t1, t2 * 1e3, gb / t2);
    float* B;
    CK(hipMalloc(&B, (size_t)copies * ld * 32 * 4)); CK(hipMemset(B, 0, (size_t)copies * ld * 32 * 4));
    const float f0 = run_like<0>(A, B, out, ld, copies, 10), f1 = run_like<1>(A, B, out, ld, copies, 10), f2 = run_like<2>(A, B, out, ld, copies, 10),
                f3 = run_like<3>(A, B, out, ld, copies, 10), f7 = run_like<7>(A, B, out, ld, copies, 10), f15 = run_like<15>(A, B, out, ld, copies, 10),
                f4 = run_like<4>(A, B, out, ld, copies, 10), f17 = run_like<17>(A, B, out, ld, copies, 10), f19 = run_like<19>(A, B, out, ld, copies, 10);
    printf("   two accumulators (independent MFMA chains): +MFMA %.2f | +MFMA +B %.2f TB/s\n", gb / f17, gb / f19);
    printf("   k_conv-like K loop, TB/s of A: plain %.2f | +MFMA %.2f | +B loads %.2f | +MFMA +B %.2f | +MFMA +B +25 KB LDS %.2f | + tile exchange %.2f | plain + 25 KB LDS %.2f\n",
           gb / f0, gb / f1, gb / f2, gb / f3, gb / f7, gb / f15, gb / f4);
    return 0;
}
