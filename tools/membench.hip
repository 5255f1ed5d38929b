// Microbenchmark (measurement tool, not product code): read-modify-write bandwidth of the fused mask kernel's
// access pattern on MI355X.  7 tile reads + 8 tile writes per tile pair, 16 B per lane, for
//   mode 0: row-major square arrays (a 32x32 tile = 32 pieces of 128 B at stride 4*ld)
//   mode 1: tile-major arrays (a tile = one contiguous 4 KB block)
//   mode 2: plain streaming RMW of the same bytes (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_tiles(float* M, float* m, float* v, const float* A, float* Ab, int nb, int ld, int mode) {
    // decode tile pair index -> (I, J), I <= J
    int pidx = blockIdx.x, I = 0;
    while (pidx >= nb - I) { pidx -= nb - I; ++I; }
    const int J = I + pidx;
    const int lane = threadIdx.x, rl = lane >> 3, c4 = (lane & 7) * 4;
    f32x4 a[4], b[4], c[4], d[4], e[4], f[4], g[4];
    size_t own[4], par[4];
    for (int q = 0; q < 4; ++q) {
        if (mode == 0) {
            own[q] = (size_t)(I * 32 + 8 * q + rl) * ld + J * 32 + c4;
            par[q] = (size_t)(J * 32 + 8 * q + rl) * ld + I * 32 + c4;
        } else {
            own[q] = ((size_t)I * nb + J) * 1024 + (8 * q + rl) * 32 + c4;
            par[q] = ((size_t)J * nb + I) * 1024 + (8 * q + rl) * 32 + c4;
        }
        a[q] = *(const f32x4*)(M + own[q]); b[q] = *(const f32x4*)(m + own[q]); c[q] = *(const f32x4*)(v + own[q]);
        d[q] = *(const f32x4*)(A + own[q]);
        e[q] = *(const f32x4*)(M + par[q]); f[q] = *(const f32x4*)(m + par[q]); g[q] = *(const f32x4*)(v + par[q]);
    }
    for (int q = 0; q < 4; ++q) {
        a[q] = a[q] * 0.999f + d[q]; b[q] = b[q] * 0.9f + a[q]; c[q] = c[q] * 0.99f + b[q];
        e[q] = e[q] * 0.999f + d[q]; f[q] = f[q] * 0.9f + e[q]; g[q] = g[q] * 0.99f + f[q];
        *(f32x4*)(M + own[q]) = a[q]; *(f32x4*)(m + own[q]) = b[q]; *(f32x4*)(v + own[q]) = c[q]; *(f32x4*)(Ab + own[q]) = a[q] + e[q];
        if (I != J) { *(f32x4*)(M + par[q]) = e[q]; *(f32x4*)(m + par[q]) = f[q]; *(f32x4*)(v + par[q]) = g[q]; *(f32x4*)(Ab + par[q]) = a[q] + e[q]; }
    }
}
__global__ __launch_bounds__(256) void k_stream(float* M, float* m, float* v, const float* A, float* Ab, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 a = ((f32x4*)M)[i], b = ((f32x4*)m)[i], c = ((f32x4*)v)[i], d = ((const f32x4*)A)[i];
        a = a * 0.999f + d; b = b * 0.9f + a; c = c * 0.99f + b;
        ((f32x4*)M)[i] = a; ((f32x4*)m)[i] = b; ((f32x4*)v)[i] = c; ((f32x4*)Ab)[i] = a + b;
    }
}
int main() {
    for (int nb : {51, 160}) {
        const int ld = nb * 32; const size_t Q = (size_t)ld * ld;
        float *M, *m, *v, *A, *Ab;
        hipMalloc(&M, Q * 4); hipMalloc(&m, Q * 4); hipMalloc(&v, Q * 4); hipMalloc(&A, Q * 4); hipMalloc(&Ab, Q * 4);
        hipMemset(M, 0, Q * 4); hipMemset(m, 0, Q * 4); hipMemset(v, 0, Q * 4); hipMemset(A, 0, Q * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int pairs = nb * (nb + 1) / 2, reps = 30;
        for (int mode = 0; mode < 3; ++mode) {
            for (int w = 0; w < 3; ++w) {
                if (mode < 2) k_tiles<<<pairs, 64>>>(M, m, v, A, Ab, nb, ld, mode); else k_stream<<<2048, 256>>>(M, m, v, A, Ab, Q / 4);
            }
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (mode < 2) k_tiles<<<pairs, 64>>>(M, m, v, A, Ab, nb, ld, mode); else k_stream<<<2048, 256>>>(M, m, v, A, Ab, Q / 4);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            const double bytes = 8.0 * Q * 4;  // 4 arrays read + 4 written
            printf("nb=%d Q=%.1f MB/array mode=%d  %.1f us  %.2f TB/s\n", nb, Q * 4 / 1e6, mode, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
        }
        hipFree(M); hipFree(m); hipFree(v); hipFree(A); hipFree(Ab);
    }
    return 0;
}
