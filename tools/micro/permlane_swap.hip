// Measurement / semantics check (GPU box): v + __shfl_xor(v, 32) and v + __shfl_xor(v, 16) through gfx950's v_permlane32_swap / v_permlane16_swap
// (VALU lane swaps, no LDS crossbar) - bit-identical to the ds_bpermute forms on random data, and the latency of a dependent chain of each.
//   hipcc --offload-arch=gfx950 -O3 -o permlane_swap permlane_swap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__device__ __forceinline__ float x32_swap(float v) {
    // (inline assembly: with this compiler - ROCm 7.2 - the sum of the builtin's two results comes out as r[0] + r[0]; the wait states the
    //  hazard recogniser would insert around a lane swap are written out)
    int a = __builtin_bit_cast(int, v), b = a;
    asm volatile("s_nop 0\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float x16_swap(float v) {
    int a = __builtin_bit_cast(int, v), b = a;
    asm volatile("s_nop 0\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__global__ void k_check(const float* in, float* out) {
    const float v = in[threadIdx.x];
    out[threadIdx.x] = x32_swap(v);
    out[64 + threadIdx.x] = v + __shfl_xor(v, 32);
    out[128 + threadIdx.x] = x16_swap(v);
    out[192 + threadIdx.x] = v + __shfl_xor(v, 16);
}
template <int MODE>
__global__ void k_chain(float* out, int n, long long* ticks) {
    float v = out[threadIdx.x];
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) v = x32_swap(v) * 0.5f;
        if (MODE == 1) v = (v + __shfl_xor(v, 32)) * 0.5f;
        if (MODE == 2) v = x16_swap(v) * 0.5f;
        if (MODE == 3) v = (v + __shfl_xor(v, 16)) * 0.5f;
    }
    const long long t1 = wall_clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
int main() {
    float h[64], o[256];
    srand(7);
    for (int i = 0; i < 64; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    float *din, *dout;
    long long* dt;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, sizeof(o)); hipMalloc(&dt, 8);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("v + shfl_xor(v, 32): swap form bit-identical in %s lanes; v + shfl_xor(v, 16): %s\n",
           memcmp(o, o + 64, 256) == 0 ? "all 64" : "NOT all", memcmp(o + 128, o + 192, 256) == 0 ? "all 64" : "NOT all");
    const int n = 4000;
    const char* names[4] = {"xor 32 via v_permlane32_swap", "xor 32 via ds_bpermute", "xor 16 via v_permlane16_swap", "xor 16 via ds_bpermute"};
    for (int m = 0; m < 4; ++m) {
        long long t = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, dout, n, dt);
            if (m == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, dout, n, dt);
            if (m == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, dout, n, dt);
            if (m == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, dout, n, dt);
            hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
        }
        printf("%-32s dependent step (sum + one multiply): %.1f ns\n", names[m], (double)t * 10.0 / n);   // wall_clock64: 100 MHz
    }
    return 0;
}
