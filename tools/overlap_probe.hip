// Measurement tool: does a long, partial-occupancy kernel on stream B overlap with a chain of small dependent
// kernels replayed from a hipGraph on stream A?  (Viability of "resident kernel for small targets beside the
// launch chain of the large ones".)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, float* out) {
    const long long t0 = clock64();
    float x = threadIdx.x;
    while (clock64() - t0 < cycles) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) out[0] = x;
}
__global__ void small(float* buf, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = buf[i] * 1.0001f + 1.0f;
}
int main() {
    float *buf, *out; hipMalloc(&buf, 4 << 20); hipMalloc(&out, 64);
    hipStream_t a, b; hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(a, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < 1500; ++i) small<<<800, 256, 0, a>>>(buf, 1 << 20);
    hipStreamEndCapture(a, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](bool chain, bool side, int side_blocks) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (side) spin<<<side_blocks, 256, 0, b>>>(10LL * 1000 * 100, out);  // 100 MHz clock64 -> ~10 ms
        if (chain) hipGraphLaunch(ge, a);
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    run(true, false, 0);
    printf("chain alone            %.2f ms\n", run(true, false, 0));
    for (int blocks : {64, 256, 1024}) {
        printf("spin(%4d WG) alone     %.2f ms\n", blocks, run(false, true, blocks));
        printf("chain + spin(%4d WG)   %.2f ms\n", blocks, run(true, true, blocks));
    }
    return 0;
}
