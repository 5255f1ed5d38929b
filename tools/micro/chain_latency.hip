// Measurement tool (GPU box): latencies of the DEPENDENT operations the on-chip-resident kernels' iteration chain is made of
// (gnnx_sparse.hpp: gather -> combine -> row-local -> store -> barrier, seven phases per iteration), measured the way the kernels
// meet them - one wave per SIMD (a lone chain), two waves per SIMD (the 512-thread class: 8 waves per CU), and with every CU busy.
// bench.py's `roofline.chain` prices the dependent operations of an iteration with these figures (profiles/r05_chain_latency.txt).
//   lds_load     : pointer chase through LDS, ds_read_b32 -> address of the next ds_read_b32
//   lds_gather2  : (index load -> row load) pairs, the two dependent loads of a sparse gather step
//   shuffle      : __shfl_xor (ds_bpermute through the LDS crossbar) on a dependent value
//   dpp          : row_shl DPP add on a dependent value (no LDS)
//   fma          : dependent v_fma_f32
//   mfma32x32x2  : dependent-accumulator v_mfma_f32_32x32x2_f32
//   transc       : dependent v_exp_f32 -> v_rcp_f32 (a sigmoid's two hardware forms) + the add between them
//   store_sync_ld: ds_write_b32 -> wave-level sync -> ds_read_b32 of another lane's value (a hand-over inside one wave)
//   barrier      : __syncthreads() with all waves of the workgroup arriving together
// Build: hipcc --offload-arch=gfx950 -O3 chain_latency.hip -o chain_latency ; run: ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDS_WORDS = 8192;

template <int OP>
__global__ void k_chain(int steps, float* out, long long* cyc) {
    __shared__ int lds[LDS_WORDS];
    const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
    // a permutation with a long cycle, strided so that the 64 lanes of a wave hit 64 different banks (no conflicts)
    for (int e = tid; e < LDS_WORDS; e += nt) lds[e] = (e + 64 * 37) % LDS_WORDS;
    __syncthreads();
    int idx = tid % LDS_WORDS;
    float v = 1.0f + 1e-3f * lane;
    f32x16 c16;
    for (int g = 0; g < 16; ++g) c16[g] = 0.0f;
    const long long t0 = clock64();
    // (every chain is unrolled 16-fold: a lone wave spends ~24 cycles on a loop trip's own add / compare / branch)
    if (OP == 0) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) idx = lds[idx];
        }
    } else if (OP == 1) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int col = lds[idx];                       // the entry's column
                idx = lds[(col + 1) & (LDS_WORDS - 1)];         // the row it points at
            }
        }
    } else if (OP == 2) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v = v * 0.999f + __shfl_xor(v, 32);      // (ds_bpermute + one fma: the kernels' xor32_sum)
        }
    } else if (OP == 3) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u)      // (one v_fmac with a DPP source: the kernels' row_shl combine step)
                v = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true)), 0.5f, v * 0.999f);
        }
    } else if (OP == 4) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v = fmaf(v, 0.999f, 1e-3f);
        }
    } else if (OP == 5) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 1e-3f, c16, 0, 0, 0);
        }
        v += c16[0] + c16[7];
    } else if (OP == 6) {
        for (int s = 0; s < steps; s += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) v = __builtin_amdgcn_rcpf(1.0f + __expf(-v));      // (one sigmoid: exp, add, rcp; "transc" = half of it)
        }
    } else if (OP == 7) {
        float* fl = reinterpret_cast<float*>(lds);
        for (int s = 0; s < steps; ++s) {
            fl[tid] = v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v = fl[tid ^ 1] * 0.999f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    } else if (OP == 8) {
        for (int s = 0; s < steps; ++s) {
            v = fmaf(v, 0.999f, 1e-3f);
            __syncthreads();
        }
    }
    const long long t1 = clock64();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * nt + tid] = v + (float)idx;
}

template <int OP>
static void run(const char* name, int threads, int blocks, int steps, float* out, long long* cyc) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_chain<OP>, dim3(blocks), dim3(threads), 0, 0, 64, out, cyc);   // code object + warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chain<OP>, dim3(blocks), dim3(threads), 0, 0, steps, out, cyc);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> c(blocks);
    CK(hipMemcpy(c.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double mean = 0.0;
    for (long long x : c) mean += (double)x;
    mean /= blocks;
    printf("%-14s %5d threads x %4d workgroups: %8.2f ns per dependent step (launch %.3f ms / %d steps), %7.1f clock64 ticks per step\n", name, threads, blocks,
           ms * 1e6 / steps, ms, steps, mean / steps);
}

int main() {
    float* out;
    long long* cyc;
    CK(hipMalloc(&out, sizeof(float) * 1024 * 1024));
    CK(hipMalloc(&cyc, sizeof(long long) * 1024));
    const int steps = 204800;
    const int shapes[4][2] = {{64, 1}, {256, 256}, {512, 256}, {1024, 256}};   // one wave alone; 1 / 2 / 4 waves per SIMD on every CU
    for (auto& sh : shapes) {
        run<0>("lds_load", sh[0], sh[1], steps, out, cyc);
        run<1>("lds_gather2", sh[0], sh[1], steps, out, cyc);
        run<2>("shuffle", sh[0], sh[1], steps, out, cyc);
        run<3>("dpp", sh[0], sh[1], steps, out, cyc);
        run<4>("fma", sh[0], sh[1], steps, out, cyc);
        run<5>("mfma32x32x2", sh[0], sh[1], steps, out, cyc);
        run<6>("transc", sh[0], sh[1], steps, out, cyc);
        run<7>("store_sync_ld", sh[0], sh[1], steps, out, cyc);
        run<8>("barrier", sh[0], sh[1], steps, out, cyc);
    }
    return 0;
}
