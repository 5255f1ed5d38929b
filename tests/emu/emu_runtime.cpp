#include <cstdlib>
#include <cmath>
// TEST INFRASTRUCTURE ONLY — fiber scheduler behind tests/emu/include/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

emu_uint3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {
namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
};
struct WaveSlot {
    int gen = 0, arrived = 0;
    float fa[2][64], fb[2][64];
    int ia[2][64];
};
struct Block {
    std::vector<Fiber> fibers;
    ucontext_t main;
    int cur = 0, nthreads = 0;
    int bar_count = 0, bar_gen = 0;
    int nexited = 0;   // threads that have returned: s_barrier waits on the surviving waves only (the hardware's rule)
    std::vector<WaveSlot> waves;
    const std::function<void()>* body = nullptr;
};
Block* g_blk = nullptr;
unsigned long g_progress = 0;  // bumped whenever a collective / barrier completes or a fiber ends

void yield() { swapcontext(&g_blk->fibers[g_blk->cur].ctx, &g_blk->main); }

void trampoline() {
    Block* b = g_blk;
    (*b->body)();
    b->fibers[b->cur].done = true;
    ++g_progress;
    ++b->nexited;      // a barrier the others already wait at may be complete now ("if some waves have terminated, waits on the surviving ones")
    if (b->bar_count > 0 && b->bar_count == b->nthreads - b->nexited) {
        b->bar_count = 0;
        b->bar_gen++;
    }
    swapcontext(&b->fibers[b->cur].ctx, &b->main);
}

template <class F>
void wave_collective(F publish) {
    Block* b = g_blk;
    const int tid = b->cur;
    WaveSlot& w = b->waves[tid >> 6];
    const int g = w.gen;
    publish(w, g & 1, tid & 63);
    const int wsize = std::min(64, b->nthreads - (tid >> 6) * 64);
    if (++w.arrived == wsize) {
        w.arrived = 0;
        w.gen++;
        ++g_progress;
    } else {
        while (w.gen == g) yield();
    }
}
}  // namespace

void syncthreads() {
    Block* b = g_blk;
    const int g = b->bar_gen;
    if (++b->bar_count == b->nthreads - b->nexited) {
        b->bar_count = 0;
        b->bar_gen++;
        ++g_progress;
    } else {
        while (b->bar_gen == g) yield();
    }
}

float shfl_xor_f(float v, int mask) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.fa[bf][ln] = v; buf = bf; lane = ln; });
    return g_blk->waves[g_blk->cur >> 6].fa[buf][(lane ^ mask) & 63];
}

float shfl_f(float v, int src) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.fa[bf][ln] = v; buf = bf; lane = ln; });
    (void)lane;
    return g_blk->waves[g_blk->cur >> 6].fa[buf][src & 63];
}

int shfl_i(int v, int src) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.ia[bf][ln] = v; buf = bf; lane = ln; });
    (void)lane;
    return g_blk->waves[g_blk->cur >> 6].ia[buf][src & 63];
}

unsigned long long ballot(bool pred) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.ia[bf][ln] = pred ? 1 : 0; buf = bf; lane = ln; });
    (void)lane;
    const WaveSlot& w = g_blk->waves[g_blk->cur >> 6];
    const int wsize = std::min(64, g_blk->nthreads - (g_blk->cur >> 6) * 64);
    unsigned long long m = 0;
    for (int l = 0; l < wsize; ++l)
        if (w.ia[buf][l]) m |= 1ull << l;
    return m;
}

// rendezvous of the lanes of one wave (fibers between collectives are NOT in lockstep, unlike hardware lanes)
void wave_sync() {
    wave_collective([&](WaveSlot&, int, int) {});
}

int dpp_row_shl(int v, int shift) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.ia[bf][ln] = v; buf = bf; lane = ln; });
    const int src = lane + shift;
    if ((src >> 4) != (lane >> 4)) return 0;  // beyond the 16-lane row: bound_ctrl -> 0
    return g_blk->waves[g_blk->cur >> 6].ia[buf][src];
}

// DPP row_ror:S (dpp_ctrl 0x120 + S): lane l <- lane (l - S) mod 16 of its 16-lane row
int dpp_row_ror(int v, int shift) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.ia[bf][ln] = v; buf = bf; lane = ln; });
    const int src = (lane & ~15) | ((lane - shift) & 15);
    return g_blk->waves[g_blk->cur >> 6].ia[buf][src];
}

int shfl_xor_i(int v, int mask) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.ia[bf][ln] = v; buf = bf; lane = ln; });
    return g_blk->waves[g_blk->cur >> 6].ia[buf][(lane ^ mask) & 63];
}

// v_mfma_f32_32x32x2_f32: A[i][k] from lane i + 32k, B[k][j] from lane j + 32k,
// D[i][j] in lane j + 32*((i>>2)&1), register (i&3) + 4*(i>>3); k-ordered fma chain.
// GNNX_EMU_ULP_NOISE=<seed>: the results of the "hardware form" intrinsics move by -1 / 0 / +1 ulp at random - a stand-in for
// v_rcp_f32 / v_sqrt_f32 / v_exp_f32, used to ask which targets' trajectories depend on that last bit (tools/ulp_sensitivity.py)
float hw_form(float v) {
    static int mode = -1;
    static unsigned long long state = 0;
    if (mode < 0) {
        const char* env = std::getenv("GNNX_EMU_ULP_NOISE");
        mode = env ? 1 : 0;
        state = env ? 0x9E3779B97F4A7C15ull * (unsigned long long)(std::atoll(env) + 1) : 0;
    }
    if (!mode || !std::isfinite(v) || v == 0.0f) return v;
    state = state * 6364136223846793005ull + 1442695040888963407ull;
    const int k = (int)((state >> 33) % 3) - 1;
    return k == 0 ? v : std::nextafter(v, k > 0 ? INFINITY : -INFINITY);
}

f32x16 mfma_32x32x2(float a, float b, f32x16 c) {
    int buf = 0, lane = 0;
    wave_collective([&](WaveSlot& w, int bf, int ln) { w.fa[bf][ln] = a; w.fb[bf][ln] = b; buf = bf; lane = ln; });
    const WaveSlot& w = g_blk->waves[g_blk->cur >> 6];
    const int j = lane & 31, h = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = std::fmaf(w.fa[buf][i + 32 * k], w.fb[buf][j + 32 * k], acc);
        c[r] = acc;
    }
    return c;
}

void launch(unsigned grid, unsigned block, const std::function<void()>& body) {
    static std::vector<char*> stacks;
    while (stacks.size() < block) stacks.push_back(static_cast<char*>(malloc(STACK)));
    Block blk;
    blk.nthreads = (int)block;
    blk.body = &body;
    blk.fibers.resize(block);
    blk.waves.resize((block + 63) / 64);
    g_blk = &blk;
    gridDim = {grid, 1, 1};
    blockDim = {block, 1, 1};
    for (unsigned bid = 0; bid < grid; ++bid) {
        blk.bar_count = 0;
        blk.nexited = 0;
        for (auto& w : blk.waves) w.arrived = 0;
        for (unsigned t = 0; t < block; ++t) {
            Fiber& f = blk.fibers[t];
            f.done = false;
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = stacks[t];
            f.ctx.uc_stack.ss_size = STACK;
            f.ctx.uc_link = &blk.main;
            makecontext(&f.ctx, trampoline, 0);
        }
        unsigned ndone = 0;
        while (ndone < block) {
            const unsigned long before = g_progress;
            ndone = 0;
            for (unsigned t = 0; t < block; ++t) {
                if (blk.fibers[t].done) { ++ndone; continue; }
                blk.cur = (int)t;
                threadIdx = {t, 0, 0};
                blockIdx = {bid, 0, 0};
                swapcontext(&blk.main, &blk.fibers[t].ctx);
            }
            if (ndone < block && g_progress == before) {
                fprintf(stderr, "hip emulator: deadlock in block %u (a wave collective or __syncthreads is not reached by every "
                                "lane: divergent shuffle / barrier)\n", bid);
                abort();
            }
        }
    }
    g_blk = nullptr;
}
}  // namespace emu
