// gnnx_host_rng.cpp — include/gnnx_host.h: the seeded initial masks of a batch, drawn by ATen's own CPU normal_ from C++ threads.
// Replaces, for a whole batch, the per-target `mask.normal_(1.0, std)` of construct_edge_mask (explainer/explain.py:645-652).
#include <ATen/ATen.h>
#include <ATen/CPUGeneratorImpl.h>
#include <c10/core/InferenceMode.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gnnx_host.h"

namespace {

thread_local std::string g_err;

// A small persistent pool: starting 16 threads costs about as much as drawing the syn1 masks.
class Pool {
  public:
    explicit Pool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this, i] { loop(i); });
    }
    int size() const { return (int)workers_.size(); }
    // run fn(i) for i in [0, parts) on the workers and wait
    // (limit: at most that many workers take items - the ranks of a node share its cores, each is told its share)
    void run(int parts, const std::function<void(int)>& fn, int limit = 1 << 30) {
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        limit_ = limit;
        parts_ = parts;
        next_ = 0;
        done_ = 0;
        ++epoch_;
        cv_.notify_all();
        cv_done_.wait(lk, [&] { return done_ == parts_; });
        fn_ = nullptr;
    }

  private:
    void loop(int me) {
        c10::InferenceMode ng;
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return epoch_ != seen; });
            seen = epoch_;
            while (me < limit_ && next_ < parts_) {
                const int i = next_++;
                lk.unlock();
                (*fn_)(i);
                lk.lock();
                if (++done_ == parts_) cv_done_.notify_all();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(int)>* fn_ = nullptr;
    int parts_ = 0, next_ = 0, done_ = 0, limit_ = 1 << 30;
    unsigned long long epoch_ = 0;
};

Pool* g_pool = nullptr;
std::mutex g_pool_mu;

}  // namespace

extern "C" const char* gnnx_host_last_error(void) { return g_err.c_str(); }

// One target's n x n values are ONE normal_ call of the reference (explain.py:645-652), i.e. one sequential pass over one mt19937
// stream: the largest target of a batch (BA-House x100k: n = 5600, 31 M values = 0.3 s on one thread) used to bound the whole draw.
// ATen's CPU normal_ on a contiguous float tensor of >= 16 values (normal_fill, aten/src/ATen/native/cpu/DistributionTemplates.h) first
// fills the tensor with ONE 32-bit engine draw per value, then transforms 16 values at a time (Box-Muller on values i .. i + 7 and
// i + 8 .. i + 15), and if the size is not a multiple of 16 redraws the last 16 values from 16 further engine draws.  So values
// [a, b) of the tensor, a and b multiples of 16, are exactly normal_ of b - a values from the engine state after `a` draws, and a last
// slice [a, N) is normal_ of N - a values from that state (its tail redraw consumes the draws that follow the N fill draws, as in the
// whole tensor).  A "walker" steps the raw engine (about 2 ns per draw instead of the ~10 ns of a full normal) and leaves a copy of
// its state at every slice boundary; the slices are then drawn by ATen itself, in parallel.  Bit-identical to the one-call draw (test).
namespace {
struct Slice { int k; int64_t a, b; at::mt19937 eng; };

// Stepping at::mt19937 without producing outputs: its state after d draws is the seeded state after ceil(d / 624) block updates (the
// recurrence of MT19937RNGEngine.h's next_state, the standard MT19937 one) with next_ = r, left_ = 625 - r, r = (d - 1) % 624 + 1 draws
// taken from the current block.  The block update alone is a short vectorisable loop (0.3 ns per skipped draw instead of the 2-3 ns of
// operator(), which also tempers every output).
__attribute__((target_clones("avx512f", "avx2", "default"))) void mt_block_update(uint32_t* st) {   // (resolved once at load time for the host's widest vectors)
    constexpr int N = 624, M = 397;
    constexpr uint32_t A = 0x9908b0dfu, UM = 0x80000000u, LM = 0x7fffffffu;
    auto tw = [](uint32_t u, uint32_t v) { return (((u & UM) | (v & LM)) >> 1) ^ ((v & 1u) ? A : 0u); };
    const uint32_t first = st[0];
    for (int k = 0; k < N - M; ++k) st[k] = st[k + M] ^ tw(st[k], st[k + 1]);
    for (int k = N - M; k < N - 1; ++k) st[k] = st[k + M - N] ^ tw(st[k], st[k + 1]);
    (void)first;
    st[N - 1] = st[M - 1] ^ tw(st[N - 1], st[0]);   // (st[0] is already the new one, as in next_state)
}
// engine seeded with `seed`, positioned after `before` draws -> positioned after `after` >= before draws
inline void mt_advance(at::mt19937& eng, int64_t before, int64_t after) {
    if (after == before) return;
    at::mt19937_data_pod pod = eng.data();
    const int64_t blocks = (after + 623) / 624 - (before + 623) / 624;
    for (int64_t b = 0; b < blocks; ++b) mt_block_update(pod.state_.data());
    const int r = (int)((after - 1) % 624) + 1;
    pod.next_ = (uint32_t)r;
    pod.left_ = 625 - r;
    eng.set_data(pod);
}
}

extern "C" int gnnx_host_draw_masks(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* off, float* out, int32_t threads) {
    return gnnx_host_draw_masks_sliced(T, n, seeds, off, out, threads, (int64_t)1 << 21);
}

extern "C" int gnnx_host_draw_masks_sliced(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* off, float* out, int32_t threads,
                                           int64_t slice_values) {
    if (T < 0 || (T > 0 && (!n || !seeds || !off || !out))) {
        g_err = "null argument";
        return 1;
    }
    if (T == 0) return 0;
    threads = std::max(1, std::min<int32_t>(threads, 128));
    slice_values = std::max<int64_t>(16, slice_values / 16 * 16);
    // targets of more than two slices are cut; the others are dealt in equal shares of the VALUES (cut points in their prefix sums)
    std::vector<int> big, small;
    std::vector<int64_t> small_off(1, 0);
    for (int k = 0; k < T; ++k) {
        const int64_t nn = (int64_t)n[k] * n[k];
        if (threads > 1 && nn > 2 * slice_values) {
            big.push_back(k);
        } else {
            small.push_back(k);
            small_off.push_back(small_off.back() + nn);
        }
    }
    const int S = (int)small.size();
    const int small_parts = S ? (int)std::min<int64_t>(threads, S) : 0;
    std::vector<int> cut(small_parts + 1, S);
    if (small_parts) cut[0] = 0;
    for (int i = 1; i < small_parts; ++i) {
        const int64_t want = small_off[S] * i / small_parts;
        cut[i] = (int)(std::lower_bound(small_off.begin(), small_off.begin() + S, want) - small_off.begin());
    }
    std::vector<std::vector<Slice>> slices(big.size());
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto fail = [&](const std::exception& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        failed = true;
        err = e.what();
    };
    auto std_of = [&](int k) {   // nn.init.calculate_gain("relu") * math.sqrt(2.0 / (n + n)), evaluated in double like the Python expression
        return std::sqrt(2.0) * std::sqrt(2.0 / ((double)n[k] + (double)n[k]));
    };
    // phase 1: the walkers of the big targets (first: they are the long poles) and the small targets
    auto phase1 = [&](int part) {
        try {
            c10::InferenceMode ng;
            if (part < (int)big.size()) {
                const int k = big[part];
                const int64_t nn = (int64_t)n[k] * n[k];
                at::mt19937 eng((uint64_t)seeds[k]);    // == CPUGeneratorImpl::set_current_seed(seed)
                std::vector<Slice>& sl = slices[part];
                for (int64_t a = 0; a < nn; a += slice_values) {
                    const int64_t b = (nn - a < slice_values + 16) ? nn : a + slice_values;   // no last slice shorter than 16 values
                    sl.push_back(Slice{k, a, b, eng});
                    if (b == nn) break;
                    mt_advance(eng, a, b);
                }
                return;
            }
            const int sp = part - (int)big.size();
            at::Generator gen = at::detail::createCPUGenerator(0);
            for (int q = cut[sp]; q < cut[sp + 1]; ++q) {
                const int k = small[q];
                const int64_t nn = (int64_t)n[k] * n[k];
                if (nn == 0) continue;
                gen.set_current_seed((uint64_t)seeds[k]);   // == torch.manual_seed(seed) on the default generator: fresh mt19937, no cached normal
                at::Tensor view = at::from_blob(out + off[k], {nn}, at::TensorOptions().dtype(at::kFloat));
                view.normal_(1.0, std_of(k), gen);
            }
        } catch (const std::exception& e) {
            fail(e);
        }
    };
    std::vector<const Slice*> all;
    auto phase2 = [&](int part) {
        try {
            c10::InferenceMode ng;
            const Slice& s = *all[part];
            at::Generator gen = at::detail::createCPUGenerator(0);
            auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
            impl->set_engine(s.eng);
            impl->set_next_float_normal_sample(std::optional<float>());
            at::Tensor view = at::from_blob(out + off[s.k] + s.a, {s.b - s.a}, at::TensorOptions().dtype(at::kFloat));
            view.normal_(1.0, std_of(s.k), gen);
        } catch (const std::exception& e) {
            fail(e);
        }
    };
    const int parts1 = (int)big.size() + small_parts;
    if (parts1 == 1 && big.empty()) {
        phase1(0);
    } else {
        std::lock_guard<std::mutex> lk(g_pool_mu);   // one batch at a time draws on the pool
        if (!g_pool || g_pool->size() < threads) g_pool = new Pool(std::max<int>(threads, 16));   // (an outgrown pool is leaked on purpose: its threads sleep)
        // (limit = the caller's `threads`: the pool may hold more workers than this call was given - the ranks of a node share its cores)
        if (parts1) g_pool->run(parts1, phase1, threads);
        for (auto& sl : slices)
            for (auto& x : sl) all.push_back(&x);
        if (!failed && !all.empty()) g_pool->run((int)all.size(), phase2, threads);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}

// The edge-sparse kernels read the initial mask only on the EDGES of a sub-graph (the other n^2 - 2E entries of construct_edge_mask's
// draw never reach an output of the reference, explain.py:665-678).  The values on the edges are positions of ONE mt19937 stream per
// target, so the engine has to pass over the whole stream - but only its STATE: ATen's normal_ (normal_fill, see above) is one engine
// draw per value followed by a Box-Muller transform of 16 values at a time, and 624 = 39 x 16, so a 16-value block never straddles two
// states of the engine.  Per target (per chunk of a large one) a walker regenerates the raw state block by block (0.3 ns per draw: no
// tempering, no uniform, no logarithm / sine / cosine), copies the 16 raw words of every block that holds an edge entry into a staging
// state, and lets ATEN ITSELF temper and transform 38 such blocks per call: an engine whose state array holds the staged words yields
// exactly the draws of those blocks, so `normal_` of 16 x 38 values from it IS the reference's arithmetic on them - bit-identical to the
// full draw by construction (tests/test_host_api.py), whatever vector path ATen takes on the host.  A ragged stream (n^2 not a multiple
// of 16) has its last 16 values redrawn from the 16 draws that follow the fill: the same staging, from those words.  Streams of fewer
// than 16 values take ATen's scalar path as a whole.  For the 16 384-target BA-House x100k set 6.6 % of the 62.5 M blocks hold an edge
// entry: the draw that bounded its batches (1.0e9 normals, 4.6 ns of one core each) becomes 1.0e9 block-update steps + 4 M staged blocks.
// rc [E][2] int32: the edges of all targets, target after target (eoff [T + 1]), r < c, local node ids; out[e] = (M[r][c], M[c][r]).
namespace {
constexpr int MTN = 624;
constexpr int STAGE_SLOTS = 38;     // 16-word blocks per staged engine state: words 1 .. 608 (next_ = 1, left_ = 624: a valid engine that
                                    // yields 623 draws without regenerating)
inline void mt_seed_state(uint32_t* st, uint64_t seed) {   // at::mt19937::init_with_uint32
    st[0] = (uint32_t)(seed & 0xffffffffu);
    for (int j = 1; j < MTN; ++j) st[j] = 1812433253u * (st[j - 1] ^ (st[j - 1] >> 30)) + (uint32_t)j;
}
struct EdgeTarget {                 // one target of the edge draw
    int k = 0;
    int64_t nn = 0, reg_end = 0;    // values; positions >= reg_end come from the redrawn last 16 values (reg_end = nn when nn % 16 == 0)
    std::vector<std::pair<int64_t, int64_t>> pos;   // (position in the n x n stream, index into out), ascending
};
struct EdgeChunk { int ti; int64_t b0, b1; std::vector<uint32_t> start; };   // state blocks [b0, b1); `start` = the raw state before block b0's update
struct Stager {
    at::Generator gen = at::detail::createCPUGenerator(0);
    at::CPUGeneratorImpl* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
    uint32_t words[1 + 16 * STAGE_SLOTS];
    float vals[16 * STAGE_SLOTS];
    int pairs = 0;                                  // staged Box-Muller pairs: pair s sits in lanes (s % 8, s % 8 + 8) of synthetic block s / 8
    std::vector<std::pair<int, int64_t>> wants;     // (index into vals, index into out)
    double std_ = 1.0;
    float* out = nullptr;
    // The transform of a 16-value block works on the eight pairs (j, j + 8) independently - values j and j + 8 come from the uniforms j and
    // j + 8 alone - and every lane computes the same function, so only the PAIR that holds a wanted value is staged, eight pairs of any
    // blocks to a synthetic block: an eighth of the logarithms / sines / cosines of staging whole blocks.  (The bit-comparison against the
    // full draw in tests/test_host_api.py covers every lane position.)
    int add_pair(uint32_t a, uint32_t b) {
        uint32_t* w = words + 1 + 16 * (pairs >> 3) + (pairs & 7);
        w[0] = a;
        w[8] = b;
        return pairs++;
    }
    bool full() const { return pairs == 8 * STAGE_SLOTS; }
    // whole-block staging (the fallback when the host's normal_ should ever treat lanes differently - see pair_staging_ok): the eight pairs of a
    // real block in their own lanes of one synthetic block; returns the synthetic block's first value index
    int add_block(const uint32_t* w16) {
        while (pairs & 7) add_pair(0u, 0u);
        if (full()) flush();
        const int base = 16 * (pairs >> 3);
        for (int j = 0; j < 8; ++j) add_pair(w16[j], w16[j + 8]);
        return base;
    }
    static int value_index(int pair, bool second) { return 16 * (pair >> 3) + (pair & 7) + (second ? 8 : 0); }
    void flush() {
        if (!pairs) return;
        const int blocks = (pairs + 7) >> 3;
        for (int s = pairs; s < 8 * blocks; ++s) {   // (unused lanes of the last synthetic block: any defined words - u = 0 gives radius 0)
            words[1 + 16 * (s >> 3) + (s & 7)] = 0u;
            words[1 + 16 * (s >> 3) + (s & 7) + 8] = 0u;
        }
        at::mt19937 eng;
        at::mt19937_data_pod pod = eng.data();
        pod.seeded_ = true;
        pod.next_ = 1;
        pod.left_ = MTN;
        std::memcpy(pod.state_.data() + 1, words + 1, sizeof(uint32_t) * 16 * (size_t)blocks);
        eng.set_data(pod);
        impl->set_engine(eng);
        impl->set_next_float_normal_sample(std::optional<float>());
        at::Tensor view = at::from_blob(vals, {(int64_t)16 * blocks}, at::TensorOptions().dtype(at::kFloat));
        view.normal_(1.0, std_, gen);       // 16 * blocks >= 16 values, a multiple of 16: normal_fill, no redraw of a ragged tail
        for (const auto& w : wants) out[w.second] = vals[w.first];
        wants.clear();
        pairs = 0;
    }
};
// per worker thread: one staging engine (creating a generator per target cost as much as a small target's draw) and the merge scratch
Stager& thread_stager() {
    thread_local Stager sg;
    return sg;
}
// Pair staging rests on one property of the host's normal_ beyond its documented structure: every lane of the 16-value transform computes
// the same function of its own pair of uniforms.  Checked once per process on a fixed stream (pairs staged into OTHER lanes than their own
// must reproduce ATen's one-call draw bit for bit); if it ever fails, whole blocks are staged in their own lanes instead (slower, no assumption).
bool pair_staging_ok() {
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [] {
        try {
            c10::InferenceMode ng;
            constexpr int NV = 96;
            float ref[NV];
            at::Generator gen = at::detail::createCPUGenerator(0);
            gen.set_current_seed(20240923u);
            at::Tensor view = at::from_blob(ref, {(int64_t)NV}, at::TensorOptions().dtype(at::kFloat));
            view.normal_(1.0, 0.25, gen);
            uint32_t st[MTN];
            mt_seed_state(st, 20240923u);
            mt_block_update(st);                       // the words of draws 0 .. 623
            Stager sg;
            float got[NV];
            sg.std_ = 0.25;
            sg.out = got;
            for (int k = 0; k < NV; ++k) {             // positions in a scrambled order: a pair lands in another lane than its own
                const int p = (k * 37 + 11) % NV, q = p / 16, lane = p % 16, j = lane & 7;
                if (sg.full()) sg.flush();
                sg.wants.emplace_back(Stager::value_index(sg.add_pair(st[16 * q + j], st[16 * q + j + 8]), lane >= 8), (int64_t)p);
            }
            sg.flush();
            ok = std::memcmp(ref, got, sizeof ref) == 0;
        } catch (...) {
            ok = false;
        }
    });
    return ok;
}
std::vector<std::pair<int64_t, int64_t>>& scratch_up() {
    thread_local std::vector<std::pair<int64_t, int64_t>> v;
    return v;
}
std::vector<std::pair<int64_t, int64_t>>& scratch_lo() {
    thread_local std::vector<std::pair<int64_t, int64_t>> v;
    return v;
}
std::vector<int64_t>& scratch_cnt() {
    thread_local std::vector<int64_t> v;
    return v;
}
}  // namespace

extern "C" int gnnx_host_draw_edge_masks(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* eoff, const int32_t* rc, float* out,
                                         int32_t threads, int64_t slice_values) {
    if (T < 0 || (T > 0 && (!n || !seeds || !eoff || !rc || !out))) {
        g_err = "null argument";
        return 1;
    }
    if (T == 0) return 0;
    threads = std::max(1, std::min<int32_t>(threads, 128));
    const int64_t CHB = std::max<int64_t>(1, 32 * std::max<int64_t>(1024, slice_values) / MTN);   // engine state blocks per work item
    static const bool force_blocks = std::getenv("GNNX_HOST_STAGE_BLOCKS") != nullptr;             // (test knob: exercise the fallback)
    const bool by_pairs = !force_blocks && pair_staging_ok();
    std::vector<EdgeTarget> tg;
    tg.reserve(T);
    for (int k = 0; k < T; ++k) {
        const int64_t nn = (int64_t)n[k] * n[k];
        if (nn == 0 || eoff[k + 1] == eoff[k]) continue;                         // no edge: nothing of this target's stream is needed
        EdgeTarget t;
        t.k = k;
        t.nn = nn;
        t.reg_end = (nn % 16 == 0) ? nn : nn - 16;
        tg.push_back(std::move(t));
    }
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto fail = [&](const std::exception& e) {
        std::lock_guard<std::mutex> lk(err_mu);
        failed = true;
        err = e.what();
    };
    auto std_of = [&](int k) { return std::sqrt(2.0) * std::sqrt(2.0 / ((double)n[k] + (double)n[k])); };
    // the blocks [b0, b1) of one target from the raw state `st` (the state BEFORE block b0's update); `last`: this call also serves the
    // redrawn tail of a ragged stream
    // (b0 == b1: the tail alone, from the state a walker left behind the last regular block)
    auto run_blocks = [&](const EdgeTarget& t, uint32_t* st, int64_t b0, int64_t b1, bool last, Stager& sg) {
        sg.std_ = std_of(t.k);
        sg.out = out;
        auto it = std::lower_bound(t.pos.begin(), t.pos.end(), std::make_pair(b0 == b1 ? t.reg_end : b0 * MTN, (int64_t)-1));
        const auto end = t.pos.end();
        int64_t b = b0;
        for (; b < b1; ++b) {
            mt_block_update(st);                                                 // st = the words of draws [624 b, 624 b + 624)
            const int64_t lim = std::min<int64_t>((b + 1) * MTN, t.reg_end);
            int64_t last_key = -1;                                               // (block, pair) of the pair staged last: entries of one pair share it
            int last_pair = 0;
            while (!by_pairs && it != end && it->first < lim) {                  // fallback: whole blocks in their own lanes
                const int64_t q16 = it->first / 16;
                const int base = sg.add_block(st + (16 * q16 - b * MTN));
                for (; it != end && it->first < 16 * q16 + 16 && it->first < t.reg_end; ++it)
                    sg.wants.emplace_back(base + (int)(it->first - 16 * q16), it->second);
            }
            while (it != end && it->first < lim) {
                const int64_t q16 = it->first / 16;
                const int lane = (int)(it->first - 16 * q16), j = lane & 7;
                if (8 * q16 + j != last_key) {
                    if (sg.full()) sg.flush();
                    const uint32_t* w = st + (16 * q16 - b * MTN);
                    last_pair = sg.add_pair(w[j], w[j + 8]);
                    last_key = 8 * q16 + j;
                }
                sg.wants.emplace_back(Stager::value_index(last_pair, lane >= 8), it->second);
                ++it;
            }
        }
        if (last && t.reg_end != t.nn && it != end) {                            // entries among the last 16 values of a ragged stream: draws [nn, nn + 16)
            const int64_t bt = t.nn / MTN;
            for (; b <= bt; ++b) mt_block_update(st);                            // st = block bt (b1 - 1 <= bt always)
            uint32_t w16[16];
            const int w0 = (int)(t.nn % MTN), first = std::min(16, MTN - w0);
            std::memcpy(w16, st + w0, sizeof(uint32_t) * first);
            if (first < 16) {
                mt_block_update(st);
                std::memcpy(w16 + first, st, sizeof(uint32_t) * (16 - first));
            }
            if (!by_pairs) {
                const int base = sg.add_block(w16);
                for (; it != end; ++it) sg.wants.emplace_back(base + (int)(it->first - (t.nn - 16)), it->second);
            }
            for (; it != end; ++it) {
                const int lane = (int)(it->first - (t.nn - 16)), j = lane & 7;
                if (sg.full()) sg.flush();
                sg.wants.emplace_back(Stager::value_index(sg.add_pair(w16[j], w16[j + 8]), lane >= 8), it->second);
            }
        }
        sg.flush();
    };
    std::vector<EdgeChunk> chunks;     // the chunks of the large targets (filled by their walkers in phase 1)
    std::mutex chunks_mu;
    // phase 1, one work item per target: build its sorted positions; a stream of fewer than 16 values as a whole (ATen's scalar path); a target
    // of up to CHB blocks entirely; a larger one only WALKS here and leaves the raw state at every chunk boundary for phase 2
    auto phase1 = [&](int ti) {
        try {
            c10::InferenceMode ng;
            EdgeTarget& t = tg[ti];
            const int k = t.k;
            const int64_t nk = n[k];
            // Positions in ascending order without a sort when the caller's list is row-major (what gnnx_edge_layout / gnnx_gather_edges emit):
            // the entries (r, c) then ascend as they come, and the mirrored ones (c, r) ascend after a stable bucketing by column - two
            // sorted runs, merged.  Any other order falls back to the sort.
            const int64_t e0 = eoff[k], ne = eoff[k + 1] - eoff[k];
            for (int64_t e = e0; e < e0 + ne; ++e)      // the ids index this target's n x n stream (and the bucket table below): check them
                if (rc[2 * e] < 0 || rc[2 * e] >= nk || rc[2 * e + 1] < 0 || rc[2 * e + 1] >= nk)
                    throw std::out_of_range("gnnx_host_draw_edge_masks: edge (r, c) outside its target's n x n block");
            bool row_major = true;
            for (int64_t e = e0 + 1; e < e0 + ne && row_major; ++e)
                row_major = (int64_t)rc[2 * e - 2] * nk + rc[2 * e - 1] < (int64_t)rc[2 * e] * nk + rc[2 * e + 1];
            t.pos.resize(2 * (size_t)ne);
            if (row_major) {
                std::vector<std::pair<int64_t, int64_t>>& up = scratch_up();
                std::vector<std::pair<int64_t, int64_t>>& lo = scratch_lo();
                std::vector<int64_t>& cnt = scratch_cnt();
                up.resize((size_t)ne);
                lo.resize((size_t)ne);
                cnt.assign((size_t)nk + 1, 0);
                for (int64_t e = e0; e < e0 + ne; ++e) ++cnt[(size_t)rc[2 * e + 1] + 1];
                for (int64_t c = 0; c < nk; ++c) cnt[(size_t)c + 1] += cnt[(size_t)c];
                for (int64_t e = e0; e < e0 + ne; ++e) {
                    const int64_t r = rc[2 * e], cc = rc[2 * e + 1];
                    up[(size_t)(e - e0)] = {r * nk + cc, 2 * e};
                    lo[(size_t)cnt[(size_t)cc]++] = {cc * nk + r, 2 * e + 1};
                }
                std::merge(up.begin(), up.end(), lo.begin(), lo.end(), t.pos.begin());
            } else {
                for (int64_t e = e0; e < e0 + ne; ++e) {
                    const int64_t r = rc[2 * e], cc = rc[2 * e + 1];
                    t.pos[2 * (size_t)(e - e0)] = {r * nk + cc, 2 * e};
                    t.pos[2 * (size_t)(e - e0) + 1] = {cc * nk + r, 2 * e + 1};
                }
                std::sort(t.pos.begin(), t.pos.end());
            }
            if (t.nn < 16) {
                at::Generator& gen = thread_stager().gen;
                gen.set_current_seed((uint64_t)seeds[k]);   // (fresh mt19937, no cached normal - as torch.manual_seed)
                float buf[16];
                at::Tensor view = at::from_blob(buf, {t.nn}, at::TensorOptions().dtype(at::kFloat));
                view.normal_(1.0, std_of(k), gen);
                for (const auto& pr : t.pos) out[pr.second] = buf[pr.first];
                return;
            }
            const bool tail = t.reg_end != t.nn && t.pos.back().first >= t.reg_end;
            int64_t last_reg = -1;                                               // last wanted position in the regular part
            for (auto it = t.pos.rbegin(); it != t.pos.rend(); ++it)
                if (it->first < t.reg_end) {
                    last_reg = it->first;
                    break;
                }
            const int64_t nblk = last_reg >= 0 ? last_reg / MTN + 1 : 0;         // state blocks the regular entries need
            thread_local std::vector<uint32_t> st;
            st.resize(MTN);
            mt_seed_state(st.data(), (uint64_t)seeds[k]);
            if (nblk <= CHB) {
                run_blocks(t, st.data(), 0, nblk, true, thread_stager());
                return;
            }
            std::vector<EdgeChunk> mine;
            for (int64_t b0 = 0; b0 < nblk; b0 += CHB) {
                const int64_t b1 = std::min(nblk, b0 + CHB);
                mine.push_back(EdgeChunk{ti, b0, b1, st});
                if (b1 < nblk || tail)                                           // (the last chunk's own pass ends the walk unless the tail needs a start of its own)
                    for (int64_t b = b0; b < b1; ++b) mt_block_update(st.data());
            }
            if (tail) mine.push_back(EdgeChunk{ti, nblk, nblk, st});             // an empty block range: run_blocks goes straight to the tail
            std::lock_guard<std::mutex> lk(chunks_mu);
            for (auto& c : mine) chunks.push_back(std::move(c));
        } catch (const std::exception& e) {
            fail(e);
        }
    };
    auto phase2 = [&](int ci) {
        try {
            c10::InferenceMode ng;
            EdgeChunk& c = chunks[ci];
            // (a regular chunk stops at its own last block - the entries of later blocks belong to later chunks; the tail chunk is the empty range)
            run_blocks(tg[c.ti], c.start.data(), c.b0, c.b1, c.b0 == c.b1, thread_stager());
        } catch (const std::exception& e) {
            fail(e);
        }
    };
    const int nt = (int)tg.size();
    if (nt == 0) return 0;
    if (threads == 1) {
        for (int i = 0; i < nt; ++i) phase1(i);
        for (int i = 0; i < (int)chunks.size() && !failed; ++i) phase2(i);
    } else {
        // the largest targets first (their walkers are the long poles), then dynamic hand-out
        std::vector<int> order(nt);
        for (int i = 0; i < nt; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tg[a].nn > tg[b].nn; });
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool || g_pool->size() < threads) g_pool = new Pool(std::max<int>(threads, 16));
        g_pool->run(nt, [&](int i) { phase1(order[i]); }, threads);
        if (!failed && !chunks.empty()) g_pool->run((int)chunks.size(), phase2, threads);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}

extern "C" int gnnx_host_pair_staging_ok(void) { return pair_staging_ok() ? 1 : 0; }

// The values on the edges from the raw engine words the device picked (gnnx_mt_edge_words: k_mt_stream walks every target's mt19937 on the
// GPU, k_mt_gather_pairs picks the two words of each entry's Box-Muller pair).  What is left for the host is what only the host can do
// bit for bit - ATen's tempering + transform of exactly those pairs: the pair staging of gnnx_host_draw_edge_masks without its walk, O(E)
// instead of O(sum n^2).  Work items = runs of targets of about 64 K entries; one Stager flush per target (the standard deviation is the
// target's).  Targets of fewer than 16 values are drawn whole from their seed (ATen's scalar path, as there).
extern "C" int gnnx_host_transform_edge_words(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* eoff, const int32_t* rc,
                                              const uint32_t* words, float* out, int32_t threads) {
    if (T < 0 || (T > 0 && (!n || !seeds || !eoff || !rc || !words || !out))) {
        g_err = "null argument";
        return 1;
    }
    if (T == 0) return 0;
    if (!pair_staging_ok()) {
        g_err = "the host's normal_ does not have the pair-staging property: use gnnx_host_draw_edge_masks";
        return 1;
    }
    threads = std::max(1, std::min<int32_t>(threads, 128));
    std::vector<std::pair<int, int>> items;      // [first target, last target) per work item
    {
        const int64_t per = 65536;
        int a = 0;
        while (a < T) {
            int b = a + 1;
            while (b < T && eoff[b] - eoff[a] < per) ++b;
            items.emplace_back(a, b);
            a = b;
        }
    }
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto work = [&](int it) {
        try {
            c10::InferenceMode ng;
            Stager& sg = thread_stager();
            sg.out = out;
            sg.pairs = 0;          // (an exception in an earlier call on this thread may have left pairs staged for ANOTHER out buffer: ADVICE r5)
            sg.wants.clear();
            for (int k = items[it].first; k < items[it].second; ++k) {
                const int64_t nk = n[k], nn = nk * nk, e0 = eoff[k], e1 = eoff[k + 1];
                if (nn == 0 || e1 == e0) continue;
                // a target's edge ids are validated BEFORE anything is staged (as gnnx_host_draw_edge_masks does): nothing half-staged survives a throw
                for (int64_t e = e0; e < e1; ++e) {
                    const int64_t r = rc[2 * e], c = rc[2 * e + 1];
                    if (r < 0 || r >= nk || c < 0 || c >= nk) throw std::out_of_range("gnnx_host_transform_edge_words: edge outside its target's block");
                }
                sg.std_ = std::sqrt(2.0) * std::sqrt(2.0 / ((double)nk + (double)nk));
                if (nn < 16) {      // ATen's scalar path on the whole stream
                    at::Generator& gen = sg.gen;
                    gen.set_current_seed((uint64_t)seeds[k]);
                    float buf[16];
                    at::Tensor view = at::from_blob(buf, {nn}, at::TensorOptions().dtype(at::kFloat));
                    view.normal_(1.0, sg.std_, gen);
                    for (int64_t e = e0; e < e1; ++e) {
                        const int64_t r = rc[2 * e], c = rc[2 * e + 1];
                        if (r < 0 || r >= nk || c < 0 || c >= nk) throw std::out_of_range("gnnx_host_transform_edge_words: edge outside its target's block");
                        out[2 * e] = buf[r * nk + c];
                        out[2 * e + 1] = buf[c * nk + r];
                    }
                    continue;
                }
                const int64_t reg_end = (nn % 16 == 0) ? nn : nn - 16;
                for (int64_t e = e0; e < e1; ++e) {
                    const int64_t r = rc[2 * e], c = rc[2 * e + 1];
                    if (r < 0 || r >= nk || c < 0 || c >= nk) throw std::out_of_range("gnnx_host_transform_edge_words: edge outside its target's block");
                    for (int dir = 0; dir < 2; ++dir) {
                        const int64_t p = dir ? c * nk + r : r * nk + c;
                        const int lane = (int)(p < reg_end ? p % 16 : p - (nn - 16));
                        if (sg.full()) sg.flush();
                        const int pr = sg.add_pair(words[4 * e + 2 * dir], words[4 * e + 2 * dir + 1]);
                        sg.wants.emplace_back(Stager::value_index(pr, lane >= 8), 2 * e + dir);
                    }
                }
                sg.flush();      // (the next target has another standard deviation)
            }
        } catch (const std::exception& e) {
            Stager& sg = thread_stager();
            sg.pairs = 0;
            sg.wants.clear();
            std::lock_guard<std::mutex> lk(err_mu);
            failed = true;
            err = e.what();
        }
    };
    if (threads == 1 || items.size() == 1) {
        for (int i = 0; i < (int)items.size(); ++i) work(i);
    } else {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool || g_pool->size() < threads) g_pool = new Pool(std::max<int>(threads, 16));
        g_pool->run((int)items.size(), work, threads);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}
