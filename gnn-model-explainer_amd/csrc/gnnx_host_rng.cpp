// gnnx_host_rng.cpp — include/gnnx_host.h: the seeded initial masks of a batch, drawn by ATen's own CPU normal_ from C++ threads.
// Replaces, for a whole batch, the per-target `mask.normal_(1.0, std)` of construct_edge_mask (explainer/explain.py:645-652).
#include <ATen/ATen.h>
#include <ATen/CPUGeneratorImpl.h>
#include <c10/core/InferenceMode.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
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
    void run(int parts, const std::function<void(int)>& fn) {
        std::unique_lock<std::mutex> lk(mu_);
        fn_ = &fn;
        parts_ = parts;
        next_ = 0;
        done_ = 0;
        ++epoch_;
        cv_.notify_all();
        cv_done_.wait(lk, [&] { return done_ == parts_; });
        fn_ = nullptr;
    }

  private:
    void loop(int) {
        c10::InferenceMode ng;
        unsigned long long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return epoch_ != seen; });
            seen = epoch_;
            while (next_ < parts_) {
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
    int parts_ = 0, next_ = 0, done_ = 0;
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
inline void mt_block_update(uint32_t* st) {
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
        if (parts1) g_pool->run(parts1, phase1);
        for (auto& sl : slices)
            for (auto& x : sl) all.push_back(&x);
        if (!failed && !all.empty()) g_pool->run((int)all.size(), phase2);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}

// The edge-sparse kernels read the initial mask only on the EDGES of a sub-graph (the other n^2 - 2E entries of construct_edge_mask's
// draw never reach an output of the reference, explain.py:665-678), but the values on the edges are positions of ONE mt19937 stream
// per target, so the whole stream still has to be generated - what need not happen is writing it: the full draw of the 16 384-target
// BA-House x100k set is 4 GB through the host's memory system (94-140 ms on 32 threads, SLOWER on more: tools/probe_rng_big.py), plus a
// 4 GB H2D copy and a scatter kernel.  Here every thread draws its part of the stream slice by slice into a cache-resident buffer (the
// same ATen normal_ calls on the same engine states as gnnx_host_draw_masks_sliced: bit-identical values) and keeps the two values of
// every edge, out[e] = (M[r][c], M[c][r]) for edge e = (r, c) of the target's upper-triangle edge list - 12 MB instead of 4 GB leave
// the host, and the draw scales with the cores again.
// rc [E][2] int32: the edges of all targets, target after target (eoff [T + 1]), r < c, local node ids.  A chunk of a large target starts
// from the seeded engine advanced by its first value's index (mt_advance: 0.3 ns per skipped draw).
extern "C" int gnnx_host_draw_edge_masks(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* eoff, const int32_t* rc, float* out,
                                         int32_t threads, int64_t slice_values) {
    if (T < 0 || (T > 0 && (!n || !seeds || !eoff || !rc || !out))) {
        g_err = "null argument";
        return 1;
    }
    if (T == 0) return 0;
    threads = std::max(1, std::min<int32_t>(threads, 128));
    const int64_t L = std::max<int64_t>(1024, slice_values / 16 * 16);        // values per normal_ call (cache-resident buffer)
    const int64_t CH = 32 * L;                                                   // values per work item
    struct Chunk { int k; int64_t a, b; };
    std::vector<Chunk> chunks;
    for (int k = 0; k < T; ++k) {
        const int64_t nn = (int64_t)n[k] * n[k];
        if (nn == 0 || eoff[k + 1] == eoff[k]) continue;                         // no edge: nothing of this target's stream is needed
        for (int64_t a = 0; a < nn; a += CH) {
            const int64_t b = (nn - a < CH + 16) ? nn : a + CH;
            chunks.push_back(Chunk{k, a, b});
            if (b == nn) break;
        }
    }
    // largest work items first, then dynamic hand-out
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& x, const Chunk& y) { return (x.b - x.a) > (y.b - y.a); });
    // per target: (position in the n x n stream, index into out) of its 2E entries, sorted by position - built by the first chunk that needs it
    std::vector<std::vector<std::pair<int64_t, int64_t>>> pos(T);
    std::vector<std::once_flag> built(T);
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto work = [&](int ci) {
        try {
            c10::InferenceMode ng;
            const Chunk c = chunks[ci];
            const int k = c.k;
            const int64_t nk = n[k], nn = nk * nk;
            std::call_once(built[k], [&] {
                auto& p = pos[k];
                p.reserve(2 * (size_t)(eoff[k + 1] - eoff[k]));
                for (int64_t e = eoff[k]; e < eoff[k + 1]; ++e) {
                    const int64_t r = rc[2 * e], cc = rc[2 * e + 1];
                    p.emplace_back(r * nk + cc, 2 * e);
                    p.emplace_back(cc * nk + r, 2 * e + 1);
                }
                std::sort(p.begin(), p.end());
            });
            const auto& p = pos[k];
            size_t it = std::lower_bound(p.begin(), p.end(), std::make_pair(c.a, (int64_t)-1)) - p.begin();
            if (it == p.size() || p[it].first >= c.b) return;                     // no edge entry in this chunk: its values are not needed
            at::Generator gen = at::detail::createCPUGenerator(0);
            gen.set_current_seed((uint64_t)seeds[k]);
            auto* impl = at::check_generator<at::CPUGeneratorImpl>(gen);
            const double std_ = std::sqrt(2.0) * std::sqrt(2.0 / ((double)nk + (double)nk));
            std::vector<float> buf((size_t)(L + 16));
            int64_t at_draw = 0;      // draws the engine has made
            for (int64_t s = c.a; s < c.b;) {
                int64_t t = (c.b - s < L + 16) ? c.b : s + L;                     // (no last call shorter than 16 values: ATen redraws the last 16 of a ragged tensor)
                if (it < p.size() && p[it].first >= t && t < c.b) {              // nothing wanted in [s, t): skip it without drawing
                    s = t;
                    continue;
                }
                if (at_draw != s) {
                    at::mt19937 eng((uint64_t)seeds[k]);
                    mt_advance(eng, 0, s);
                    impl->set_engine(eng);
                    impl->set_next_float_normal_sample(std::optional<float>());
                    at_draw = s;
                }
                at::Tensor view = at::from_blob(buf.data(), {t - s}, at::TensorOptions().dtype(at::kFloat));
                view.normal_(1.0, std_, gen);
                at_draw = t;                                                      // (a ragged last call draws 16 more, but it is the last of its target)
                for (; it < p.size() && p[it].first < t; ++it) out[p[it].second] = buf[(size_t)(p[it].first - s)];
                s = t;
                if (it == p.size() || p[it].first >= c.b) break;
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(err_mu);
            failed = true;
            err = e.what();
        }
    };
    if (chunks.size() == 1 || threads == 1) {
        for (int ci = 0; ci < (int)chunks.size(); ++ci) work(ci);
    } else if (!chunks.empty()) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool || g_pool->size() < threads) g_pool = new Pool(std::max<int>(threads, 16));
        g_pool->run((int)chunks.size(), work);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}
