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

extern "C" int gnnx_host_draw_masks(int32_t T, const int32_t* n, const int64_t* seeds, const int64_t* off, float* out, int32_t threads) {
    if (T < 0 || (T > 0 && (!n || !seeds || !off || !out))) {
        g_err = "null argument";
        return 1;
    }
    if (T == 0) return 0;
    threads = std::max(1, std::min<int32_t>(threads, 64));
    // equal shares of the VALUES (not of the targets): cut points in the prefix sums
    const int64_t total = off[T - 1] + (int64_t)n[T - 1] * n[T - 1];
    const int parts = (int)std::min<int64_t>(threads, T);
    std::vector<int> cut(parts + 1, T);
    cut[0] = 0;
    for (int i = 1; i < parts; ++i) {
        const int64_t want = total * i / parts;
        cut[i] = (int)(std::lower_bound(off, off + T, want) - off);
    }
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mu;
    auto work = [&](int part) {
        try {
            c10::InferenceMode ng;
            at::Generator gen = at::detail::createCPUGenerator(0);
            for (int k = cut[part]; k < cut[part + 1]; ++k) {
                const int64_t nn = (int64_t)n[k] * n[k];
                if (nn == 0) continue;
                gen.set_current_seed((uint64_t)seeds[k]);   // == torch.manual_seed(seed) on the default generator: fresh mt19937, no cached normal
                // nn.init.calculate_gain("relu") * math.sqrt(2.0 / (n + n)), evaluated in double like the Python expression
                const double std_ = std::sqrt(2.0) * std::sqrt(2.0 / ((double)n[k] + (double)n[k]));
                at::Tensor view = at::from_blob(out + off[k], {nn}, at::TensorOptions().dtype(at::kFloat));
                view.normal_(1.0, std_, gen);
            }
        } catch (const std::exception& e) {
            std::lock_guard<std::mutex> lk(err_mu);
            failed = true;
            err = e.what();
        }
    };
    if (parts == 1) {
        work(0);
    } else {
        std::lock_guard<std::mutex> lk(g_pool_mu);   // one batch at a time draws on the pool
        if (!g_pool || g_pool->size() < parts) g_pool = new Pool(std::max(parts, 16));   // (an outgrown pool is leaked on purpose: its threads sleep)
        g_pool->run(parts, work);
    }
    if (failed) {
        g_err = err;
        return 1;
    }
    return 0;
}
