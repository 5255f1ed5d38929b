// gnnx_capi.hip — C ABI (include/gnnx.h) over the gfx950 kernels of gnnx_kernels.hpp.
// Host side only: plan (tile tables, packed model block), workspace carving, the launch sequence of
// one mask-optimisation job and its optional hipGraph capture.  No compute happens on the host.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <mutex>
#include <vector>

constexpr int MAX_DEVICES_POOL = 16;

#include "../../include/gnnx.h"
#include "gnnx_kernels.hpp"
#include "gnnx_resident.hpp"
#include "gnnx_sparse.hpp"
#include "gnnx_sparse_large.hpp"
#include "gnnx_graph.hpp"
#include "gnnx_att.hpp"

using namespace gnnx;

static thread_local std::string g_err;
static int fail(const std::string& m) {
    g_err = m;
    return 1;
}
#define HIPCK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
            return fail(std::string(#x) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
    } while (0)

// ---- device-memory pool for the library's own tables ----------------------------------------------------------------------
// A plan owns a dozen small device tables (tile tables, target meta, id lists, the CSR of the large-class targets, the Adam
// scalar table).  hipFree synchronises the whole device and hipMalloc takes the driver's allocation lock: with one plan per
// batch both sat on the critical path of a pipelined job (the next batch's plan is built while the previous batch's loop is
// running - a hipFree there waits for that loop).  Blocks are therefore recycled: size classes of powers of two (8 MB steps
// above 64 MB), per device, never returned to the driver before gnnx_pool_trim().  A block handed back by gnnx_destroy must
// not be in use by work still in flight - the same contract hipFree imposed by synchronising.
#include <map>
#include <unordered_map>
namespace {
struct DevPool {
    std::mutex mu;
    std::unordered_map<void*, size_t> bucket_of;          // every block this pool ever allocated -> its size class
    std::map<size_t, std::vector<void*>> free_blocks;     // size class -> idle blocks
    size_t cached_bytes = 0;
};
DevPool g_pool[MAX_DEVICES_POOL];
inline size_t pool_bucket(size_t bytes) {
    if (bytes <= 256) return 256;
    if (bytes > (size_t(64) << 20)) return (bytes + (size_t(8) << 20) - 1) / (size_t(8) << 20) * (size_t(8) << 20);
    size_t b = 256;
    while (b < bytes) b <<= 1;
    return b;
}
inline DevPool& pool_here() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES_POOL) dev = 0;
    return g_pool[dev];
}
}  // namespace
template <class T>
static hipError_t pool_malloc(T** out, size_t bytes) {
    DevPool& P = pool_here();
    const size_t b = pool_bucket(bytes);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.free_blocks.find(b);
        if (it != P.free_blocks.end() && !it->second.empty()) {
            *out = static_cast<T*>(it->second.back());
            it->second.pop_back();
            P.cached_bytes -= b;
            return hipSuccess;
        }
    }
    void* ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, b);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        P.bucket_of[ptr] = b;
    }
    *out = static_cast<T*>(ptr);
    return hipSuccess;
}
static hipError_t pool_free(void* ptr) {
    if (!ptr) return hipSuccess;
    DevPool& P = pool_here();
    std::lock_guard<std::mutex> lk(P.mu);
    auto it = P.bucket_of.find(ptr);
    if (it == P.bucket_of.end()) return hipFree(ptr);   // not ours (cannot happen; stay correct)
    P.free_blocks[it->second].push_back(ptr);
    P.cached_bytes += it->second;
    return hipSuccess;
}
// Pinned host staging for the table uploads (BlockBuilder::commit): an upload from pinned memory is only ENQUEUED, so the thread that
// builds a plan does not wait for a copy kernel to find a free compute unit; hipHostMalloc is as expensive as hipMalloc, hence the
// same recycling (process-wide: pinned memory belongs to no device).
namespace {
struct PinPool {
    std::mutex mu;
    std::unordered_map<void*, size_t> bucket_of;
    std::map<size_t, std::vector<void*>> free_blocks;
};
PinPool g_pin;
}  // namespace
static void* pin_alloc(size_t bytes) {
    const size_t b = pool_bucket(bytes);
    {
        std::lock_guard<std::mutex> lk(g_pin.mu);
        auto it = g_pin.free_blocks.find(b);
        if (it != g_pin.free_blocks.end() && !it->second.empty()) {
            void* ptr = it->second.back();
            it->second.pop_back();
            return ptr;
        }
    }
    void* ptr = nullptr;
    if (hipHostMalloc(&ptr, b, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_pin.mu);
    g_pin.bucket_of[ptr] = b;
    return ptr;
}
static void pin_free(void* ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_pin.mu);
    auto it = g_pin.bucket_of.find(ptr);
    if (it == g_pin.bucket_of.end()) return;
    g_pin.free_blocks[it->second].push_back(ptr);
}
extern "C" int gnnx_sparse_tiny_per_workgroup(int32_t D, int32_t H, int32_t C) { return gnnx::sp_mix_tiny(D, H, C); }

extern "C" int gnnx_pool_trim(void) {
    DevPool& P = pool_here();
    {
        std::lock_guard<std::mutex> lk(P.mu);
        for (auto& kv : P.free_blocks) {
            for (void* ptr : kv.second) {
                (void)hipFree(ptr);
                P.bucket_of.erase(ptr);
            }
            kv.second.clear();
        }
        P.cached_bytes = 0;
    }
    std::lock_guard<std::mutex> lk(g_pin.mu);
    for (auto& kv : g_pin.free_blocks) {
        for (void* ptr : kv.second) {
            (void)hipHostFree(ptr);
            g_pin.bucket_of.erase(ptr);
        }
        kv.second.clear();
    }
    return 0;
}

// ---- table uploads ---------------------------------------------------------------------------------------------------------
// A synchronous hipMemcpy runs on the null stream, whose hardware queue may be the one a 4 ms optimisation launch occupies: the
// upload of the next batch's tile tables then waits for it.  The calling thread can name the stream its plan-building uploads go
// to (gnnx_set_service_stream: the prepare stream of pipeline.BatchPipeline); default = the null stream as before.
static thread_local hipStream_t g_service_stream = nullptr;
extern "C" int gnnx_set_service_stream(void* stream) {
    g_service_stream = static_cast<hipStream_t>(stream);
    return 0;
}
static hipError_t upload_sync(void* dst, const void* src, size_t bytes) {
    if (!g_service_stream) return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_service_stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(g_service_stream);   // the sources are host temporaries
}

// measurement / calibration hooks: keep a stream busy for `micros` microseconds; the launch lanes as hipStream_t
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static_assert(GNNX_FEAT_STRIDE == FS && GNNX_MAX_CLASSES == CMAX && GNNX_LOSS_TERMS == NLOSS, "header mismatch");

struct GraphKey {
    const void *A, *X, *yhat, *M, *Abar, *fm, *loss, *ws;
    gnnx_resume rs;
    gnnx_hyper hy;
    bool operator==(const GraphKey& o) const { return std::memcmp(this, &o, sizeof(GraphKey)) == 0; }
};

constexpr int N_SPC = 5;                       // size classes of k_sparse_resident (0..2, 4) + k_sparse_large (3)
constexpr int SPC_THREADS[N_SPC] = {1024, 256, 64, SPL_THREADS, 512};
constexpr int SPC_LARGE = 3, SPC_512 = 4;
constexpr int N_SIDE = RES_NBMAX + N_SPC;
// Resident launches of one run that must overlap go to different LANES: a small process-wide set of side streams, created
// once per device and never destroyed.  A HIP stream is bound to one of a handful of hardware queues when it is created, and
// two streams on the same queue run their kernels back to back: with eight fresh streams per plan (round 1) which groups
// overlapped depended on how many streams the process had created before (measured: the 64- and 256-thread launches of syn5
// overlapped in one session, 3.6 ms, and serialised in the next, 6.3 ms).
constexpr int N_LANES = 3;
constexpr int MAX_DEVICES = 16;
static hipStream_t g_lane[MAX_DEVICES][N_LANES] = {};
static std::mutex g_lane_mutex;   // gnnx_run may be entered from several host threads (ctypes releases the GIL)
static hipStream_t lane_stream(int i) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    std::lock_guard<std::mutex> lock(g_lane_mutex);
    hipStream_t& st = g_lane[dev][i % N_LANES];
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    return st;
}
// a stream restricted to the compute units whose bits are set in `mask` (bit c of word c / 32 = CU c): pipeline.BatchPipeline keeps a few
// CUs for the kernels of its prepare / fetch stages so that they do not wait for a CU behind the optimisations of the batches in flight
// (every workgroup of a resident launch holds its CU for milliseconds; there is no preemption).  NULL on failure.
extern "C" void* gnnx_stream_create_cu_mask(const uint32_t* mask, int32_t words) {
    hipStream_t st = nullptr;
    if (!mask || words < 1 || hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask) != hipSuccess) return nullptr;
    return (void*)st;
}
extern "C" void* gnnx_lane_stream(int32_t i) { return (i >= 0 && i < N_LANES) ? (void*)lane_stream(i) : nullptr; }
__global__ void k_debug_lane_sums(const float* in, float* out) {
    const float v = in[threadIdx.x];
    out[threadIdx.x] = xor32_sum(v);
    out[64 + threadIdx.x] = v + __shfl_xor(v, 32);
    out[128 + threadIdx.x] = xor16_sum(v);
    out[192 + threadIdx.x] = v + __shfl_xor(v, 16);
}
extern "C" int gnnx_debug_lane_sums(const float* in, float* out, void* stream) {
    if (!in || !out) return fail("null argument");
    hipLaunchKernelGGL(k_debug_lane_sums, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), in, out);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int gnnx_debug_spin(void* stream, int32_t micros) {
    int rate_khz = 100000;   // wall_clock64 ticks at 100 MHz on gfx950
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)micros * rate_khz / 1000);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
constexpr int CAT_SPARSE = RES_NBMAX + 1;      // CAT_SPARSE + k: sparse resident kernel of size class k

struct gnnx_plan_s {
    gnnx_problem prob{};
    std::vector<TargetMeta> meta;
    int64_t Q = 0, R = 0;
    int n_conv = 0, n_mask = 0;      // tile tables over ALL targets (streaming path for everything)
    // hybrid split: node-mode targets of up to res_nbmax row blocks run in the on-chip-resident kernels, the rest stream
    int n_res = 0, n_big = 0, n_conv_big = 0, n_mask_big = 0;
    int res_nbmax = RES_NBMAX;
    int res_count[RES_NBMAX + 1] = {};  // resident targets with nb row blocks
    int res_first[RES_NBMAX + 1] = {};  // their first index in d_res
    int32_t* d_res = nullptr;        // target ids of the resident set (grouped by nb, largest first)
    int32_t* d_big = nullptr;        // target ids of the streaming set
    ConvTile* d_conv_big = nullptr;
    MaskTile* d_mask_big = nullptr;  // tile pairs of the streaming set: its own pool block, built on first use (ensure_mask_big)
    std::vector<int32_t> big_ids;    // the streaming set on the host (largest first)
    // work units of k_conv (groups of adjacent row blocks, largest targets first) for the two tile tables
    ConvUnit* d_unit = nullptr;
    ConvUnit* d_unit_big = nullptr;
    ConvJoin* d_join = nullptr;
    ConvJoin* d_join_big = nullptr;
    int n_unit = 0, n_unit_big = 0, n_join = 0, n_join_big = 0;
    float* d_cpart = nullptr;        // slabs of the K slices (shared by the two tables: one of them runs at a time)
    size_t cap_slabs = 0;
    void* create_block = nullptr;    // d_meta, d_conv, d_mask, d_wts, d_raw_off, d_unit, d_join live in it (one upload)
    void* split_block = nullptr;     // d_res, d_sp[], d_big, d_conv_big, d_mask_big, d_unit_big, d_join_big (one upload per split)
    // The two blocks are uploaded ASYNCHRONOUSLY from pinned staging when the calling thread has named a service stream
    // (gnnx_set_service_stream): up_ev is recorded behind the last such upload on up_stream, every entry point that enqueues readers of
    // the tables on another stream makes that stream wait for it (use_tables), and the staging blocks are recycled only once it completed.
    void* create_pin = nullptr;
    void* split_pin = nullptr;
    hipEvent_t up_ev = nullptr;
    hipStream_t up_stream = nullptr;
    bool split_dirty = false;        // the split lists exist on the host only (gnnx_plan_create defers the upload: gnnx_plan_analyze usually
                                     // replaces the default split at once); committed by the first call that needs them (use_tables)
    std::vector<int64_t> edge_counts;   // upper-triangle edges per target, counted by the last gnnx_plan_analyze* (with the row starts in d_rowcnt)
    bool adam_shared = false;        // d_adam points into the process-wide table cache (not this plan's to free)
    float* d_watt = nullptr;         // method="att": attention weights of the three layers (gnnx_set_att_weights); every run takes k_att
    // the resident kernels run beside the streaming launches, each group on its own stream:
    // [0..RES_NBMAX) dense resident kernels by row blocks, [RES_NBMAX + k] sparse resident kernel of size class k
    hipEvent_t ev_in = nullptr, ev_out[N_SIDE + 1] = {};   // (slot N_SIDE: the packed single-wave launch, k_sparse_resident_tiny*)
    hipEvent_t ev_t0[N_SIDE + 1] = {};   // start of the resident launch on its side stream (timed, for gnnx_resident_times)
    bool launched[N_SIDE + 1] = {};
    // Packed single-wave targets (k_sparse_resident_tiny16 / 12, gnnx_sparse.hpp): tiny_pack[t] = 1 for the targets of the 64-thread class whose
    // slim LDS form fits a slice; they come FIRST in d_sp[2] (n_tiny_pack of them) and take their own launch, the others run where they ran.
    std::vector<char> tiny_pack;
    int n_tiny_pack = 0;
    int tiny_per_cu = 0;             // 16 / 12: single-wave targets per compute unit of the packed launch; 0: no packed launch (GNNX_TINY_PACK)
    std::vector<int> order;          // targets, largest first
    std::vector<int> cat;            // per target: 0 streaming, 1..RES_NBMAX dense resident kernel of that many row blocks, CAT_SPARSE
    int xconst = 0;                  // every target of the sparse resident classes has constant feature rows (gnnx_plan_analyze_features):
                                     // 0 general form, 1 the general form's products without the gathers (bit-identical), 2 the algebraic form
    std::vector<int32_t> nnz;        // per target (directed edge entries, row slots) from gnnx_plan_analyze, empty before
    int n_sp[N_SPC] = {};            // targets of the sparse resident kernel, per size class (1024 / 256 / 64 threads)
    bool pair256 = false;            // the 256-thread class runs two targets to a 512-thread workgroup of the mixed launch (gnnx_plan_analyze)
    int32_t* d_sp[N_SPC] = {};
    int n_sparse() const { return n_sp[0] + n_sp[1] + n_sp[2] + n_sp[3] + n_sp[4]; }
    int32_t* d_nnz = nullptr;
    int32_t* d_rowdeg = nullptr;       // off-diagonal non-zeros of every row of the batch (gnnx_plan_analyze, k_row_degrees)
    int32_t* d_csr_rowptr = nullptr;   // CSR of the targets of k_sparse_large (gnnx_plan_analyze)
    unsigned short* d_csr_col = nullptr;
    unsigned short* d_csr_row = nullptr;   // row of every directed entry
    long long* d_csr_off = nullptr;    // [2 T]: offsets of target t into the two arrays
    float* d_adam = nullptr;         // per-iteration Adam scalars for the resident kernel
    DeadBlock* d_dead = nullptr;     // work list of k_dead_entries (loss logging on the edge-sparse kernels); rebuilt after a new split
    int n_dead = 0;
    uint32_t* trace_gates = nullptr; // gnnx_set_trace: caller's device buffers, filled by the runs that follow (null: no trace)
    int32_t* trace_pool = nullptr;
    int32_t* d_rowcnt = nullptr;     // [R] scratch of gnnx_edge_counts
    int64_t* d_raw_off = nullptr;    // [T] float offset of target t's n x n block in the unpadded RNG stream (gnnx_scatter_masks)
    int64_t total_raw = 0;
    std::vector<float> adam_host;
    gnnx_hyper adam_for{};
    int adam_first = 0;              // first_iter the table was built for (gnnx_run_resume)
    TargetMeta* d_meta = nullptr;
    ConvTile* d_conv = nullptr;   // every 32-row block of every target
    MaskTile* d_mask = nullptr;   // tile pairs of ALL targets: built on first use (ensure_mask_all)
    float* d_wts = nullptr;
    double sum_n2 = 0;
    // workspace offsets in bytes
    size_t o_mM, o_vM, o_XT, o_Zraw, o_g3, o_z3p, o_U[3], o_UT[3], o_rn[3], o_dZ[3], o_dZT[3], o_dE, o_arg, o_df, o_f[2], o_mf, o_vf,
        o_probs, o_Xn[2], o_XnT[2], o_bnr[2], ws_bytes;
    hipGraphExec_t gexec = nullptr;
    GraphKey gkey{};
    // streams that may still be executing work that reads this plan's device tables, each with the event recorded behind the last such
    // call: the tables go back to the recycling pool (pool_free does not synchronise, unlike the hipFree it replaced) only after these
    // events completed - gnnx_destroy of a job whose kernels still run, a new split under a running job
    std::vector<std::pair<hipStream_t, hipEvent_t>> busy;
};

// the call that just enqueued work on `s` reads the plan's tables: remember where that work ends
static void mark_busy(gnnx_handle h, hipStream_t s) {
    for (auto& b : h->busy)
        if (b.first == s) {
            (void)hipEventRecord(b.second, s);
            return;
        }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamSynchronize(s);   // no event to remember it by: wait now
        return;
    }
    (void)hipEventRecord(e, s);
    h->busy.push_back({s, e});
}
// every piece of enqueued work that reads the plan's tables has finished (cheap when it already has)
static void wait_idle(gnnx_handle h) {
    for (auto& b : h->busy) (void)hipEventSynchronize(b.second);
}

// the asynchronous table uploads of this plan have completed (cheap when they already have)
static void wait_uploads(gnnx_handle h) {
    if (h->up_stream && h->up_ev) (void)hipEventSynchronize(h->up_ev);
}
static int build_split(gnnx_handle h, bool upload = true);
static void edge_rows(gnnx_handle h, const float* A, int32_t* rowcnt, int64_t* counts, hipStream_t s);
// Called by every entry point that enqueues work reading the plan's tables on stream `s`: the split lists are on the device
// (need_split) and `s` is ordered behind the last asynchronous table upload.
static int use_tables(gnnx_handle h, hipStream_t s, bool need_split = true) {
    if (need_split && h->split_dirty)
        if (int rc = build_split(h)) return rc;
    if (h->up_stream && h->up_ev && s != h->up_stream) {
        if (hipStreamWaitEvent(s, h->up_ev, 0) != hipSuccess) (void)hipEventSynchronize(h->up_ev);
    }
    return 0;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Units of k_conv for a tile table (row blocks of a target are adjacent in it), longest K range first, and the row blocks whose
// slices k_conv_reduce joins.  Both knobs are OFF by default - measured on the BA-House x100k streaming set (243 targets, ld up
// to 4992; tools/probe_conv.py, profiles/r03_conv_*): with the ring K loop one row block per workgroup over whole rows runs at
// 82-90 us per launch, K slices of 2048 / 1024 / 512 rows at 95 / 97 / 109, groups of 2 / 4 row blocks at 90 / 105.
//   GNNX_CONV_KU   = K rows per slice (0 = whole rows; a target of ld >= 1.5 KU is cut into round(ld / KU) slices whose
//                    partial tiles meet in slabs; k_conv_reduce joins them in the next launch - an in-launch join with agent-scope
//                    release / acquire fences was measured too: 166 -> 530 us as the slices get shorter, the L2 write-back of
//                    the release dominates)
//   GNNX_CONV_WIDE = row blocks per workgroup for ld >= 256 (1, 2 or 4)
struct UnitTables {
    std::vector<ConvUnit> units;
    std::vector<ConvJoin> joins;
    int slabs = 0;
};
static UnitTables make_units(const std::vector<ConvTile>& tiles) {
    UnitTables o;
    o.units.reserve(tiles.size());
    const char* e_wide = getenv("GNNX_CONV_WIDE");
    const char* e_ku = getenv("GNNX_CONV_KU");
    const int wide = e_wide ? atoi(e_wide) : 1;
    const int ku = e_ku ? atoi(e_ku) : 0;
    for (size_t i = 0; i < tiles.size();) {
        const ConvTile& tl = tiles[i];
        const int ld = tl.tm.ld, nb = ld / TILE;
        int nrb = 1;
        if (ld >= 256 && wide > 1) {
            const int left = nb - tl.rb;
            nrb = (left >= 4 && wide >= 4) ? 4 : (left >= 2) ? 2 : 1;
        }
        int nks = (ku >= TILE && ld >= ku + ku / 2) ? (ld + ku / 2) / ku : 1;
        if (nks > nb) nks = nb;
        for (int ks = 0; ks < nks; ++ks) {
            const int b0 = (int)((int64_t)nb * ks / nks), b1 = (int)((int64_t)nb * (ks + 1) / nks);
            o.units.push_back({tl.t, tl.rb, nrb, b0 * TILE, b1 * TILE, ks, nks, nks > 1 ? o.slabs : 0, tl.tm});
        }
        if (nks > 1) {
            for (int j = 0; j < nrb; ++j) o.joins.push_back({tl.t, tl.rb + j, nks, o.slabs + j * nks, tl.tm});
            o.slabs += nrb * nks;
        }
        i += nrb;
    }
    std::stable_sort(o.units.begin(), o.units.end(), [](const ConvUnit& a, const ConvUnit& b) { return a.ke - a.kb > b.ke - b.kb; });
    return o;
}

// The tables of a plan travel as ONE block: every upload is a host-blocking round trip through a stream whose small copy kernel has
// to find a free compute unit - in a pipelined job (pipeline.BatchPipeline) the chip is full of the optimisations of the batches
// ahead, and six uploads per plan + three per split cost the preparing thread nine such waits.  The builder packs the host arrays
// (256-B aligned), uploads once and points the plan's table pointers into the block.
struct BlockBuilder {
    std::vector<char> host;
    std::vector<std::pair<void**, size_t>> fix;
    template <class T>
    void add(T*& slot, const T* data, size_t count) {
        slot = nullptr;
        if (!count) return;
        const size_t off = (host.size() + 255) / 256 * 256;
        host.resize(off + sizeof(T) * count);
        std::memcpy(host.data() + off, data, sizeof(T) * count);
        fix.push_back({reinterpret_cast<void**>(&slot), off});
    }
    template <class T>
    void add(T*& slot, const std::vector<T>& v) { add(slot, v.data(), v.size()); }
    // (the caller has waited for the work that reads the block this one replaces: wait_idle)
    hipError_t commit(gnnx_handle h, void*& block, void*& pin) {
        if (block || pin) wait_uploads(h);   // an upload of the block / staging block about to be recycled may still be in flight
        if (block) (void)pool_free(block);
        block = nullptr;
        pin_free(pin);
        pin = nullptr;
        if (host.empty()) return hipSuccess;
        hipError_t e = pool_malloc(&block, host.size());
        if (e != hipSuccess) return e;
        if (g_service_stream) pin = pin_alloc(host.size());
        if (pin) {   // enqueue only: the preparing thread does not wait for the copy kernel to get a compute unit
            std::memcpy(pin, host.data(), host.size());
            e = hipMemcpyAsync(block, pin, host.size(), hipMemcpyHostToDevice, g_service_stream);
            if (e != hipSuccess) return e;
            if (!h->up_ev) e = hipEventCreateWithFlags(&h->up_ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(h->up_ev, g_service_stream);
            if (e != hipSuccess) {   // no event to order the readers by: wait here
                (void)hipStreamSynchronize(g_service_stream);
                h->up_stream = nullptr;
            } else {
                h->up_stream = g_service_stream;
            }
        } else {
            e = upload_sync(block, host.data(), host.size());
            if (e != hipSuccess) return e;
        }
        for (auto& f : fix) *f.first = static_cast<char*>(block) + f.second;
        return hipSuccess;
    }
};

static hipError_t add_units(gnnx_handle h, BlockBuilder& bb, const std::vector<ConvTile>& tiles, ConvUnit*& d_units, int& n_units,
                            ConvJoin*& d_joins, int& n_joins) {
    const UnitTables ut = make_units(tiles);
    n_units = (int)ut.units.size();
    n_joins = (int)ut.joins.size();
    bb.add(d_units, ut.units);
    bb.add(d_joins, ut.joins);
    if ((size_t)ut.slabs > h->cap_slabs) {
        if (h->d_cpart) (void)pool_free(h->d_cpart);
        h->d_cpart = nullptr;
        h->cap_slabs = 0;
        hipError_t e = pool_malloc(&h->d_cpart, sizeof(float) * TILE * FS * (size_t)ut.slabs);
        if (e != hipSuccess) return e;
        h->cap_slabs = ut.slabs;
    }
    return hipSuccess;
}

// (Re)build the hybrid split from h->cat: id lists of the resident kernels, tile tables of the streaming remainder,
// side streams.  Invalidates a captured graph.
static int build_split(gnnx_handle h, bool upload) {
#define SPLITCK(x)                                                                 \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)
    wait_idle(h);   // the split block (and the slabs) it replaces may still be read by an earlier run of this plan
    if (h->gexec) {
        (void)hipGraphExecDestroy(h->gexec);
        h->gexec = nullptr;
    }
    h->d_res = h->d_big = nullptr;   // (they point into split_block, which the commit below replaces)
    if (h->d_dead) {
        (void)pool_free(h->d_dead);
        h->d_dead = nullptr;
        h->n_dead = 0;
    }
    for (int k = 0; k < N_SPC; ++k) h->d_sp[k] = nullptr;
    h->d_conv_big = nullptr;
    if (h->d_mask_big) (void)pool_free(h->d_mask_big);
    h->d_mask_big = nullptr;
    h->d_unit_big = nullptr;
    h->d_join_big = nullptr;
    h->n_unit_big = h->n_join_big = 0;
    std::vector<ConvTile> conv_big;
    long long n_mask_big = 0;
    std::vector<int32_t> res_ids, sp_ids[N_SPC], big_ids, packed_ids;
    for (int k = 0; k <= RES_NBMAX; ++k) h->res_count[k] = h->res_first[k] = 0;
    for (int t : h->order) {  // sorted by ld: the dense resident groups are contiguous
        const int nb = h->meta[t].ld / TILE, c = h->cat[t];
        if (c >= CAT_SPARSE) {
            sp_ids[c - CAT_SPARSE].push_back(t);
            if (c == CAT_SPARSE + 2 && !h->tiny_pack.empty() && h->tiny_pack[t]) packed_ids.push_back(t);
        } else if (c >= 1) {
            if (h->res_count[nb]++ == 0) h->res_first[nb] = (int)res_ids.size();
            res_ids.push_back(t);
        } else {
            big_ids.push_back(t);
            for (int rb = 0; rb < nb; ++rb) conv_big.push_back({t, rb, h->meta[t]});
            n_mask_big += (long long)nb * (nb + 1) / 2;   // the tile pairs themselves: on first use (ensure_mask_big)
        }
    }
    h->n_tiny_pack = (int)packed_ids.size();
    if (h->n_tiny_pack) {   // the packed targets lead the 64-thread class's list (both parts keep the plan's order)
        std::vector<int32_t> rest;
        for (int t : sp_ids[2])
            if (!h->tiny_pack[t]) rest.push_back(t);
        sp_ids[2] = packed_ids;
        sp_ids[2].insert(sp_ids[2].end(), rest.begin(), rest.end());
    }
    h->big_ids = big_ids;
    h->n_res = (int)res_ids.size();
    for (int k = 0; k < N_SPC; ++k) h->n_sp[k] = (int)sp_ids[k].size();
    h->n_big = (int)big_ids.size();
    h->n_conv_big = (int)conv_big.size();
    h->n_mask_big = (int)n_mask_big;
    BlockBuilder bb;
    bb.add(h->d_res, res_ids);
    for (int k = 0; k < N_SPC; ++k) bb.add(h->d_sp[k], sp_ids[k]);
    const bool any_resident = h->n_res || h->n_sparse();
    if (any_resident && h->n_big) {
        bb.add(h->d_big, big_ids);
        bb.add(h->d_conv_big, conv_big);
        SPLITCK(add_units(h, bb, conv_big, h->d_unit_big, h->n_unit_big, h->d_join_big, h->n_join_big));
    }
    if (upload) {
        SPLITCK(bb.commit(h, h->split_block, h->split_pin));
        h->split_dirty = false;
    } else {
        // counts only; the device pointers stay null until use_tables() / the next build_split commits the lists
        h->d_res = h->d_big = nullptr;
        for (int k = 0; k < N_SPC; ++k) h->d_sp[k] = nullptr;
        h->d_conv_big = nullptr;
        h->d_unit_big = nullptr;
        h->d_join_big = nullptr;
        h->split_dirty = true;
    }
    if (any_resident && !h->ev_in) SPLITCK(hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    // (Disjoint compute-unit masks for the sparse and the single-tile dense launch - hipExtStreamCreateWithCUMask - were
    // measured on syn1 and made the sparse launch slower, 9.7 vs 6.4 ms in situ: not used.)
    for (int k = 0; k <= N_SIDE; ++k) {
        const bool need = k == N_SIDE ? h->n_tiny_pack > 0 : k < RES_NBMAX ? h->res_count[k + 1] > 0 : h->n_sp[k - RES_NBMAX] > 0;
        if (need && !h->ev_out[k]) {
            SPLITCK(hipEventCreate(&h->ev_out[k]));
            SPLITCK(hipEventCreate(&h->ev_t0[k]));
        }
    }
#undef SPLITCK
    return 0;
}

extern "C" const char* gnnx_last_error(void) { return g_err.c_str(); }
extern "C" const char* gnnx_version(void) { return "gnnx-hip 0.1 (gfx950, mfma_f32_32x32x2f32)"; }

extern "C" int gnnx_plan_create(const gnnx_problem* prob, const gnnx_model* model, gnnx_handle* out) {
    if (!prob || !model || !out) return fail("null argument");
    if (prob->num_targets <= 0) return fail("num_targets must be positive");
    if (prob->D < 1 || prob->D > FS || prob->H < 1 || prob->H > FS || prob->O < 1 || prob->O > FS)
        return fail("D, H, O must be in [1, 32] for the 32-wide MFMA feature tile");
    if (prob->C < 1 || prob->C > CMAX) return fail("C must be in [1, 32]");
    auto* h = new gnnx_plan_s();
    h->prob = *prob;
    const int T = prob->num_targets;
    h->meta.resize(T);
    std::vector<ConvTile> conv;
    long long n_mask_all = 0;
    for (int t = 0; t < T; ++t) {
        const int n = prob->n[t];
        if (n < 1) {
            delete h;
            return fail("empty sub-graph (n < 1) for target " + std::to_string(t));
        }
        TargetMeta& m = h->meta[t];
        m.n = n;
        m.ld = (n + TILE - 1) / TILE * TILE;
        m.t = prob->graph_mode ? 0 : prob->target_row[t];
        m.y_gt = prob->gt_label[t];
        if (!prob->graph_mode && (m.t < 0 || m.t >= n)) {
            delete h;
            return fail("target_row out of range for target " + std::to_string(t));
        }
        if (m.y_gt < 0 || m.y_gt >= prob->C) {
            delete h;
            return fail("gt_label out of range for target " + std::to_string(t));
        }
        m.offQ = h->Q;
        m.offR = h->R;
        h->Q += (int64_t)m.ld * m.ld;
        h->R += m.ld;
        h->sum_n2 += (double)n * n;
    }
    // tile tables, largest targets first so the long poles start early
    h->order.resize(T);
    for (int t = 0; t < T; ++t) h->order[t] = t;
    std::stable_sort(h->order.begin(), h->order.end(), [&](int a, int b) { return h->meta[a].ld > h->meta[b].ld; });
    const bool resident_ok = !prob->graph_mode && prob->C <= RES_CMAX && !prob->mask_relu && !prob->bn;
    if (const char* env = std::getenv("GNNX_RESIDENT_MAX_BLOCKS")) {  // tuning knob, see include/gnnx.h
        const int v = std::atoi(env);
        h->res_nbmax = v < 0 ? 0 : (v > RES_NBMAX ? RES_NBMAX : v);
    }
    // Default split (before gnnx_plan_analyze has seen the adjacency): a batch whose targets ALL fit the dense
    // resident kernels runs entirely on chip (no streaming launch at all).  Otherwise only the single-tile targets ride
    // beside the streaming chain of the larger ones: measured on syn1 (400 targets, 72 of them with >= 4 row blocks),
    // moving the 2- and 3-block targets beside the chain as well made the batch slower (22.2 -> 24-38 ms: their
    // workgroups hold whole CUs for the full run and add hardware queues), while the streaming chain of the large
    // targets stays the long pole either way.
    {
        int max_nb = 0;
        for (int t = 0; t < T; ++t) max_nb = std::max(max_nb, h->meta[t].ld / TILE);
        if (max_nb > h->res_nbmax) h->res_nbmax = std::min(h->res_nbmax, 1);
    }
    h->cat.assign(T, 0);
    for (int t : h->order) {
        const int nb = h->meta[t].ld / TILE;
        if (resident_ok && nb <= h->res_nbmax) h->cat[t] = nb;
        for (int rb = 0; rb < nb; ++rb) conv.push_back({t, rb, h->meta[t]});
        n_mask_all += (long long)nb * (nb + 1) / 2;
    }
    // The tile-pair table of k_mask over ALL targets (sum of nb (nb + 1) / 2 entries: 3.4 M entries = 164 MB for the 16 384-target
    // BA-House x100k set, 0.28 s of host loops + upload per plan) is built on first use (ensure_mask_all): a plan whose targets the
    // analysis routes to the edge-sparse kernels never needs it.
    if (n_mask_all > 0x7fffffffLL) {
        delete h;
        return fail("too many tile pairs in one plan");
    }
    h->n_conv = (int)conv.size();
    h->n_mask = (int)n_mask_all;

    // packed, zero padded model block
    std::vector<float> w(WT_TOTAL, 0.0f);
    const int din[3] = {prob->D, prob->H, prob->H}, dout[3] = {prob->H, prob->H, prob->O};
    for (int l = 0; l < 3; ++l) {
        for (int k = 0; k < din[l]; ++k)
            for (int c = 0; c < dout[l]; ++c) w[WT_W + l * 1024 + k * 32 + c] = model->W[l][k * dout[l] + c];
        for (int c = 0; c < dout[l]; ++c) w[WT_B + l * 32 + c] = model->b[l] ? model->b[l][c] : 0.0f;
    }
    const int E = prob->H + prob->H + prob->O;
    const int eoff[3] = {0, prob->H, 2 * prob->H};
    for (int c = 0; c < prob->C; ++c) {
        for (int l = 0; l < 3; ++l)
            for (int j = 0; j < dout[l]; ++j) w[WT_WP + c * 96 + l * 32 + j] = model->Wp[c * E + eoff[l] + j];
        w[WT_BP + c] = model->bp[c];
    }

#define PLANCK(x)                                                                                      \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            gnnx_destroy(h);                                                                           \
            return fail(std::string(#x) + ": " + hipGetErrorString(e_));                               \
        }                                                                                              \
    } while (0)
    {
        std::vector<int64_t> ro(T);
        for (int t = 0; t < T; ++t) {
            ro[t] = h->total_raw;
            h->total_raw += (int64_t)h->meta[t].n * h->meta[t].n;
        }
        BlockBuilder bb;
        bb.add(h->d_meta, h->meta);
        bb.add(h->d_conv, conv);
        PLANCK(add_units(h, bb, conv, h->d_unit, h->n_unit, h->d_join, h->n_join));
        bb.add(h->d_wts, w);
        bb.add(h->d_raw_off, ro);
        PLANCK(bb.commit(h, h->create_block, h->create_pin));
    }
#undef PLANCK
    if (int rc = build_split(h, /*upload=*/false)) {
        gnnx_destroy(h);
        return rc;
    }

    // workspace carving
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    const size_t qb = sizeof(float) * (size_t)h->Q, rb32 = sizeof(float) * (size_t)h->R * FS, rb1 = sizeof(float) * (size_t)h->R;
    h->o_mM = take(qb);
    h->o_vM = take(qb);
    h->o_XT = take(rb32);
    h->o_Zraw = take(rb32);
    h->o_g3 = take(rb1);
    h->o_z3p = take(rb1);
    for (int l = 0; l < 3; ++l) {
        h->o_U[l] = take(rb32);
        h->o_UT[l] = take(rb32);
        h->o_rn[l] = take(rb1);
        h->o_dZ[l] = take(rb32);
        h->o_dZT[l] = take(rb32);
    }
    h->o_dE = take(sizeof(float) * T * 96);
    h->o_arg = take(sizeof(int32_t) * T * 96);
    h->o_df = take(rb1);  // (R / 32) row blocks x 32 floats
    h->o_f[0] = take(sizeof(float) * T * FS);
    h->o_f[1] = take(sizeof(float) * T * FS);
    h->o_mf = take(sizeof(float) * T * FS);
    h->o_vf = take(sizeof(float) * T * FS);
    h->o_probs = take(sizeof(float) * T * CMAX);
    for (int l = 0; l < 2; ++l) {  // --bn: standardised activations of the two hidden layers and 1 / std per row
        h->o_Xn[l] = prob->bn ? take(rb32) : 0;
        h->o_XnT[l] = prob->bn ? take(rb32) : 0;
        h->o_bnr[l] = prob->bn ? take(rb1) : 0;
    }
    h->ws_bytes = o;
    *out = h;
    return 0;
}

extern "C" int gnnx_destroy(gnnx_handle h) {
    if (!h) return 0;
    wait_idle(h);   // a job dropped while its kernels run (an abandoned pipeline, an exception between launch and fetch) must not hand
                    // its tables to the next plan under them
    for (auto& b : h->busy) (void)hipEventDestroy(b.second);
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    for (int k = 0; k <= N_SIDE; ++k) {
        if (h->ev_out[k]) (void)hipEventDestroy(h->ev_out[k]);
        if (h->ev_t0[k]) (void)hipEventDestroy(h->ev_t0[k]);
    }
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    if (h->d_nnz) (void)pool_free(reinterpret_cast<char*>(h->d_nnz) - sizeof(int64_t) * h->prob.num_targets);
    if (h->d_dead) (void)pool_free(h->d_dead);
    if (h->d_mask) (void)pool_free(h->d_mask);
    if (h->d_mask_big) (void)pool_free(h->d_mask_big);
    if (h->d_rowdeg) (void)pool_free(h->d_rowdeg);
    if (h->d_csr_rowptr) (void)pool_free(h->d_csr_rowptr);
    if (h->d_csr_col) (void)pool_free(h->d_csr_col);
    if (h->d_csr_row) (void)pool_free(h->d_csr_row);
    if (h->d_csr_off) (void)pool_free(h->d_csr_off);
    if (h->d_adam && !h->adam_shared) (void)pool_free(h->d_adam);
    if (h->d_rowcnt) (void)pool_free(h->d_rowcnt);
    if (h->d_cpart) (void)pool_free(h->d_cpart);
    if (h->d_watt) (void)pool_free(h->d_watt);
    wait_uploads(h);   // (a plan destroyed before anything read its tables: the uploads themselves may still be in flight)
    if (h->up_ev) (void)hipEventDestroy(h->up_ev);
    pin_free(h->create_pin);
    pin_free(h->split_pin);
    if (h->create_block) (void)pool_free(h->create_block);
    if (h->split_block) (void)pool_free(h->split_block);
    delete h;
    return 0;
}

extern "C" int64_t gnnx_total_q(gnnx_handle h) { return h ? h->Q : -1; }
extern "C" int64_t gnnx_total_rows(gnnx_handle h) { return h ? h->R : -1; }
extern "C" size_t gnnx_workspace_bytes(gnnx_handle h) { return h ? h->ws_bytes : 0; }

extern "C" int gnnx_get_layout(gnnx_handle h, int32_t* ld, int64_t* offQ, int64_t* offR) {
    if (!h) return fail("null plan");
    for (size_t t = 0; t < h->meta.size(); ++t) {
        if (ld) ld[t] = h->meta[t].ld;
        if (offQ) offQ[t] = h->meta[t].offQ;
        if (offR) offR[t] = h->meta[t].offR;
    }
    return 0;
}

static Params make_params(gnnx_handle h, const gnnx_hyper* hy, const float* A, const float* X, const float* yhat,
                          float* M, float* Abar, float* loss, void* ws) {
    Params p{};
    char* w = static_cast<char*>(ws);
    p.meta = h->d_meta;
    p.A = A;
    p.M = M;
    p.mM = reinterpret_cast<float*>(w + h->o_mM);
    p.vM = reinterpret_cast<float*>(w + h->o_vM);
    p.Abar = Abar;
    p.X = X;
    p.XT = reinterpret_cast<float*>(w + h->o_XT);
    p.yhat = yhat;
    p.Zraw = reinterpret_cast<float*>(w + h->o_Zraw);
    p.g3 = reinterpret_cast<float*>(w + h->o_g3);
    p.z3p = reinterpret_cast<float*>(w + h->o_z3p);
    p.cpart = h->d_cpart;
    for (int l = 0; l < 3; ++l) {
        p.U[l] = reinterpret_cast<float*>(w + h->o_U[l]);
        p.UT[l] = reinterpret_cast<float*>(w + h->o_UT[l]);
        p.rn[l] = reinterpret_cast<float*>(w + h->o_rn[l]);
        p.dZ[l] = reinterpret_cast<float*>(w + h->o_dZ[l]);
        p.dZT[l] = reinterpret_cast<float*>(w + h->o_dZT[l]);
    }
    p.dE = reinterpret_cast<float*>(w + h->o_dE);
    p.argrow = reinterpret_cast<int32_t*>(w + h->o_arg);
    p.df = reinterpret_cast<float*>(w + h->o_df);
    p.f[0] = reinterpret_cast<float*>(w + h->o_f[0]);
    p.f[1] = reinterpret_cast<float*>(w + h->o_f[1]);
    p.mf = reinterpret_cast<float*>(w + h->o_mf);
    p.vf = reinterpret_cast<float*>(w + h->o_vf);
    p.probs = reinterpret_cast<float*>(w + h->o_probs);
    p.bn = h->prob.bn;
    for (int l = 0; l < 2; ++l) {
        p.Xn[l] = h->prob.bn ? reinterpret_cast<float*>(w + h->o_Xn[l]) : nullptr;
        p.XnT[l] = h->prob.bn ? reinterpret_cast<float*>(w + h->o_XnT[l]) : nullptr;
        p.bnr[l] = h->prob.bn ? reinterpret_cast<float*>(w + h->o_bnr[l]) : nullptr;
    }
    p.loss = loss;
    p.wts = h->d_wts;
    p.D = h->prob.D;
    p.H = h->prob.H;
    p.O = h->prob.O;
    p.C = h->prob.C;
    p.graph_mode = h->prob.graph_mode;
    if (hy) {
        p.num_iters = hy->num_iters;
        p.opt = hy->opt;
        p.edge_only = hy->edge_results_only;
        p.lr = (float)hy->lr;
        p.beta2 = (float)(hy->opt == 2 ? hy->alpha : hy->beta2);
        p.omb1 = (float)(1.0 - hy->beta1);
        p.omb2 = (float)(1.0 - (hy->opt == 2 ? hy->alpha : hy->beta2));
        p.eps = (float)hy->eps;
        p.c_size = hy->c_size;
        p.c_feat_size = hy->c_feat_size;
        p.c_ent = hy->c_ent;
        p.c_lap = hy->c_lap;
    }
    return p;
}

// per-iteration scalars of the optimiser (adam_update, gnnx_kernels.hpp): `it` = steps already taken (first_iter + k), `k` = index
// of the iteration within this call (the learning-rate schedule is per call)
static void adam_scalars(const gnnx_hyper* hy, int it, float* step_size, float* bc2s, int k = -1) {
    const double lr = (hy->lr_schedule && k >= 0) ? hy->lr_schedule[k] : hy->lr;
    if (hy->opt == 0) {
        const double b1 = 1.0 - std::pow(hy->beta1, (double)(it + 1));
        const double b2 = 1.0 - std::pow(hy->beta2, (double)(it + 1));
        *step_size = (float)(lr / b1);
        *bc2s = (float)std::sqrt(b2);
    } else {
        *step_size = (float)lr;
        *bc2s = (hy->opt == 1 && it > 0) ? (float)hy->momentum : 0.0f;   // SGD: the momentum buffer starts as the first gradient
    }
}

// Process-wide cache of per-iteration optimiser scalar tables (a few KB each, at most ADAM_CACHE entries per device, never freed): key =
// the hyper-parameter struct (no schedule) + the first iteration.  Returns null when the cache is full or the upload fails - the caller
// then keeps a private table as before.
namespace {
struct AdamEntry { gnnx_hyper hy; int first; int dev; float* d; };
constexpr size_t ADAM_CACHE = 64;
std::mutex g_adam_mu;
std::vector<AdamEntry> g_adam_cache;
}  // namespace
static float* shared_adam_table(const gnnx_hyper* hy, int first_iter, const std::vector<float>& host) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(g_adam_mu);
    for (const AdamEntry& e : g_adam_cache)
        if (e.dev == dev && e.first == first_iter && std::memcmp(&e.hy, hy, sizeof(gnnx_hyper)) == 0) return e.d;
    if (g_adam_cache.size() >= ADAM_CACHE) return nullptr;
    float* d = nullptr;
    if (hipMalloc(&d, sizeof(float) * host.size()) != hipSuccess) return nullptr;
    if (upload_sync(d, host.data(), sizeof(float) * host.size()) != hipSuccess) {
        (void)hipFree(d);
        return nullptr;
    }
    g_adam_cache.push_back({*hy, first_iter, dev, d});
    return d;
}

// which tile tables a launch sequence walks: all targets, or only the streaming ("big") set of a hybrid run
struct Tables {
    const ConvUnit* unit;  // k_conv: groups of adjacent row blocks x K slice
    int n_unit;
    const ConvJoin* join;  // k_conv_reduce: the row blocks cut into slices
    int n_join;
    const ConvTile* conv;
    int n_conv;
    const MaskTile* mask;
    int n_mask;
    const int32_t* ids;  // target ids for per-target kernels (null = 0..T-1)
    int n_targets;
};
// the tile-pair table of k_mask / k_grad_edges over all targets, largest targets first (the order of the conv table)
static int ensure_mask_all(gnnx_handle h) {
    if (h->d_mask || !h->n_mask) return 0;
    std::vector<MaskTile> mask;
    mask.reserve((size_t)h->n_mask);
    for (int t : h->order) {
        const int nb = h->meta[t].ld / TILE;
        for (int I = 0; I < nb; ++I)
            for (int J = I; J < nb; ++J) mask.push_back({t, I, J, 0, h->meta[t]});
    }
    HIPCK(pool_malloc(&h->d_mask, sizeof(MaskTile) * mask.size()));
    HIPCK(upload_sync(h->d_mask, mask.data(), sizeof(MaskTile) * mask.size()));
    return 0;
}
static int ensure_mask_big(gnnx_handle h) {
    if (h->d_mask_big || !h->n_mask_big) return 0;
    std::vector<MaskTile> mask;
    mask.reserve((size_t)h->n_mask_big);
    for (int t : h->big_ids) {
        const int nb = h->meta[t].ld / TILE;
        for (int I = 0; I < nb; ++I)
            for (int J = I; J < nb; ++J) mask.push_back({t, I, J, 0, h->meta[t]});
    }
    HIPCK(pool_malloc(&h->d_mask_big, sizeof(MaskTile) * mask.size()));
    HIPCK(upload_sync(h->d_mask_big, mask.data(), sizeof(MaskTile) * mask.size()));
    return 0;
}
static Tables tables_all(gnnx_handle h) { return {h->d_unit, h->n_unit, h->d_join, h->n_join, h->d_conv, h->n_conv, h->d_mask, h->n_mask, nullptr, h->prob.num_targets}; }
static Tables tables_big(gnnx_handle h) { return {h->d_unit_big, h->n_unit_big, h->d_join_big, h->n_join_big, h->d_conv_big, h->n_conv_big, h->d_mask_big, h->n_mask_big, h->d_big, h->n_big}; }

template <int MODE>
static void launch_conv(const Tables& tb, const Params& p, int it, hipStream_t s) {
    hipLaunchKernelGGL((k_conv<MODE>), dim3(tb.n_unit), dim3(256), 0, s, p, tb.unit, it);
    if (MODE != BWD3 && tb.n_join) hipLaunchKernelGGL((k_conv_reduce<MODE>), dim3(tb.n_join), dim3(256), 0, s, p, tb.join, it);
}

template <bool UPDATE, bool WRITE_ABAR>
static void launch_mask(gnnx_handle h, const Tables& tb, const Params& p, int it, float ss, float b2, hipStream_t s) {
    const dim3 g(tb.n_mask), b(256);
    const bool node = !h->prob.graph_mode, loss = UPDATE && p.loss != nullptr;
    if (h->prob.mask_relu) {  // mask_act == "ReLU": no loss logging (gnnx_run rejects the combination)
        if (node) hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, true, false, true>), g, b, 0, s, p, tb.mask, it, ss, b2);
        else hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, false, false, true>), g, b, 0, s, p, tb.mask, it, ss, b2);
        return;
    }
    if (node && loss) hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, true, UPDATE>), g, b, 0, s, p, tb.mask, it, ss, b2);
    else if (node) hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, true, false>), g, b, 0, s, p, tb.mask, it, ss, b2);
    else if (loss) hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, false, UPDATE>), g, b, 0, s, p, tb.mask, it, ss, b2);
    else hipLaunchKernelGGL((k_mask<UPDATE, WRITE_ABAR, false, false>), g, b, 0, s, p, tb.mask, it, ss, b2);
}

// forward up to the head (+ in node mode the fused start of the backward pass)
static void launch_forward(gnnx_handle h, const Tables& tb, const Params& p, int it, hipStream_t s) {
    launch_conv<FWD1>(tb, p, it, s);
    launch_conv<FWD2>(tb, p, it, s);
    if (h->prob.graph_mode) {
        launch_conv<FWD3>(tb, p, it, s);
        hipLaunchKernelGGL(k_head, dim3(h->prob.num_targets), dim3(256), 0, s, p, it);
    } else {
        hipLaunchKernelGGL(k_node_head, dim3(tb.n_conv), dim3(256), 0, s, p, tb.conv, it);
    }
}

static void launch_backward(gnnx_handle h, const Tables& tb, const Params& p, int it, hipStream_t s) {
    if (h->prob.graph_mode) {
        launch_conv<BWD3>(tb, p, it, s);
        launch_conv<BWD2>(tb, p, it, s);
    }
    launch_conv<BWD1>(tb, p, it, s);
}

// the streaming job over the given tables, stream-ordered: usable directly or under stream capture
// (the Adam moments of M are initialised by the caller, init_stream_state, BEFORE the resident launches of a hybrid run are
// released: with gnnx_resume.m_out they live in the caller's buffer, which those launches write too)
static int enqueue_job(gnnx_handle h, const Tables& tb, const gnnx_hyper* hy, const Params& p, int first_iter, hipStream_t s) {
    const int T = h->prob.num_targets;
    if (p.loss) HIPCK(hipMemsetAsync(p.loss, 0, sizeof(float) * (size_t)T * hy->num_iters * NLOSS, s));
    hipLaunchKernelGGL(k_prep, dim3(tb.n_targets), dim3(256), 0, s, p, (const float*)nullptr, tb.ids);
    launch_mask<false, true>(h, tb, p, 0, 0.0f, 1.0f, s);
    for (int it = 0; it < hy->num_iters; ++it) {
        launch_forward(h, tb, p, it, s);
        launch_backward(h, tb, p, it, s);
        float ss, b2;
        adam_scalars(hy, first_iter + it, &ss, &b2, it);
        if (it + 1 < hy->num_iters)
            launch_mask<true, true>(h, tb, p, it, ss, b2, s);
        else
            launch_mask<true, false>(h, tb, p, it, ss, b2, s);  // keep Abar of the LAST forward (explain.py:209-211)
    }
    if (p.fs_out) hipLaunchKernelGGL(k_export_fstate, dim3(tb.n_targets), dim3(64), 0, s, p, tb.ids);
    HIPCK(hipGetLastError());
    return 0;
}

// Adam moments of the streaming path: zeros (a fresh run) or the caller's state (gnnx_run_resume)
static int init_stream_state(gnnx_handle h, const Params& p, hipStream_t s) {
    const size_t qb = sizeof(float) * (size_t)h->Q;
    if (p.m_in) {
        if (p.m_in != p.mM) HIPCK(hipMemcpyAsync(p.mM, p.m_in, qb, hipMemcpyDeviceToDevice, s));
    } else {
        HIPCK(hipMemsetAsync(p.mM, 0, qb, s));
    }
    if (p.v_in) {
        if (p.v_in != p.vM) HIPCK(hipMemcpyAsync(p.vM, p.v_in, qb, hipMemcpyDeviceToDevice, s));
    } else {
        HIPCK(hipMemsetAsync(p.vM, 0, qb, s));
    }
    return 0;
}

// the sparse resident kernel, instantiated per size class for the reference's encoders (node: D = 10, graph: D = 14;
// hidden 20) and for the general 32-wide case
// the specialised instantiations (compile-time widths) serve exactly the reference's encoders: D = 10 (node) / 14 (graph), H = O = 20
static bool exact_shape(gnnx_handle h, int D) { return h->prob.D == D && h->prob.H == 20 && h->prob.O == 20; }
// widths the <5, 10> / <7, 10> trip counts hold without being the reference's: those instantiations with run-time widths (EX = false)
static bool small_shape(gnnx_handle h, int D) { return h->prob.D <= D && h->prob.H <= 20 && h->prob.O <= 20; }
// ... and hidden / output widths up to 32 with the reference's input width (node mode): <5, 16> - the D-wide arrays stay short
static bool wide_shape(gnnx_handle h, int D) { return h->prob.D <= D && h->prob.H <= 32 && h->prob.O <= 32; }
// Round 6: the hidden widths people train the reference's node encoder with besides its default 20 - --hidden-dim = --output-dim
// = 16 or 32 on 10 input features - have compile-time-width instantiations of the node-mode resident kernels too (<5, 8> / <5, 16>, general and
// algebraic constant-feature form): 16 / 228 B of scratch per lane under the mixed kernel's register cap instead of 396 / 784 B with run-time
// widths, and constant feature rows take the algebraic form there as well.  (64 exceeds the 32-column tile of the row-local MFMA parts: torch route.)
static bool exact_hidden(gnnx_handle h, int H) { return h->prob.D == 10 && h->prob.H == H && h->prob.O == H; }
static bool exact_node_shape(gnnx_handle h) { return exact_hidden(h, 20) || exact_hidden(h, 16) || exact_hidden(h, 32); }

template <int NT>
static void launch_sparse_nt(gnnx_handle h, const Params& p, const int32_t* ids, int cnt, const float* adam_tab, hipStream_t s, bool log) {
    const dim3 grid(cnt), block(NT);
    if (log) {   // the logging form (loss scalars + decision trace): exact shapes only, checked by the caller; of the form the plan runs
        if (h->prob.graph_mode) hipLaunchKernelGGL((k_sparse_resident<7, 10, true, NT, 0, true>), grid, block, 0, s, p, ids, adam_tab);
        else if (h->xconst == 2) hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT, 2, true>), grid, block, 0, s, p, ids, adam_tab);
        else hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT, 0, true>), grid, block, 0, s, p, ids, adam_tab);
        return;
    }
    if (h->prob.graph_mode) {
        if (exact_shape(h, 14)) hipLaunchKernelGGL((k_sparse_resident<7, 10, true, NT>), grid, block, 0, s, p, ids, adam_tab);
        else if (small_shape(h, 14)) hipLaunchKernelGGL((k_sparse_resident<7, 10, true, NT, 0, false, false>), grid, block, 0, s, p, ids, adam_tab);
        else hipLaunchKernelGGL((k_sparse_resident<16, 16, true, NT>), grid, block, 0, s, p, ids, adam_tab);
    } else {
        if (exact_shape(h, 10) && h->xconst == 2) hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT, 2>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_shape(h, 10) && h->xconst == 1) hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT, 1>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_shape(h, 10)) hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_hidden(h, 16) && h->xconst == 2) hipLaunchKernelGGL((k_sparse_resident<5, 8, false, NT, 2>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_hidden(h, 16)) hipLaunchKernelGGL((k_sparse_resident<5, 8, false, NT>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_hidden(h, 32) && h->xconst == 2) hipLaunchKernelGGL((k_sparse_resident<5, 16, false, NT, 2>), grid, block, 0, s, p, ids, adam_tab);
        else if (exact_hidden(h, 32)) hipLaunchKernelGGL((k_sparse_resident<5, 16, false, NT>), grid, block, 0, s, p, ids, adam_tab);
        else if (small_shape(h, 10)) hipLaunchKernelGGL((k_sparse_resident<5, 10, false, NT, 0, false, false>), grid, block, 0, s, p, ids, adam_tab);
        else if (wide_shape(h, 10)) hipLaunchKernelGGL((k_sparse_resident<5, 16, false, NT, 0, false, false>), grid, block, 0, s, p, ids, adam_tab);
        else hipLaunchKernelGGL((k_sparse_resident<16, 16, false, NT>), grid, block, 0, s, p, ids, adam_tab);
    }
}
static void launch_sparse(gnnx_handle h, const Params& p, int cls, const float* adam_tab, hipStream_t s, bool log = false) {
    if (cls == SPC_LARGE) {  // node-mode targets beyond the LDS-resident classes: row arrays in HBM / L2 (gnnx_sparse_large.hpp)
        const dim3 grid(h->n_sp[cls]), block(SPL_THREADS);
        if (log)      // the logging form (loss scalars + decision trace): exact shapes only, checked by the caller
            hipLaunchKernelGGL((k_sparse_large<5, 10, true>), grid, block, 0, s, p, h->d_sp[cls], adam_tab, h->d_csr_rowptr,
                               h->d_csr_col, h->d_csr_row, h->d_csr_off, h->d_nnz + 2 * h->prob.num_targets, XlIo{});
        else if (exact_shape(h, 10))
            hipLaunchKernelGGL((k_sparse_large<5, 10>), grid, block, 0, s, p, h->d_sp[cls], adam_tab, h->d_csr_rowptr,
                               h->d_csr_col, h->d_csr_row, h->d_csr_off, h->d_nnz + 2 * h->prob.num_targets, XlIo{});
        else if (small_shape(h, 10))
            hipLaunchKernelGGL((k_sparse_large<5, 10, false, false>), grid, block, 0, s, p, h->d_sp[cls], adam_tab, h->d_csr_rowptr,
                               h->d_csr_col, h->d_csr_row, h->d_csr_off, h->d_nnz + 2 * h->prob.num_targets, XlIo{});
        else if (wide_shape(h, 10))
            hipLaunchKernelGGL((k_sparse_large<5, 16, false, false>), grid, block, 0, s, p, h->d_sp[cls], adam_tab, h->d_csr_rowptr,
                               h->d_csr_col, h->d_csr_row, h->d_csr_off, h->d_nnz + 2 * h->prob.num_targets, XlIo{});
        else
            hipLaunchKernelGGL((k_sparse_large<16, 16>), grid, block, 0, s, p, h->d_sp[cls], adam_tab, h->d_csr_rowptr,
                               h->d_csr_col, h->d_csr_row, h->d_csr_off, h->d_nnz + 2 * h->prob.num_targets, XlIo{});
        return;
    }
    if (cls == SPC_512) launch_sparse_nt<512>(h, p, h->d_sp[cls], h->n_sp[cls], adam_tab, s, log);
    else if (cls == 0) launch_sparse_nt<1024>(h, p, h->d_sp[0], h->n_sp[0], adam_tab, s, log);
    else if (cls == 1) launch_sparse_nt<256>(h, p, h->d_sp[1], h->n_sp[1], adam_tab, s, log);
    else launch_sparse_nt<64>(h, p, h->d_sp[2], h->n_sp[2], adam_tab, s, log);
}

extern "C" int gnnx_set_trace(gnnx_handle h, uint32_t* gates, int32_t* pool_rows) {
    if (!h) return fail("null argument");
    h->trace_gates = gates;
    h->trace_pool = pool_rows;
    return 0;
}

// work list of k_dead_entries: every ld x ld block of the edge-sparse resident targets, DEAD_THREADS x DEAD_Q entries per workgroup
static int build_dead_list(gnnx_handle h) {
    if (h->d_dead) return 0;
    std::vector<DeadBlock> blocks;
    const long long per = (long long)DEAD_THREADS * DEAD_Q;
    for (int t = 0; t < h->prob.num_targets; ++t) {
        const int c = h->cat[t];
        if (!(c == CAT_SPARSE || c == CAT_SPARSE + 1 || c == CAT_SPARSE + 2 || c == CAT_SPARSE + SPC_512 || c == CAT_SPARSE + SPC_LARGE)) continue;
        const long long q = (long long)h->meta[t].ld * h->meta[t].ld;
        for (long long f = 0; f < q; f += per) blocks.push_back(DeadBlock{t, 0, f});
    }
    h->n_dead = (int)blocks.size();
    if (!h->n_dead) return 0;
    HIPCK(pool_malloc(&h->d_dead, sizeof(DeadBlock) * blocks.size()));
    HIPCK(upload_sync(h->d_dead, blocks.data(), sizeof(DeadBlock) * blocks.size()));
    return 0;
}

extern "C" int gnnx_run(gnnx_handle h, const gnnx_hyper* hy, const float* A, const float* X, const float* yhat,
                        float* M, float* Abar, float* feat_mask, float* loss, void* workspace,
                        size_t workspace_bytes, void* stream) {
    return gnnx_run_resume(h, hy, nullptr, A, X, yhat, M, Abar, feat_mask, loss, workspace, workspace_bytes, stream);
}

extern "C" int gnnx_set_att_weights(gnnx_handle h, const float* att_weights) {
    if (!h || !att_weights) return fail("null argument");
    if (h->prob.bn || h->prob.mask_relu) return fail("method=att is not implemented together with --bn / mask_act=ReLU");
    if (!h->d_watt) HIPCK(pool_malloc(&h->d_watt, sizeof(float) * 3 * 1024));
    HIPCK(upload_sync(h->d_watt, att_weights, sizeof(float) * 3 * 1024));
    return 0;
}

// method="att": every target in k_att (gnnx_att.hpp).  The edge arrays are sized from a count of the batch's directed entries, so this
// path synchronises with the host once before and once after the launch - it is the complete path for a rarely used flag, not a
// pipelined one.
static int run_att(gnnx_handle h, const gnnx_hyper* hy, const gnnx_resume& rs, Params p, const float* A, float* feat_mask, hipStream_t s) {
    const int T = h->prob.num_targets;
    if (p.loss) return fail("loss logging is not implemented for method=att");
    int32_t* d_cnt = nullptr;
    HIPCK(pool_malloc(&d_cnt, sizeof(int32_t) * T));
    hipLaunchKernelGGL(k_att_count, dim3(T), dim3(256), 0, s, h->d_meta, A, d_cnt);
    std::vector<int32_t> cnt(T);
    HIPCK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int32_t) * T, hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    (void)pool_free(d_cnt);
    std::vector<long long> eoff(T + 1, 0);
    for (int t = 0; t < T; ++t) eoff[t + 1] = eoff[t] + cnt[t];
    const size_t E = (size_t)std::max<long long>(eoff[T], 1), R = (size_t)h->R;
    // one allocation, carved: edge arrays (col, mir, w, s[3], q, dA), row arrays (xin[3], u[3], U[3], dZ, dX, rn[3]), offsets, row pointers
    size_t o = 0;
    auto take = [&](size_t bytes) {
        const size_t r = o;
        o = align_up(o + bytes, 256);
        return r;
    };
    const size_t o_eoff = take(sizeof(long long) * (T + 1)), o_rowptr = take(sizeof(int32_t) * (R + T)), o_col = take(4 * E), o_mir = take(4 * E);
    const size_t o_wgorder = take(sizeof(int32_t) * T);      // workgroup -> target, longest chains (most edges) first
    size_t o_e[6], o_r[11], o_rn[3];
    for (auto& x : o_e) x = take(4 * E);
    for (auto& x : o_r) x = take(4 * R * FS);
    for (auto& x : o_rn) x = take(4 * R);
    // round 5 (hop pruning, chunked rows): levels, rows by level, chunk table, rows of several chunks, their partial sums
    const size_t NCH = R + (size_t)T + E / 32 + (size_t)T, NMU = E / 32 + (size_t)T;
    const size_t o_lev = take(4 * R), o_order = take(4 * (R + T)), o_chunk = take(4 * NCH), o_mrow = take(4 * NMU), o_mfirst = take(4 * NMU),
                 o_pacc = take(4 * NCH * 32);
    char* d = nullptr;
    HIPCK(pool_malloc(&d, o));
    // rows / entries a pruned phase does not write are READ (times an exact zero) by its neighbours: no NaN patterns of a recycled block
    HIPCK(hipMemsetAsync(d, 0, o, s));
    HIPCK(hipStreamSynchronize(s));      // (the uploads below go through the service stream)
    float* d_tab = nullptr;
    std::vector<float> tab(2 * (size_t)hy->num_iters);
    for (int it = 0; it < hy->num_iters; ++it) adam_scalars(hy, rs.first_iter + it, &tab[2 * it], &tab[2 * it + 1], it);
    hipError_t e = pool_malloc(&d_tab, sizeof(float) * tab.size());
    if (e == hipSuccess) e = upload_sync(d_tab, tab.data(), sizeof(float) * tab.size());
    if (e == hipSuccess) e = upload_sync(d + o_eoff, eoff.data(), sizeof(long long) * (T + 1));
    std::vector<int32_t> wg_order(T);
    for (int t = 0; t < T; ++t) wg_order[t] = t;
    std::stable_sort(wg_order.begin(), wg_order.end(), [&](int x, int y) { return cnt[x] > cnt[y]; });
    if (e == hipSuccess) e = upload_sync(d + o_wgorder, wg_order.data(), sizeof(int32_t) * T);
    if (e != hipSuccess) {
        (void)pool_free(d);
        if (d_tab) (void)pool_free(d_tab);
        return fail(std::string("method=att set-up: ") + hipGetErrorString(e));
    }
    AttScratch a{};
    a.watt = h->d_watt;
    a.eoff = reinterpret_cast<const long long*>(d + o_eoff);
    a.rowptr = reinterpret_cast<int32_t*>(d + o_rowptr);
    a.col = reinterpret_cast<int32_t*>(d + o_col);
    a.mir = reinterpret_cast<int32_t*>(d + o_mir);
    a.w = reinterpret_cast<float*>(d + o_e[0]);
    for (int l = 0; l < 3; ++l) a.s[l] = reinterpret_cast<float*>(d + o_e[1 + l]);
    a.q = reinterpret_cast<float*>(d + o_e[4]);
    a.dA = reinterpret_cast<float*>(d + o_e[5]);
    for (int l = 0; l < 3; ++l) {
        a.xin[l] = reinterpret_cast<float*>(d + o_r[l]);
        a.u[l] = reinterpret_cast<float*>(d + o_r[3 + l]);
        a.U[l] = reinterpret_cast<float*>(d + o_r[6 + l]);
        a.rn[l] = reinterpret_cast<float*>(d + o_rn[l]);
    }
    a.dZ = reinterpret_cast<float*>(d + o_r[9]);
    a.dX = reinterpret_cast<float*>(d + o_r[10]);
    a.lev = reinterpret_cast<int32_t*>(d + o_lev);
    a.order = reinterpret_cast<int32_t*>(d + o_order);
    a.chunk = reinterpret_cast<int32_t*>(d + o_chunk);
    a.mrow = reinterpret_cast<int32_t*>(d + o_mrow);
    a.mfirst = reinterpret_cast<int32_t*>(d + o_mfirst);
    a.pacc = reinterpret_cast<float*>(d + o_pacc);
    hipLaunchKernelGGL(k_att, dim3(T), dim3(ATT_THREADS), 0, s, p, a, d_tab, reinterpret_cast<const int32_t*>(d + o_wgorder));
    if (feat_mask)
        (void)hipMemcpyAsync(feat_mask, p.f[hy->num_iters & 1], sizeof(float) * T * FS, hipMemcpyDeviceToDevice, s);
    e = hipStreamSynchronize(s);
    (void)pool_free(d);
    (void)pool_free(d_tab);
    if (e != hipSuccess) return fail(std::string("method=att run: ") + hipGetErrorString(e));
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int gnnx_run_resume(gnnx_handle h, const gnnx_hyper* hy, const gnnx_resume* resume, const float* A, const float* X,
                               const float* yhat, float* M, float* Abar, float* feat_mask, float* loss, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!h || !hy || !A || !X || !M || !Abar || !workspace) return fail("null argument");
    gnnx_resume rs;
    std::memset(&rs, 0, sizeof rs);   // (padding included: the struct is part of the hipGraph key)
    if (resume) {
        rs.first_iter = resume->first_iter;
        rs.m = resume->m;
        rs.v = resume->v;
        rs.feat = resume->feat;
        rs.m_out = resume->m_out;
        rs.v_out = resume->v_out;
        rs.feat_out = resume->feat_out;
    }
    if (rs.first_iter < 0) return fail("first_iter must be >= 0");
    if (!h->prob.graph_mode && !yhat) return fail("yhat is required in node mode (Laplacian term)");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    if (hy->num_iters < 1) return fail("num_iters must be >= 1");
    if (hy->opt < 0 || hy->opt > 3) return fail("opt must be 0 (Adam), 1 (SGD), 2 (RMSprop) or 3 (Adagrad)");
    if (hy->lr_schedule && hy->use_graph) return fail("a learning-rate schedule cannot be captured into a hipGraph (use_graph = 0)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* lossp = hy->record_loss ? loss : nullptr;
    if (int rc = use_tables(h, s)) return rc;
    if (hy->record_loss && !loss) return fail("record_loss set but loss buffer is null");
    if (hy->record_loss && h->prob.mask_relu) return fail("loss logging is not implemented for mask_act = ReLU (the reference's loss is NaN there)");
    Params p = make_params(h, hy, A, X, yhat, M, Abar, lossp, workspace);
    p.m_in = rs.m;
    p.v_in = rs.v;
    p.fs_in = rs.feat;
    p.m_out = rs.m_out;
    p.v_out = rs.v_out;
    p.fs_out = rs.feat_out;
    if (h->d_watt) return run_att(h, hy, rs, p, A, feat_mask, s);
    if (rs.m_out) p.mM = rs.m_out;   // the streaming kernels keep their moments in the caller's arrays when it wants them back
    if (rs.v_out) p.vM = rs.v_out;
    // hybrid split: small targets (<= res_nbmax row blocks) -> on-chip-resident kernels on side streams (overlap with the
    // streaming launches of the other targets); loss logging is a streaming-path feature
    // Logging runs (loss scalars, decision trace) stay on the edge-sparse resident kernel - its LOG form plus k_dead_entries for the
    // entries off the edges - when EVERY target of the plan is routed there and the encoder has the reference's widths (the usual case:
    // explainer_main.py --explain-node / the default node list, explain.py:149-159); any other plan logs on the dense streaming kernels.
    const bool want_trace = h->trace_gates || h->trace_pool;
    // (round 5: k_sparse_large has a logging form too, so a logging run leaves the edge kernels only for targets that stream)
    const bool log_resident = (lossp || want_trace) && hy->use_resident && h->n_res == 0 && h->n_big == 0 &&
                              h->n_sparse() > 0 && exact_shape(h, h->prob.graph_mode ? 14 : 10);
    if (want_trace && !log_resident)
        return fail("gnnx_set_trace: the decision trace is recorded by the sparse on-chip-resident kernel only (every target routed there, "
                    "D = 10 / 14, H = O = 20, use_resident = 1)");
    p.trace_gates = h->trace_gates;
    p.trace_pool = h->trace_pool;
    p.trace_rows = h->R;
    const bool resident = hy->use_resident && (h->n_res > 0 || h->n_sparse() > 0) && (!lossp || log_resident);
    const bool streaming = !resident || h->n_big > 0;
    if (streaming)
        if (int rc = (resident && h->n_big > 0) ? ensure_mask_big(h) : ensure_mask_all(h)) return rc;
    const Tables tb = (resident && h->n_big > 0) ? tables_big(h) : tables_all(h);
    if (streaming)
        if (int rc = init_stream_state(h, p, s)) return rc;
    if (resident) {
        if (!h->d_adam || hy->lr_schedule || h->adam_first != rs.first_iter || std::memcmp(&h->adam_for, hy, sizeof(gnnx_hyper)) != 0) {
            if (h->d_adam && !h->adam_shared) {
                HIPCK(hipStreamSynchronize(s));   // an earlier run of this plan may still be reading the table (its side lanes join `s`)
                (void)pool_free(h->d_adam);
            }
            h->d_adam = nullptr;
            h->adam_shared = false;
            h->adam_host.resize(2 * (size_t)hy->num_iters);
            for (int it = 0; it < hy->num_iters; ++it)
                adam_scalars(hy, rs.first_iter + it, &h->adam_host[2 * it], &h->adam_host[2 * it + 1], it);
            // The table depends on the hyper-parameters alone: the plans of a long job (one per batch) share ONE device copy, uploaded by the
            // first of them - no per-batch upload on the launching thread.  (Runs with a learning-rate schedule keep a private table.)
            if (!hy->lr_schedule) h->d_adam = shared_adam_table(hy, rs.first_iter, h->adam_host);
            if (h->d_adam) {
                h->adam_shared = true;
            } else {
                HIPCK(pool_malloc(&h->d_adam, sizeof(float) * h->adam_host.size()));
                HIPCK(upload_sync(h->d_adam, h->adam_host.data(), sizeof(float) * h->adam_host.size()));
            }
            h->adam_for = *hy;
            h->adam_first = rs.first_iter;
        }
        if (log_resident) {
            const size_t T = (size_t)h->prob.num_targets;
            if (p.trace_gates) HIPCK(hipMemsetAsync(p.trace_gates, 0, sizeof(uint32_t) * 2 * (size_t)h->R * hy->num_iters, s));
            if (p.trace_pool) HIPCK(hipMemsetAsync(p.trace_pool, 0xff, sizeof(int32_t) * 96 * T * hy->num_iters, s));
            if (lossp) {   // the entries off the edges first (they add to the size / entropy columns; the resident launch completes the rows)
                HIPCK(hipMemsetAsync(lossp, 0, sizeof(float) * T * hy->num_iters * NLOSS, s));
                if (int rc = build_dead_list(h)) return rc;
                if (h->n_dead) hipLaunchKernelGGL(k_dead_entries, dim3(h->n_dead), dim3(DEAD_THREADS), 0, s, p, h->d_dead, h->d_adam);
            }
        }
        HIPCK(hipEventRecord(h->ev_in, s));
        for (int k = 0; k <= N_SIDE; ++k) h->launched[k] = false;      // (which groups launch can change between runs: a class may ride in the mixed launch)
        // Launch order: sparse resident kernel (gnnx_plan_analyze), largest size class FIRST - its workgroups need a whole
        // CU (1024 threads x 128 VGPRs), so they must be placed before the small workgroups of the other launches spread
        // over every CU (measured on syn1: 21.5 -> 13.4 ms) - then the dense resident kernels.  Every group has its own
        // side stream.  (Running the last group on the caller's stream instead of a side stream serialised it behind
        // another group, and a 40-200 us delay kernel in front of the small launches changed nothing - measured, not
        // used; see gnnx_plan_analyze for which groups are allowed to meet.)
        // every group of this run takes the next lane (lane_stream above); a fourth group shares the first one's
        // consecutive runs start on consecutive lanes, so that the launches of two batches in flight (pipeline.BatchPipeline keeps up to
        // three optimisations going: a syn1 batch fills 141 of 256 CUs and lasts as long as its slowest workgroup) do not queue behind each
        // other on one stream; a run with several groups still takes consecutive lanes from there
        // node-mode batches of 512-thread and single-tile (64-thread class) targets: ONE launch (k_sparse_resident_mixed)
        // (with pair workgroups the 256-thread class rides in the same launch, two targets to a workgroup: h->pair256)
        const bool pairs = !h->prob.graph_mode && h->pair256 && h->n_sp[1] > 0;
        // the packed single-wave targets lead d_sp[2] and take their own launch (below); the logging forms keep the classes' own bodies
        const int n_pack = (h->tiny_per_cu && !log_resident && h->xconst == 2 && exact_shape(h, 10)) ? h->n_tiny_pack : 0;
        const int32_t* tiny_ids = h->d_sp[2] ? h->d_sp[2] + n_pack : nullptr;
        const int n_tiny = h->n_sp[2] - n_pack;
        const bool mixed = !h->prob.graph_mode && ((h->n_sp[SPC_512] > 0 && h->n_sp[2] > 0) || pairs);
        const int mixed_at = !mixed ? -1 : h->n_sp[SPC_512] > 0 ? SPC_512 : pairs ? 1 : 2;     // the class whose turn in the launch order starts the mixed launch
        auto own_launch = [&](int k) {
            if (k == 2 && n_tiny == 0 && (!mixed || mixed_at != 2)) return false;      // the whole class went to the packed launch
            if (mixed && k == mixed_at && h->n_sp[SPC_512] == 0 && !pairs && n_tiny == 0) return false;   // (a mixed launch of single-wave targets only, all packed)
            return h->n_sp[k] > 0 && (!mixed || k == mixed_at || !(k == 2 || k == SPC_512 || (pairs && k == 1)));
        };
        // A run whose targets all sit in ONE launch group (syn1 / syn4 / syn5: the mixed launch; config 4: the 256-thread class) needs no
        // side lane at all - nothing has to overlap inside the run: it goes to the caller's stream.  Fewer busy streams = fewer
        // hardware queues (HIP has eight at most) for the streams of a pipelined job to collide with.
        int n_groups = n_pack > 0;
        for (int k = 0; k < N_SPC; ++k) n_groups += own_launch(k);
        for (int nb = 1; nb <= RES_NBMAX; ++nb) n_groups += h->res_count[nb] > 0;
        const bool single_group = n_groups == 1 && !streaming;
        static std::atomic<unsigned> g_lane_base{0};
        int next_lane = single_group ? 0 : (int)(g_lane_base.fetch_add(1) % N_LANES);
        auto group_stream = [&](int) -> hipStream_t {
            if (single_group) return s;
            hipStream_t st = lane_stream(next_lane++);
            return st ? st : s;
        };
        const int launch_order[N_SPC] = {SPC_LARGE, 0, SPC_512, 1, 2};   // the longest workgroups first, then the other whole-CU ones
        bool packed_done = n_pack == 0;
        auto launch_packed = [&]() -> void {
            if (packed_done) return;
            packed_done = true;
            hipStream_t ss = group_stream(N_SIDE);
            if (ss != s) (void)hipStreamWaitEvent(ss, h->ev_in, 0);
            (void)hipEventRecord(h->ev_t0[N_SIDE], ss);
            h->launched[N_SIDE] = true;
            if (h->tiny_per_cu == 16)
                hipLaunchKernelGGL((k_sparse_resident_tiny16<5, 10, 2>), dim3((n_pack + 15) / 16), dim3(1024), 0, ss, p, h->d_sp[2], n_pack, h->d_adam);
            else
                hipLaunchKernelGGL((k_sparse_resident_tiny12<5, 10, 2>), dim3((n_pack + 11) / 12), dim3(768), 0, ss, p, h->d_sp[2], n_pack, h->d_adam);
            (void)hipEventRecord(h->ev_out[N_SIDE], ss);
        };
        for (int ko = 0; ko < N_SPC; ++ko) {
            const int k = launch_order[ko];
            if (!own_launch(k)) continue;
            const int g = RES_NBMAX + k;      // (the mixed launch is timed in the slot of the class that starts it: 512 threads, else 256)
            hipStream_t ss = group_stream(g);
            if (ss != s) HIPCK(hipStreamWaitEvent(ss, h->ev_in, 0));
            HIPCK(hipEventRecord(h->ev_t0[g], ss));
            h->launched[g] = true;
            if (mixed && k == mixed_at) {
                const int per_wg = sp_mix_tiny(h->prob.D, h->prob.H, h->prob.C);   // single-tile targets per workgroup
                const int n_pair = pairs ? h->n_sp[1] : 0;
                const int32_t* pair_ids = pairs ? h->d_sp[1] : nullptr;
                const dim3 grid(h->n_sp[SPC_512] + (n_pair + 1) / 2 + (n_tiny + per_wg - 1) / per_wg), block(512);
                if (log_resident && h->xconst == 2)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10, 2, true>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (log_resident)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10, 0, true>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_shape(h, 10) && h->xconst == 2)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10, 2>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_shape(h, 10) && h->xconst == 1)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10, 1>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_shape(h, 10))
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_hidden(h, 16) && h->xconst == 2)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 8, 2>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_hidden(h, 16))
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 8>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_hidden(h, 32) && h->xconst == 2)
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 16, 2>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (exact_hidden(h, 32))
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 16>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (small_shape(h, 10))
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 10, 0, false, false>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else if (wide_shape(h, 10))
                    hipLaunchKernelGGL((k_sparse_resident_mixed<5, 16, 0, false, false>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
                else
                    hipLaunchKernelGGL((k_sparse_resident_mixed<16, 16>), grid, block, 0, ss, p, h->d_sp[SPC_512], h->n_sp[SPC_512],
                                       tiny_ids, n_tiny, h->d_adam, per_wg, sp_model_floats(h->prob.D, h->prob.H, h->prob.C), pair_ids, n_pair);
            } else if (k == 2) {
                launch_sparse_nt<64>(h, p, tiny_ids, n_tiny, h->d_adam, ss, log_resident);
            } else {
                launch_sparse(h, p, k, h->d_adam, ss, log_resident);
            }
            HIPCK(hipEventRecord(h->ev_out[g], ss));
            if (k == SPC_512 || (mixed && k == mixed_at)) launch_packed();   // behind the whole-CU workgroups, ahead of nothing that needs a whole CU
        }
        launch_packed();
        for (int nb = RES_NBMAX; nb >= 1; --nb) {  // largest targets first
            if (!h->res_count[nb]) continue;
            hipStream_t ss = group_stream(nb - 1);
            if (ss != s) HIPCK(hipStreamWaitEvent(ss, h->ev_in, 0));
            HIPCK(hipEventRecord(h->ev_t0[nb - 1], ss));
            h->launched[nb - 1] = true;
            const dim3 grid(h->res_count[nb]), block(256);
            const int32_t* ids = h->d_res + h->res_first[nb];
            if (nb == 1) hipLaunchKernelGGL(k_resident<1>, grid, block, 0, ss, p, ids, h->d_adam);
            else if (nb == 2) hipLaunchKernelGGL(k_resident<2>, grid, block, 0, ss, p, ids, h->d_adam);
            else hipLaunchKernelGGL(k_resident<3>, grid, block, 0, ss, p, ids, h->d_adam);
            HIPCK(hipEventRecord(h->ev_out[nb - 1], ss));
        }
    }
    if (streaming) {
        if (!hy->use_graph) {
            int rc = enqueue_job(h, tb, hy, p, rs.first_iter, s);
            if (rc) return rc;
        } else {
            GraphKey key;
            std::memset(&key, 0, sizeof key);
            key.A = A; key.X = X; key.yhat = yhat; key.M = M; key.Abar = Abar; key.fm = (const void*)(size_t)(resident ? 1 : 0);
            key.loss = lossp; key.ws = workspace; key.rs = rs; key.hy = *hy;
            if (!h->gexec || !(key == h->gkey)) {
                if (h->gexec) {
                    (void)hipGraphExecDestroy(h->gexec);
                    h->gexec = nullptr;
                }
                hipGraph_t g = nullptr;
                HIPCK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
                int rc = enqueue_job(h, tb, hy, p, rs.first_iter, s);
                hipError_t e = hipStreamEndCapture(s, &g);
                if (rc) return rc;
                if (e != hipSuccess) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                e = hipGraphInstantiate(&h->gexec, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) {
                    h->gexec = nullptr;
                    return fail(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
                }
                h->gkey = key;
            }
            HIPCK(hipGraphLaunch(h->gexec, s));
        }
    }
    if (resident) {
        for (int k = 0; k < RES_NBMAX; ++k)
            if (h->res_count[k + 1]) HIPCK(hipStreamWaitEvent(s, h->ev_out[k], 0));
        for (int k = 0; k < N_SPC; ++k)
            if (h->n_sp[k] && h->launched[RES_NBMAX + k]) HIPCK(hipStreamWaitEvent(s, h->ev_out[RES_NBMAX + k], 0));
        if (h->launched[N_SIDE]) HIPCK(hipStreamWaitEvent(s, h->ev_out[N_SIDE], 0));
    }
    if (feat_mask)
        HIPCK(hipMemcpyAsync(feat_mask, p.f[hy->num_iters & 1], sizeof(float) * h->prob.num_targets * FS,
                             hipMemcpyDeviceToDevice, s));
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_tiny_pack_info(gnnx_handle h, int32_t* per_cu, int32_t* n_packed, int32_t* flags) {
    if (!h) return fail("null argument");
    const bool on = h->tiny_per_cu && h->n_tiny_pack > 0;
    if (per_cu) *per_cu = on ? h->tiny_per_cu : 0;
    if (n_packed) *n_packed = on ? h->n_tiny_pack : 0;
    if (flags)
        for (int t = 0; t < h->prob.num_targets; ++t)
            flags[t] = on && t < (int)h->tiny_pack.size() && h->tiny_pack[t] && h->cat[t] == CAT_SPARSE + 2;
    return 0;
}

extern "C" int gnnx_resident_times(gnnx_handle h, float* ms) {
    if (!h || !ms) return fail("null argument");
    for (int k = 0; k < N_SIDE; ++k) {
        ms[k] = 0.0f;
        if (!h->launched[k]) continue;
        HIPCK(hipEventSynchronize(h->ev_out[k]));
        HIPCK(hipEventElapsedTime(&ms[k], h->ev_t0[k], h->ev_out[k]));
    }
    if (h->launched[N_SIDE]) {   // the packed single-wave launch: reported in the 64-thread class's slot (it runs beside that class's other launch, if any)
        float t = 0.0f;
        HIPCK(hipEventSynchronize(h->ev_out[N_SIDE]));
        HIPCK(hipEventElapsedTime(&t, h->ev_t0[N_SIDE], h->ev_out[N_SIDE]));
        ms[RES_NBMAX + 2] = std::max(ms[RES_NBMAX + 2], t);
    }
    return 0;
}

extern "C" int gnnx_get_route(gnnx_handle h, int32_t* route) {
    if (!h || !route) return fail("null argument");
    for (int t = 0; t < h->prob.num_targets; ++t) route[t] = h->cat[t];
    return 0;
}

extern "C" int gnnx_plan_analyze(gnnx_handle h, const float* A, void* stream) { return gnnx_plan_analyze_features(h, A, nullptr, stream); }

static int analyze_impl(gnnx_handle h, const float* A, const float* X, hipStream_t s, bool rows_from_pack);
extern "C" int gnnx_plan_analyze_features(gnnx_handle h, const float* A, const float* X, void* stream) {
    if (!h || !A) return fail("null argument");
    return analyze_impl(h, A, X, static_cast<hipStream_t>(stream), false);
}
// rows_from_pack: k_pack has just written every row's off-diagonal and upper-triangle counts into d_rowdeg / d_rowcnt
// (gnnx_pack_csr_analyze): k_row_degrees and k_edge_rowcount are not launched
static int analyze_impl(gnnx_handle h, const float* A, const float* X, hipStream_t s, bool rows_from_pack) {
    const int T = h->prob.num_targets;
    const bool look_at_x = X && !h->prob.graph_mode;   // constant feature rows: node mode (the form exists for the node encoder's shapes)
    if (int rc = use_tables(h, s, /*need_split=*/false)) return rc;
    // device block of the analysis: [T] int64 upper-triangle edge counts (the edge layout of the results, gnnx_edge_layout), then the int32 figures
    // of the routing below - ONE copy back for both (every host-blocking round trip of a preparing thread waits for a copy kernel to
    // find a free compute unit while the optimisations of the batches ahead fill the chip)
    if (!h->d_nnz) {
        char* blk = nullptr;
        HIPCK(pool_malloc(&blk, sizeof(int64_t) * T + sizeof(int32_t) * (3 + SPL_COUNTS) * T));
        h->d_nnz = reinterpret_cast<int32_t*>(blk + sizeof(int64_t) * T);
    }
    int64_t* d_ecount = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(h->d_nnz) - sizeof(int64_t) * T);
    if (!h->d_rowcnt) HIPCK(pool_malloc(&h->d_rowcnt, sizeof(int32_t) * (size_t)h->R));
    if (rows_from_pack) {   // degrees and upper-triangle counts of every row are in d_rowdeg / d_rowcnt: one per-target launch does the rest
        hipLaunchKernelGGL(k_count_edges, dim3(T), dim3(256), 0, s, h->d_meta, A, h->d_nnz, look_at_x ? X : nullptr,
                           look_at_x ? h->d_nnz + (2 + SPL_COUNTS) * (size_t)T : nullptr, (const int32_t*)h->d_rowdeg, h->d_rowcnt, d_ecount);
    } else {
        edge_rows(h, A, h->d_rowcnt, d_ecount, s);
        hipLaunchKernelGGL(k_count_edges, dim3(T), dim3(256), 0, s, h->d_meta, A, h->d_nnz, look_at_x ? X : nullptr,
                           look_at_x ? h->d_nnz + (2 + SPL_COUNTS) * (size_t)T : nullptr, (const int32_t*)nullptr, (int32_t*)nullptr,
                           (int64_t*)nullptr);
    }
    if (!h->prob.graph_mode) {
        int nmax = 0;
        for (int t = 0; t < T; ++t) nmax = std::max(nmax, h->meta[t].n);
        if (!h->d_rowdeg) HIPCK(pool_malloc(&h->d_rowdeg, sizeof(int32_t) * (size_t)h->R));
        if (!rows_from_pack) hipLaunchKernelGGL(k_row_degrees, dim3(h->n_conv), dim3(256), 0, s, A, h->d_conv, h->d_rowdeg);
        hipLaunchKernelGGL((k_count_edges_large<0, 512, 256>), dim3(T), dim3(256), 0, s, h->d_meta, A, h->d_rowdeg, h->d_nnz + 2 * T);
        if (nmax > 512)
            hipLaunchKernelGGL((k_count_edges_large<512, 4095>), dim3(T), dim3(1024), 0, s, h->d_meta, A, h->d_rowdeg, h->d_nnz + 2 * T);
        if (nmax > 4095)
            hipLaunchKernelGGL((k_count_edges_large<4095, SPL_N_MAX>), dim3(T), dim3(1024), 0, s, h->d_meta, A, h->d_rowdeg,
                               h->d_nnz + 2 * T);
    }
    HIPCK(hipGetLastError());
    // per target: (directed entries, row slots over all rows); then k_count_edges_large's SPL_COUNTS figures
    h->nnz.assign((3 + SPL_COUNTS) * (size_t)T, -1);   // ..., then one flag per target: constant feature rows (0 when X was not given)
    {
        const size_t ints = (size_t)(h->prob.graph_mode ? 2 : (look_at_x ? 3 : 2) + SPL_COUNTS) * T;   // the written prefix of the int32 part
        std::vector<char> back(sizeof(int64_t) * T + sizeof(int32_t) * ints);
        HIPCK(hipMemcpyAsync(back.data(), d_ecount, back.size(), hipMemcpyDeviceToHost, s));
        HIPCK(hipStreamSynchronize(s));
        h->edge_counts.resize(T);
        std::memcpy(h->edge_counts.data(), back.data(), sizeof(int64_t) * T);
        std::memcpy(h->nnz.data(), back.data() + sizeof(int64_t) * T, sizeof(int32_t) * ints);
    }
    h->xconst = 0;
    const bool graph = h->prob.graph_mode != 0;
    if (h->prob.C > RES_CMAX || h->prob.mask_relu || h->prob.bn) return 0;   // mask_act = "ReLU" and --bn run on the dense streaming kernels only
    int sparse_on = 1;
    if (const char* env = std::getenv("GNNX_SPARSE_RESIDENT")) sparse_on = std::atoi(env);
    if (!sparse_on) return 0;
    // every target takes the smallest size class of the sparse resident kernel its rows / edges / LDS fit (64 threads:
    // n <= 32; 256: n <= 128; 1024: n <= 512); single-tile node-mode targets that fit none keep the dense resident
    // kernel; the rest streams.  GNNX_TINY_SPARSE=0 keeps every single-tile target on the dense resident kernel.
    int tiny_on = 1, large_on = 1, c512_on = 1;
    if (const char* env = std::getenv("GNNX_SPARSE_512")) c512_on = std::atoi(env);
    if (const char* env = std::getenv("GNNX_TINY_SPARSE")) tiny_on = std::atoi(env);
    if (const char* env = std::getenv("GNNX_SPARSE_LARGE")) large_on = std::atoi(env);  // 0: those targets stream (dense)
    bool changed = false;
    // The LDS layout of the algebraic constant-feature form is smaller (sparse_layout: slim bit 0): when EVERY target the resident classes could
    // take has constant feature rows and the plan will run that form, the classes are sized with it (the kernels pick the layout by their form;
    // a plan sized with the full layout still fits: the slim one is never larger).
    int xc_form = 2;   // GNNX_XCONST = 0 / 1 / 2: general form / bit-identical constant-feature form / algebraic form (default)
    if (const char* env = std::getenv("GNNX_XCONST")) xc_form = std::max(0, std::min(2, std::atoi(env)));
    bool xc_all = look_at_x && !graph && exact_node_shape(h) && xc_form == 2;
    for (int t = 0; t < T && xc_all; ++t)
        if (h->meta[t].ld <= SP_LD_MAX) xc_all = h->nnz[(2 + SPL_COUNTS) * (size_t)T + t] == 1;
    const int lay = xc_all ? 1 : 0;
    std::vector<int> new_cat(T, 0);
    for (int t = 0; t < T; ++t) {
        const TargetMeta& m = h->meta[t];
        const int nb = m.ld / TILE;
        int c = 0;
        if (h->nnz[2 * t] >= 0)
            for (int k = 2; k >= 0 && !c; --k)
                if (sparse_fits(SPC_THREADS[k], m.n, m.ld, h->nnz[2 * t], h->nnz[2 * t + 1], h->prob.D, h->prob.H, h->prob.C,
                                graph, h->prob.O, 0, lay) &&
                    (k < 2 || tiny_on))
                    c = CAT_SPARSE + k;
        if (!graph && nb == 1 && h->res_nbmax >= 1 && (!c || !tiny_on)) c = 1;  // dense resident: node mode only
        const int* lg = &h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t];  // directed entries, slots of 64, slots of 16 (rows within two hops), ...
        // node mode, beyond the 256-thread class: the 512-thread class when the rows within two hops fit its 256 slots
        // (no scratch, cheaper barriers), else the 1024-thread class chosen above
        if (!graph && c512_on && (c == 0 || c == CAT_SPARSE) && lg[0] >= 0 &&
            sparse_fits(512, m.n, m.ld, lg[0], lg[2], h->prob.D, h->prob.H, h->prob.C, 0, h->prob.O, 0, lay))
            c = CAT_SPARSE + SPC_512;
        if (!c && !graph && large_on && lg[0] >= 0 && sparse_large_fits(m.n, m.ld, lg, h->prob.D, h->prob.H, h->prob.C))
            c = CAT_SPARSE + SPC_LARGE;
        new_cat[t] = c;
    }
    // Batches that need the 512- / 1024-thread classes keep to them and the dense single-tile kernel: measured on syn1,
    // such a launch runs back to back with a concurrent 64- or 256-thread sparse launch (9.4-11.3 ms per batch, with or
    // without scratch in the big kernel: it is the stream-to-hardware-queue mapping), while it overlaps with
    // k_resident<1> (7.3-8.1 ms).  So there the middle class joins the big one and the single-tile node-mode targets
    // stay on k_resident<1>.  Batches without such a target (syn4, syn5, graph mode) use the small classes: 1.2-2.7x
    // faster than the alternatives.
    bool has_large = false, has_1024 = false;
    for (int t = 0; t < T; ++t) {
        has_large |= (new_cat[t] == CAT_SPARSE || new_cat[t] == CAT_SPARSE + SPC_512);
        has_1024 |= (new_cat[t] == CAT_SPARSE);
    }
    int mix_on = 1, pair_on = 1;
    if (const char* env = std::getenv("GNNX_SPARSE_MIXED")) mix_on = std::atoi(env);
    if (const char* env = std::getenv("GNNX_PAIR_256")) pair_on = std::atoi(env);
    // Throughput regime: with more big workgroups than the chip has CUs nothing needs to overlap - every launch fills the
    // GPU by itself - and what counts is workgroups per CU: a 256-thread target (n <= 128) then keeps its own class (two per
    // CU, 72 KB of LDS each) instead of taking a whole CU as a 512-thread workgroup.  Measured on the BA-House x100k set:
    // see DESIGN.md.  GNNX_KEEP_256 = 0 / 1 overrides.
    int n_bigwg = 0, n256 = 0, n64 = 0;
    for (int t = 0; t < T; ++t) {
        n_bigwg += (new_cat[t] == CAT_SPARSE || new_cat[t] == CAT_SPARSE + 1 || new_cat[t] == CAT_SPARSE + SPC_512);
        n256 += new_cat[t] == CAT_SPARSE + 1;
        n64 += new_cat[t] == CAT_SPARSE + 2;
    }
    bool keep256 = n_bigwg > 2 * 256;
    if (const char* env = std::getenv("GNNX_KEEP_256")) keep256 = std::atoi(env) != 0;
    // Pair workgroups (round 5; k_sparse_resident_mixed): below that regime the 256-thread targets of a node-mode batch stay in their class and
    // run TWO to a 512-thread workgroup of the mixed launch instead of one each (GNNX_PAIR_256=0: the round-4 behaviour).  A batch without
    // larger targets (syn4, syn5) pairs them when all its workgroups then fit the chip at once - the condition of the round-4 merge below.
    bool pair256 = false;
    if (!graph && mix_on && pair_on && tiny_on && c512_on && !has_1024 && !keep256 && n256 > 0) {
        const int per_wg = sp_mix_tiny(h->prob.D, h->prob.H, h->prob.C);
        pair256 = has_large || (n64 > 0 && (n256 + 1) / 2 + (n64 + per_wg - 1) / per_wg <= 256);
    }
    // Node-mode batches of 256-thread and single-tile targets only (syn4, syn5): when all their workgroups fit the chip at
    // once, the 256-thread targets take the 512-thread class and the batch becomes ONE mixed launch - as fast as the two
    // small-class launches when those overlap (3.5 vs 3.6 ms on syn5) and not at the mercy of the queues when they do not (6.3 ms).
    if (!graph && mix_on && tiny_on && c512_on && !has_large && !pair256) {
        int n1 = 0, n2 = 0;
        bool all_fit = true;
        for (int t = 0; t < T; ++t) {
            if (new_cat[t] == CAT_SPARSE + 2) ++n2;
            if (new_cat[t] != CAT_SPARSE + 1) continue;
            ++n1;
            const TargetMeta& m = h->meta[t];
            const int* lg = &h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t];
            all_fit &= lg[0] >= 0 && sparse_fits(512, m.n, m.ld, lg[0], lg[2], h->prob.D, h->prob.H, h->prob.C, 0, h->prob.O, 0, lay);
        }
        const int per_wg = sp_mix_tiny(h->prob.D, h->prob.H, h->prob.C);
        if (n1 > 0 && n2 > 0 && all_fit && n1 + (n2 + per_wg - 1) / per_wg <= 256) {
            for (int t = 0; t < T; ++t)
                if (new_cat[t] == CAT_SPARSE + 1) new_cat[t] = CAT_SPARSE + SPC_512;
            has_large = true;
        }
    }
    // ... unless the big targets all take the 512-thread class: then they and the single-tile targets (64-thread code path,
    // eight per workgroup) share ONE launch, k_sparse_resident_mixed, and nothing needs to overlap
    const bool mixable = !graph && mix_on && tiny_on && (has_large || pair256) && !has_1024;
    h->pair256 = pair256;
    if (has_large || pair256)
        for (int t = 0; t < T; ++t) {
            if (new_cat[t] == CAT_SPARSE + 1 && !keep256 && !pair256) {  // a 256-thread target in such a batch joins the 512-thread class (else the 1024 one)
                const TargetMeta& m = h->meta[t];
                const int* lg = &h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t];
                const bool f512 = !graph && c512_on && lg[0] >= 0 &&
                                  sparse_fits(512, m.n, m.ld, lg[0], lg[2], h->prob.D, h->prob.H, h->prob.C, 0, h->prob.O, 0, lay);
                new_cat[t] = f512 ? CAT_SPARSE + SPC_512 : CAT_SPARSE;
            }
            if (new_cat[t] == CAT_SPARSE + 2 && !graph && h->res_nbmax >= 1 && !mixable) new_cat[t] = 1;
        }
    if (std::getenv("GNNX_DEBUG_ROUTE"))
        for (int t = 0; t < T; ++t)
            if (new_cat[t] == 0)
                std::fprintf(stderr, "gnnx route: target %d n=%d ld=%d streams: nnz=%d slots=%d | two hops: nnz=%d slots64=%d slots16=%d\n",
                             t, h->meta[t].n, h->meta[t].ld, h->nnz[2 * t], h->nnz[2 * t + 1], h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t],
                             h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t + 1], h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t + 2]);
    for (int t = 0; t < T; ++t) {
        changed |= (new_cat[t] != h->cat[t]);
        h->cat[t] = new_cat[t];
    }
    // constant-feature form of the sparse resident kernel: when EVERY target it takes has constant feature rows (one launch = one form)
    if (look_at_x) {
        bool all = true, any = false;
        for (int t = 0; t < T; ++t) {
            const int c = h->cat[t];
            if (c == CAT_SPARSE || c == CAT_SPARSE + 1 || c == CAT_SPARSE + 2 || c == CAT_SPARSE + SPC_512) {
                any = true;
                all &= h->nnz[(2 + SPL_COUNTS) * (size_t)T + t] == 1;
            }
        }
        h->xconst = (any && all && exact_node_shape(h)) ? xc_form : 0;
        if (h->xconst == 1 && !exact_shape(h, 10)) h->xconst = 0;      // (the bit-identical constant-feature form, a measurement knob: the reference's widths only)
    }
    // Packed single-wave launch (k_sparse_resident_tiny16 / 12): the targets of the 64-thread class whose slim LDS form fits a slice, when the
    // plan runs the algebraic constant-feature form at the reference's widths.  GNNX_TINY_PACK = 16 (default) / 12 / 0 (off).
    {
        int per_cu = 16;
        if (const char* env = std::getenv("GNNX_TINY_PACK")) per_cu = std::atoi(env);
        if (per_cu != 16 && per_cu != 12) per_cu = 0;
        const bool form_ok = per_cu && !graph && h->xconst == 2 && exact_shape(h, 10) && h->prob.C <= 4;
        std::vector<char> flags(T, 0);
        if (form_ok)
            for (int t = 0; t < T; ++t) {
                const TargetMeta& m = h->meta[t];
                flags[t] = h->cat[t] == CAT_SPARSE + 2 &&
                           sparse_fits(64, m.n, m.ld, h->nnz[2 * t], h->nnz[2 * t + 1], h->prob.D, h->prob.H, h->prob.C, 0, h->prob.O,
                                       per_cu == 16 ? tiny_pool_floats(16) : tiny_pool_floats(12), 3);
            }
        // Only where it pays: packing trades a chain's speed for chains per compute unit (measured, syn4: one wave alone on its SIMD 2.09 ms per
        // 300 iterations, two 2.6, three 3.1, four 4.0 with the spills of the 128-register build) - below about eight single-wave targets per
        // compute unit of the chip a batch runs faster SPREAD (the class's own launch / the mixed launch's slices).  GNNX_TINY_PACK_MIN overrides.
        long long min_targets = 8 * 256;
        if (const char* env = std::getenv("GNNX_TINY_PACK_MIN")) min_targets = std::atoll(env);
        long long n_flag = 0;
        for (char f : flags) n_flag += f;
        if (n_flag < min_targets) std::fill(flags.begin(), flags.end(), (char)0);
        changed |= flags != h->tiny_pack;
        h->tiny_pack.swap(flags);
        h->tiny_per_cu = (form_ok && n_flag >= min_targets) ? per_cu : 0;
    }
    if (changed || h->split_dirty) {
        if (int rc = build_split(h)) return rc;
        if (int rc = use_tables(h, s, false)) return rc;   // (the kernels below read the new lists)
    }
    // CSR of the large-class targets, built once per plan (the kernel would otherwise rescan its dense block per launch)
    if (h->n_sp[SPC_LARGE] > 0) {
        std::vector<long long> off(2 * (size_t)T, -1);   // -1: not a target of k_sparse_large (k_csr_emit_large skips its row blocks)
        long long rp = 0, cl = 0;
        for (int t = 0; t < T; ++t)
            if (h->cat[t] == CAT_SPARSE + SPC_LARGE) {
                off[2 * t] = rp;
                off[2 * t + 1] = cl;
                rp += h->meta[t].ld + 1;
                cl += (h->nnz[2 * (size_t)T + SPL_COUNTS * (size_t)t] + 1) & ~1;
            }
        for (void* ptr : {(void*)h->d_csr_rowptr, (void*)h->d_csr_col, (void*)h->d_csr_row, (void*)h->d_csr_off})
            if (ptr) (void)pool_free(ptr);
        h->d_csr_rowptr = nullptr;
        h->d_csr_col = nullptr;
        h->d_csr_row = nullptr;
        h->d_csr_off = nullptr;
        HIPCK(pool_malloc(&h->d_csr_rowptr, sizeof(int32_t) * (size_t)rp));
        HIPCK(pool_malloc(&h->d_csr_col, sizeof(unsigned short) * (size_t)std::max<long long>(cl, 2)));
        HIPCK(pool_malloc(&h->d_csr_row, sizeof(unsigned short) * (size_t)std::max<long long>(cl, 2)));
        HIPCK(pool_malloc(&h->d_csr_off, sizeof(long long) * off.size()));
        HIPCK(upload_sync(h->d_csr_off, off.data(), sizeof(long long) * off.size()));
        hipLaunchKernelGGL(k_csr_rowptr_large, dim3(h->n_sp[SPC_LARGE]), dim3(1024), 0, s, h->d_meta, h->d_rowdeg, h->d_sp[SPC_LARGE],
                           h->d_csr_off, h->d_csr_rowptr);
        hipLaunchKernelGGL(k_csr_emit_large, dim3(h->n_conv), dim3(256), 0, s, A, h->d_conv, h->d_csr_off, h->d_csr_rowptr, h->d_csr_col,
                           h->d_csr_row);
        HIPCK(hipGetLastError());
        HIPCK(hipStreamSynchronize(s));
    }
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_pack_csr(gnnx_handle h, const int64_t* indptr, const int32_t* indices, const float* weights,
                             const float* feat, int32_t feat_stride, const float* pred_label, const int32_t* nb,
                             const int64_t* nb_off, float* A, float* X, float* yhat, void* stream) {
    if (!h || !indptr || !indices || !feat || !nb || !nb_off || !A || !X) return fail("null argument");
    if (!h->prob.graph_mode && (!pred_label || !yhat)) return fail("pred_label / yhat are required in node mode");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (k_pack zeroes its own 32-row blocks of A, X and yhat first: every block of the batch belongs to one workgroup)
    if (int rc = use_tables(h, s, false)) return rc;
    PackArgs a{indptr, indices, weights, feat, feat_stride, pred_label, nb, nb_off, A, X, yhat, h->prob.D, nullptr, nullptr};
    hipLaunchKernelGGL(k_pack, dim3(h->n_conv), dim3(256), 0, s, a, h->d_conv);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

// gnnx_pack_csr + gnnx_plan_analyze_features in one call: the packing kernel counts every row's entries while it places them, so the
// analysis starts from those counts instead of rescanning the dense blocks (two launches and two passes over A fewer per batch).
extern "C" int gnnx_pack_csr_analyze(gnnx_handle h, const int64_t* indptr, const int32_t* indices, const float* weights,
                                     const float* feat, int32_t feat_stride, const float* pred_label, const int32_t* nb,
                                     const int64_t* nb_off, float* A, float* X, float* yhat, void* stream) {
    if (!h || !indptr || !indices || !feat || !nb || !nb_off || !A || !X) return fail("null argument");
    if (!h->prob.graph_mode && (!pred_label || !yhat)) return fail("pred_label / yhat are required in node mode");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = use_tables(h, s, false)) return rc;
    if (!h->d_rowcnt) HIPCK(pool_malloc(&h->d_rowcnt, sizeof(int32_t) * (size_t)h->R));
    if (!h->d_rowdeg) HIPCK(pool_malloc(&h->d_rowdeg, sizeof(int32_t) * (size_t)h->R));
    PackArgs a{indptr, indices, weights, feat, feat_stride, pred_label, nb, nb_off, A, X, yhat, h->prob.D, h->d_rowdeg, h->d_rowcnt};
    hipLaunchKernelGGL(k_pack, dim3(h->n_conv), dim3(256), 0, s, a, h->d_conv);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return analyze_impl(h, A, X, s, true);
}

extern "C" size_t gnnx_khop_scratch_bytes(int32_t num_nodes, int32_t num_targets) {
    const size_t words = ((size_t)num_nodes + 31) / 32;
    if (words <= (size_t)KH_LDS_WORDS) return 0;
    const size_t grid = (size_t)std::min(num_targets, 1024);
    return grid * 3 * words * sizeof(uint32_t);
}

extern "C" int gnnx_khop(const int64_t* indptr, const int32_t* indices, int32_t num_nodes, int32_t n_hops, const int32_t* targets,
                         int32_t num_targets, int32_t* sizes, const int64_t* nb_off, int32_t* nb, int32_t* target_row,
                         void* scratch, size_t scratch_bytes, void* stream) {
    if (!indptr || !indices || !targets || num_nodes < 1 || num_targets < 1 || n_hops < 1) return fail("bad argument");
    const bool emit = nb != nullptr;
    if (emit ? !nb_off : !sizes) return fail("null output");   // (target_row: optional in both passes)
    const int words = (num_nodes + 31) / 32;
    const bool in_lds = words <= KH_LDS_WORDS;
    if (!in_lds && (!scratch || scratch_bytes < gnnx_khop_scratch_bytes(num_nodes, num_targets))) return fail("k-hop scratch too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    KhopArgs a{indptr, indices, num_nodes, n_hops, targets, num_targets, sizes, nb_off, nb, target_row,
               static_cast<uint32_t*>(scratch), words};
    // workgroups loop over targets: GNNX_KHOP_GRID caps their number (measurement knob; a pipelined job's k-hop pass queues for
    // compute units behind the optimisations of the batches ahead; 32 / 64 / 128 workgroups instead of one per target: k-hop stage
    // 1.6-2.3 ms against 1.7 - no gain, default unchanged)
    const char* e_grid = getenv("GNNX_KHOP_GRID");
    const int cap = e_grid ? std::max(1, atoi(e_grid)) : (in_lds ? num_targets : 1024);
    const dim3 grid(std::min(num_targets, in_lds ? cap : std::min(cap, 1024))), block(KH_THREADS);
    const bool small_lds = words <= 256;      // graphs of up to 8192 nodes: 3 KB of bitmaps instead of 48
    if (emit) {
        if (small_lds) hipLaunchKernelGGL((k_khop<true, 256>), grid, block, 0, s, a);
        else if (in_lds) hipLaunchKernelGGL((k_khop<true, KH_LDS_WORDS>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_khop<true, 0>), grid, block, 0, s, a);
    } else {
        if (small_lds) hipLaunchKernelGGL((k_khop<false, 256>), grid, block, 0, s, a);
        else if (in_lds) hipLaunchKernelGGL((k_khop<false, KH_LDS_WORDS>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((k_khop<false, 0>), grid, block, 0, s, a);
    }
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int64_t gnnx_total_raw(gnnx_handle h) { return h ? h->total_raw : -1; }

extern "C" int gnnx_scatter_masks(gnnx_handle h, const float* raw, float* M, void* stream) {
    if (!h || !raw || !M) return fail("null argument");
    if (int rc = use_tables(h, static_cast<hipStream_t>(stream), false)) return rc;
    hipLaunchKernelGGL(k_scatter_masks, dim3(h->n_conv), dim3(256), 0, static_cast<hipStream_t>(stream), raw, h->d_raw_off, M, h->d_conv);
    HIPCK(hipGetLastError());
    mark_busy(h, static_cast<hipStream_t>(stream));
    return 0;
}

static int32_t* edge_scratch(gnnx_handle h, void* workspace) { return reinterpret_cast<int32_t*>(static_cast<char*>(workspace) + h->o_g3); }
// per-row counts of upper-triangle non-zeros, scanned per target (row starts stay in the scratch) -> optional per-target totals
static void edge_rows(gnnx_handle h, const float* A, int32_t* rowcnt, int64_t* counts, hipStream_t s) {
    hipLaunchKernelGGL(k_edge_rowcount, dim3(h->n_conv), dim3(256), 0, s, A, h->d_conv, rowcnt);
    hipLaunchKernelGGL(k_edge_rowscan, dim3(h->prob.num_targets), dim3(64), 0, s, h->d_meta, rowcnt, counts);
}

extern "C" int gnnx_edge_counts(gnnx_handle h, const float* A, int64_t* counts, void* stream) {
    if (!h || !A || !counts) return fail("null argument");
    // scratch for the row counts: the tail of the plan's own device tables would do, but the counts are needed before the caller
    // has a workspace only in theory - every caller has one by now; keep the ABI: a private scratch per plan
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->d_rowcnt) HIPCK(pool_malloc(&h->d_rowcnt, sizeof(int32_t) * (size_t)h->R));
    if (int rc = use_tables(h, s, false)) return rc;
    h->edge_counts.clear();   // (the cached host counts belong to the adjacency the analysis saw; this call may count another one)
    edge_rows(h, A, h->d_rowcnt, counts, s);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_gather_edges(gnnx_handle h, const float* A, const float* Abar, const float* M, const int64_t* eoff, int32_t* rc,
                                 float* abar, float* m_rc, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !A || !eoff || !rc || !workspace) return fail("null argument");
    if ((abar && !Abar) || (m_rc && !M)) return fail("values requested without their source array");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    EdgeOut o{eoff, rc, abar, m_rc, edge_scratch(h, workspace), nullptr};
    if (int rc = use_tables(h, s, false)) return rc;
    edge_rows(h, A, o.rowcnt, nullptr, s);
    hipLaunchKernelGGL(k_edge_emit, dim3(h->n_conv), dim3(256), 0, s, A, Abar, M, h->d_conv, o);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_edge_positions(gnnx_handle h, const float* A, const int64_t* eoff, int32_t* rc, int64_t* epos, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!h || !A || !eoff || !rc || !epos || !workspace) return fail("null argument");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    EdgeOut o{eoff, rc, nullptr, nullptr, edge_scratch(h, workspace), epos};
    if (int rc = use_tables(h, s, false)) return rc;
    edge_rows(h, A, o.rowcnt, nullptr, s);
    hipLaunchKernelGGL(k_edge_emit, dim3(h->n_conv), dim3(256), 0, s, A, (const float*)nullptr, (const float*)nullptr, h->d_conv, o);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_edge_counts_host(gnnx_handle h, int64_t* counts) {
    if (!h || !counts) return fail("null argument");
    if ((int)h->edge_counts.size() != h->prob.num_targets) return fail("gnnx_edge_counts_host: no analysis has counted the edges of this plan yet");
    std::memcpy(counts, h->edge_counts.data(), sizeof(int64_t) * h->edge_counts.size());
    return 0;
}

extern "C" int gnnx_edge_layout(gnnx_handle h, const float* A, const int64_t* eoff, int32_t* rc, int64_t* epos, void* stream) {
    if (!h || !A || !eoff || !rc || !epos) return fail("null argument");
    if (!h->d_rowcnt || (int)h->edge_counts.size() != h->prob.num_targets)
        return fail("gnnx_edge_layout: needs the row starts of gnnx_plan_analyze* on this adjacency (use gnnx_edge_positions otherwise)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc2 = use_tables(h, s, false)) return rc2;
    EdgeOut o{eoff, rc, nullptr, nullptr, h->d_rowcnt, epos};
    hipLaunchKernelGGL(k_edge_emit, dim3(h->n_conv), dim3(256), 0, s, A, (const float*)nullptr, (const float*)nullptr, h->d_conv, o);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_gather_values(const int64_t* epos, int64_t num_edges, const float* Abar, const float* M, float* abar, float* m_rc,
                                  void* stream) {
    if (!epos || num_edges < 0 || (abar && !Abar) || (m_rc && !M)) return fail("bad argument");
    if (num_edges == 0) return 0;
    hipLaunchKernelGGL(k_gather_values, dim3((unsigned)((num_edges + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), epos,
                       num_edges, Abar, M, abar, m_rc);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int gnnx_mt_edge_words(gnnx_handle h, const int64_t* seeds, const int64_t* eoff, const int32_t* rc, uint32_t* scratch, uint32_t* words,
                                  void* stream) {
    if (!h || !seeds || !eoff || !rc || !scratch || !words) return fail("null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc_ = use_tables(h, s, false)) return rc_;
    const int T = h->prob.num_targets;
    if (T == 0) return 0;
    hipLaunchKernelGGL(k_mt_stream, dim3(T), dim3(256), 0, s, h->d_meta, seeds, scratch);
    hipLaunchKernelGGL(k_mt_gather_pairs, dim3(T), dim3(256), 0, s, h->d_meta, eoff, rc, scratch, words);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_denoise_edges(gnnx_handle h, const int64_t* eoff, const int32_t* rc, const float* vals, int32_t threshold_num,
                                  uint8_t* keep, float* threshold, int32_t* stats, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !eoff || !rc || !vals || !keep || !threshold || !stats || !workspace) return fail("null argument");
    if (threshold_num < 1) return fail("threshold_num must be positive");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    if (int rc = use_tables(h, static_cast<hipStream_t>(stream), false)) return rc;
    char* w = static_cast<char*>(workspace);
    DenoiseArgs a{h->d_meta, eoff, rc, vals, threshold_num, keep, threshold, stats, reinterpret_cast<int32_t*>(w + h->o_g3),
                  reinterpret_cast<int32_t*>(w + h->o_z3p)};
    hipLaunchKernelGGL(k_denoise, dim3(h->prob.num_targets), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    HIPCK(hipGetLastError());
    mark_busy(h, static_cast<hipStream_t>(stream));
    return 0;
}

extern "C" int gnnx_auc_counts(const float* vals, const uint8_t* real, int64_t num_edges, float* pos_scratch, unsigned long long* counts,
                               void* stream) {
    if (!vals || !real || !pos_scratch || !counts || num_edges < 1) return fail("bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCK(hipMemsetAsync(counts, 0, 4 * sizeof(unsigned long long), s));
    const dim3 grid((unsigned)((num_edges + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_auc_compact, grid, block, 0, s, vals, real, num_edges, pos_scratch, counts);
    hipLaunchKernelGGL(k_auc_count, grid, block, 0, s, vals, real, num_edges, (const float*)pos_scratch, counts);
    HIPCK(hipGetLastError());
    return 0;
}

extern "C" int gnnx_forward(gnnx_handle h, const float* A, const float* X, const float* M, const float* feat_mask_in,
                            float* Abar, float* probs, void* workspace, size_t workspace_bytes, void* stream) {
    if (h && h->d_watt) return fail("gnnx_forward is not implemented for a method=att plan (gnnx_set_att_weights): only gnnx_run / gnnx_run_resume are");
    if (!h || !A || !X || !M || !Abar || !probs || !workspace) return fail("null argument");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = use_tables(h, s)) return rc;
    Params p = make_params(h, nullptr, A, X, nullptr, const_cast<float*>(M), Abar, nullptr, workspace);
    p.num_iters = 1;
    if (int rc = ensure_mask_all(h)) return rc;
    const Tables tb = tables_all(h);
    hipLaunchKernelGGL(k_prep, dim3(h->prob.num_targets), dim3(256), 0, s, p, feat_mask_in, (const int32_t*)nullptr);
    launch_mask<false, true>(h, tb, p, 0, 0.0f, 1.0f, s);
    launch_forward(h, tb, p, 0, s);
    HIPCK(hipMemcpyAsync(probs, p.probs, sizeof(float) * h->prob.num_targets * CMAX, hipMemcpyDeviceToDevice, s));
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_grad_baseline(gnnx_handle h, const float* A, const float* X, float* out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (h && h->d_watt) return fail("gnnx_grad_baseline is not implemented for a method=att plan (gnnx_set_att_weights): only gnnx_run / gnnx_run_resume are");
    if (!h || !A || !X || !out || !workspace) return fail("null argument");
    if (h->prob.graph_mode) return fail("the gradient baseline is a node-mode path (the reference indexes pred_label[node_idx], explain.py:130)");
    if (h->prob.bn) return fail("the gradient baseline with --bn is not implemented");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = use_tables(h, s)) return rc;
    // the streaming forward / backward with Abar := A (unmasked adjacency, diagonal included as the reference's model(x, adj))
    Params p = make_params(h, nullptr, A, X, nullptr, nullptr, const_cast<float*>(A), nullptr, workspace);
    p.num_iters = 1;
    if (int rc = ensure_mask_all(h)) return rc;
    const Tables tb = tables_all(h);
    // phi := 1: a feature-mask parameter of 40 (sigmoid(40) == 1.0f), staged in the (yet unused) df array
    const size_t nf = (size_t)h->prob.num_targets * FS;
    std::vector<float> f40(nf, 40.0f);
    float* d_f = reinterpret_cast<float*>(static_cast<char*>(workspace) + h->o_dE);   // T * 96 floats >= T * 32; rewritten by the head later
    HIPCK(hipMemcpyAsync(d_f, f40.data(), sizeof(float) * nf, hipMemcpyHostToDevice, s));
    HIPCK(hipStreamSynchronize(s));   // f40 is a host temporary
    hipLaunchKernelGGL(k_prep, dim3(h->prob.num_targets), dim3(256), 0, s, p, (const float*)d_f, (const int32_t*)nullptr);
    launch_forward(h, tb, p, 0, s);
    launch_backward(h, tb, p, 0, s);
    hipLaunchKernelGGL(k_grad_edges, dim3(tb.n_mask), dim3(256), 0, s, p, tb.mask, out);
    HIPCK(hipGetLastError());
    mark_busy(h, s);
    return 0;
}

extern "C" int gnnx_time_kernel(gnnx_handle h, const gnnx_hyper* hy, int32_t kind, int32_t reps, const float* A,
                                const float* X, const float* yhat, float* M, float* Abar, void* workspace,
                                size_t workspace_bytes, void* stream, float* ms_avg, double* alg_bytes,
                                double* alg_flops) {
    if (h && h->d_watt) return fail("gnnx_time_kernel is not implemented for a method=att plan (gnnx_set_att_weights): only gnnx_run / gnnx_run_resume are");
    if (!h || !hy || !ms_avg || reps < 1) return fail("bad argument");
    if (workspace_bytes < h->ws_bytes) return fail("workspace too small");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = use_tables(h, s)) return rc;
    Params p = make_params(h, hy, A, X, yhat, M, Abar, nullptr, workspace);
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0));
    HIPCK(hipEventCreate(&e1));
    if (kind == 8 || kind == 9) {
        // one whole launch (all num_iters iterations) of the sparse (8) / single-tile dense (9) resident kernel on
        // `stream`, alone on the device.  Algorithmic work per SURVEY.md §8d for the targets of that launch:
        // 28 n^2 bytes and 6 n^2 (D + 2H) flop per iteration (the dense formulation the reference executes).
        const int cnt = (kind == 8) ? h->n_sp[0] : h->res_count[1];  // kind 8: the largest size class
        if (cnt == 0) {
            *ms_avg = 0.0f;
            if (alg_bytes) *alg_bytes = 0.0;
            if (alg_flops) *alg_flops = 0.0;
            return 0;
        }
        std::vector<float> tab(2 * (size_t)hy->num_iters);
        for (int it = 0; it < hy->num_iters; ++it) adam_scalars(hy, it, &tab[2 * it], &tab[2 * it + 1]);
        float* d_tab = nullptr;
        HIPCK(pool_malloc(&d_tab, sizeof(float) * tab.size()));
        HIPCK(upload_sync(d_tab, tab.data(), sizeof(float) * tab.size()));
        auto launch = [&]() {
            if (kind == 9)
                hipLaunchKernelGGL(k_resident<1>, dim3(cnt), dim3(256), 0, s, p, h->d_res + h->res_first[1], d_tab);
            else
                launch_sparse(h, p, 0, d_tab, s);
        };
        launch();  // warm
        HIPCK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) launch();
        HIPCK(hipEventRecord(e1, s));
        HIPCK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)pool_free(d_tab);
        *ms_avg = ms / reps;
        double sn2 = 0;
        for (int t = 0; t < h->prob.num_targets; ++t)
            if (h->cat[t] == (kind == 8 ? CAT_SPARSE : 1)) sn2 += (double)h->meta[t].n * h->meta[t].n;
        if (alg_bytes) *alg_bytes = 28.0 * sn2 * hy->num_iters;
        if (alg_flops) *alg_flops = 6.0 * sn2 * (h->prob.D + 2.0 * h->prob.H) * hy->num_iters;
        HIPCK(hipGetLastError());
        return 0;
    }
    float ss, b2;
    adam_scalars(hy, 0, &ss, &b2);
    // the tables gnnx_run would walk: only the streaming remainder when resident kernels take part of the batch
    const bool hybrid = hy->use_resident && (h->n_res > 0 || h->n_sparse() > 0) && h->n_big > 0;
    if (int rc = hybrid ? ensure_mask_big(h) : ensure_mask_all(h)) return rc;
    const Tables tb = hybrid ? tables_big(h) : tables_all(h);
    double sum_n2 = h->sum_n2;
    if (hybrid) {
        sum_n2 = 0;
        for (int t = 0; t < h->prob.num_targets; ++t)
            if (h->cat[t] == 0) sum_n2 += (double)h->meta[t].n * h->meta[t].n;
    }
    auto once = [&]() {
        switch (kind) {
            case 0: launch_mask<true, true>(h, tb, p, 0, ss, b2, s); break;
            case 1: launch_conv<FWD1>(tb, p, 0, s); break;
            case 2: launch_conv<FWD2>(tb, p, 0, s); break;
            case 3:
                if (h->prob.graph_mode) hipLaunchKernelGGL(k_head, dim3(h->prob.num_targets), dim3(256), 0, s, p, 0);
                else hipLaunchKernelGGL(k_node_head, dim3(tb.n_conv), dim3(256), 0, s, p, tb.conv, 0);
                break;
            case 4: launch_conv<BWD1>(tb, p, 0, s); break;
            case 5: launch_conv<FWD3>(tb, p, 0, s); break;
            case 6: launch_conv<BWD3>(tb, p, 0, s); break;
            default: launch_conv<BWD2>(tb, p, 0, s); break;
        }
    };
    once();  // warm
    HIPCK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) once();
    HIPCK(hipEventRecord(e1, s));
    HIPCK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_avg = ms / reps;
    // algorithmic work of one launch: fused mask kernel = read+write M, m, v + read A = 28 B and
    // the K = D+2H product = 2 (D+2H) flop per mask entry; a contraction = one 4-B read of Abar and 2 d flop per entry
    const double kagg = h->prob.D + 2.0 * h->prob.H;
    const bool contraction = (kind == 1 || kind == 2 || kind == 4 || kind == 5 || kind == 7);
    if (alg_bytes) *alg_bytes = (kind == 0) ? 28.0 * sum_n2 : contraction ? 4.0 * sum_n2 : 0.0;
    if (alg_flops) {
        const double d = (kind == 1) ? h->prob.D : h->prob.H;
        *alg_flops = (kind == 0) ? 2.0 * sum_n2 * kagg : contraction ? 2.0 * sum_n2 * d : 0.0;
    }
    HIPCK(hipGetLastError());
    return 0;
}

// =============================================================================================================================
// The XL route (include/gnnx.h: gnnx_xl_*): node-mode targets of ANY size, CSR-native from the resident graph to the edge lists of the
// result - k_xl_rowdeg / k_xl_rowptr / k_xl_emit (gnnx_xl.hpp) build every target's sub-graph CSR, k_sparse_large<.., XL = true>
// (gnnx_sparse_large.hpp) runs all iterations with its state in the target's scratch block.  No dense n x n block exists anywhere.
// =============================================================================================================================
#include "gnnx_xl.hpp"

struct gnnx_xl_s {
    gnnx_problem prob{};
    std::vector<TargetMeta> meta;
    std::vector<XlBlock> blocks;
    std::vector<int32_t> ids;
    int64_t R = 0;                    // rows of the row arrays (sum of ld)
    int64_t E = -1, NNZ = -1;         // upper-triangle edges / directed entries of the batch (after gnnx_xl_count)
    std::vector<int64_t> edges;       // per target
    std::vector<long long> csr_off, eoff, scr_off;
    long long scr_floats = 0;
    bool weighted = false;
    bool built = false;
    // device tables (library-owned, recycled)
    void* block1 = nullptr;           // meta, blocks, ids, wts
    TargetMeta* d_meta = nullptr;
    XlBlock* d_blocks = nullptr;
    int32_t* d_ids = nullptr;
    float* d_wts = nullptr;
    void* block2 = nullptr;           // csr_off, eoff, scr_off, rp_off
    long long* d_csr_off = nullptr;
    long long* d_eoff = nullptr;
    long long* d_scr_off = nullptr;
    long long* d_rp_off = nullptr;
    float* d_adam = nullptr;
    bool adam_shared = false;
    gnnx_hyper adam_for{};
    int adam_first = -1;
    uint32_t* trace_gates = nullptr;
    long long* clk = nullptr;         // gnnx_xl_set_clocks: [T][4] device ticks of every target's workgroup (measurement hook)
    void* mt_block = nullptr;         // gnnx_xl_mt_edge_words: segment tables + segment start states (pooled)
    // carve-out of the caller's two workspaces (bytes)
    size_t r_X, r_yhat, r_deg, r_updeg, r_rowptr, r_uprow, r_totals, rows_bytes = 0;
    size_t e_col, e_row, e_w, e_scr, entries_bytes = 0;
};

static size_t xl_take(size_t& o, size_t bytes) {
    const size_t r = o;
    o = align_up(o + bytes, 256);
    return r;
}

extern "C" int gnnx_xl_create(const gnnx_problem* prob, const gnnx_model* model, gnnx_xl_handle* out) {
    if (!prob || !model || !out) return fail("null argument");
    if (prob->num_targets <= 0) return fail("num_targets must be positive");
    if (prob->graph_mode || prob->mask_relu || prob->bn) return fail("the XL route implements node mode, sigmoid masks, no --bn");
    if (prob->D < 1 || prob->D > FS || prob->H < 2 || prob->H > FS || prob->O < 1 || prob->O > FS) return fail("D, H, O must be in [1, 32] (H >= 2)");
    if (prob->C < 1 || prob->C > RES_CMAX) return fail("the XL route takes at most " + std::to_string(RES_CMAX) + " classes");
    auto* h = new gnnx_xl_s();
    h->prob = *prob;
    const int T = prob->num_targets;
    h->meta.resize(T);
    std::vector<long long> rp_off(T);
    for (int t = 0; t < T; ++t) {
        TargetMeta& m = h->meta[t];
        m.n = prob->n[t];
        if (m.n < 1) {
            delete h;
            return fail("empty sub-graph (n < 1) for target " + std::to_string(t));
        }
        m.ld = (m.n + TILE - 1) / TILE * TILE;
        m.t = prob->target_row[t];
        m.y_gt = prob->gt_label[t];
        if (m.t < 0 || m.t >= m.n || m.y_gt < 0 || m.y_gt >= prob->C) {
            delete h;
            return fail("target_row / gt_label out of range for target " + std::to_string(t));
        }
        m.offQ = 0;
        m.offR = h->R;
        rp_off[t] = h->R + t;            // rowptr / uprow: ld + 1 ints per target
        h->R += m.ld;
        for (int r0 = 0; r0 < m.ld; r0 += XL_ROWS_PER_BLOCK) h->blocks.push_back({t, r0});
        h->ids.push_back(t);
    }
    // largest targets first: their workgroups are the long poles of the one launch
    std::stable_sort(h->ids.begin(), h->ids.end(), [&](int a, int b) { return h->meta[a].n > h->meta[b].n; });
    h->csr_off.assign(2 * (size_t)T, 0);
    for (int t = 0; t < T; ++t) h->csr_off[2 * t] = rp_off[t];
    std::vector<float> w(WT_TOTAL, 0.0f);
    const int din[3] = {prob->D, prob->H, prob->H}, dout[3] = {prob->H, prob->H, prob->O};
    for (int l = 0; l < 3; ++l) {
        for (int k = 0; k < din[l]; ++k)
            for (int c = 0; c < dout[l]; ++c) w[WT_W + l * 1024 + k * 32 + c] = model->W[l][k * dout[l] + c];
        for (int c = 0; c < dout[l]; ++c) w[WT_B + l * 32 + c] = model->b[l] ? model->b[l][c] : 0.0f;
    }
    const int Ecat = prob->H + prob->H + prob->O;
    const int eo[3] = {0, prob->H, 2 * prob->H};
    for (int c = 0; c < prob->C; ++c) {
        for (int l = 0; l < 3; ++l)
            for (int j = 0; j < dout[l]; ++j) w[WT_WP + c * 96 + l * 32 + j] = model->Wp[c * Ecat + eo[l] + j];
        w[WT_BP + c] = model->bp[c];
    }
    // one block, one (synchronous) upload: meta | blocks | ids | wts | rp_off
    std::vector<char> host;
    auto add = [&](const void* src, size_t bytes) {
        const size_t off = align_up(host.size(), 256);
        host.resize(off + bytes);
        std::memcpy(host.data() + off, src, bytes);
        return off;
    };
    const size_t o_meta = add(h->meta.data(), sizeof(TargetMeta) * T), o_blk = add(h->blocks.data(), sizeof(XlBlock) * h->blocks.size()),
                 o_ids = add(h->ids.data(), sizeof(int32_t) * T), o_w = add(w.data(), sizeof(float) * w.size()),
                 o_rp = add(rp_off.data(), sizeof(long long) * T);
    if (pool_malloc(&h->block1, host.size()) != hipSuccess || upload_sync(h->block1, host.data(), host.size()) != hipSuccess) {
        if (h->block1) (void)pool_free(h->block1);
        delete h;
        return fail("gnnx_xl_create: device table upload failed");
    }
    char* b = static_cast<char*>(h->block1);
    h->d_meta = reinterpret_cast<TargetMeta*>(b + o_meta);
    h->d_blocks = reinterpret_cast<XlBlock*>(b + o_blk);
    h->d_ids = reinterpret_cast<int32_t*>(b + o_ids);
    h->d_wts = reinterpret_cast<float*>(b + o_w);
    h->d_rp_off = reinterpret_cast<long long*>(b + o_rp);
    size_t o = 0;
    h->r_X = xl_take(o, sizeof(float) * (size_t)h->R * FS);
    h->r_yhat = xl_take(o, sizeof(float) * (size_t)h->R);
    h->r_deg = xl_take(o, sizeof(int32_t) * (size_t)h->R);
    h->r_updeg = xl_take(o, sizeof(int32_t) * (size_t)h->R);
    h->r_rowptr = xl_take(o, sizeof(int32_t) * (size_t)(h->R + T));
    h->r_uprow = xl_take(o, sizeof(int32_t) * (size_t)(h->R + T));
    h->r_totals = xl_take(o, sizeof(int32_t) * 2 * (size_t)T);
    h->rows_bytes = o;
    *out = h;
    return 0;
}

extern "C" int gnnx_xl_destroy(gnnx_xl_handle h) {
    if (!h) return 0;
    if (h->block1) (void)pool_free(h->block1);
    if (h->block2) (void)pool_free(h->block2);
    if (h->mt_block) (void)pool_free(h->mt_block);
    if (h->d_adam && !h->adam_shared) (void)pool_free(h->d_adam);
    delete h;
    return 0;
}

extern "C" int64_t gnnx_xl_total_rows(gnnx_xl_handle h) { return h ? h->R : -1; }
extern "C" size_t gnnx_xl_rows_bytes(gnnx_xl_handle h) { return h ? h->rows_bytes : 0; }
extern "C" int64_t gnnx_xl_total_edges(gnnx_xl_handle h) { return h ? h->E : -1; }
extern "C" size_t gnnx_xl_entries_bytes(gnnx_xl_handle h) { return (h && h->E >= 0) ? h->entries_bytes : 0; }
extern "C" int gnnx_xl_get_layout(gnnx_xl_handle h, int32_t* ld, int64_t* offR, int64_t* eoff) {
    if (!h) return fail("null plan");
    const size_t T = h->meta.size();
    for (size_t t = 0; t < T; ++t) {
        if (ld) ld[t] = h->meta[t].ld;
        if (offR) offR[t] = h->meta[t].offR;
    }
    if (eoff) {
        if (h->E < 0) return fail("gnnx_xl_get_layout: the edge offsets exist after gnnx_xl_count");
        for (size_t t = 0; t <= T; ++t) eoff[t] = h->eoff[t];
    }
    return 0;
}

extern "C" int gnnx_xl_count(gnnx_xl_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const int32_t* nb,
                             const int64_t* nb_off, const float* feat, int32_t feat_stride, const float* pred_label, void* ws_rows,
                             size_t ws_rows_bytes, int64_t* edges_host, void* stream) {
    if (!h || !indptr || !nb || !nb_off || !feat || !ws_rows) return fail("null argument");      // (indices may be null: a graph without edges)
    if (ws_rows_bytes < h->rows_bytes) return fail("row workspace too small");
    if (feat_stride < h->prob.D) return fail("feat_stride smaller than D");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = h->prob.num_targets;
    char* w = static_cast<char*>(ws_rows);
    int32_t* deg = reinterpret_cast<int32_t*>(w + h->r_deg);
    int32_t* updeg = reinterpret_cast<int32_t*>(w + h->r_updeg);
    int32_t* totals = reinterpret_cast<int32_t*>(w + h->r_totals);
    hipLaunchKernelGGL(k_xl_rowdeg, dim3((unsigned)h->blocks.size()), dim3(XL_ROWS_PER_BLOCK), 0, s, h->d_meta, h->d_blocks, indptr, indices, weights, nb,
                       nb_off, feat, (int)feat_stride, (int)h->prob.D, pred_label, reinterpret_cast<float*>(w + h->r_X),
                       reinterpret_cast<float*>(w + h->r_yhat), deg, updeg);
    hipLaunchKernelGGL(k_xl_rowptr, dim3(T), dim3(XL_ROWPTR_THREADS), 0, s, h->d_meta, deg, updeg, h->d_rp_off, reinterpret_cast<int32_t*>(w + h->r_rowptr),
                       reinterpret_cast<int32_t*>(w + h->r_uprow), totals);
    HIPCK(hipGetLastError());
    std::vector<int32_t> tot(2 * (size_t)T);
    HIPCK(hipMemcpyAsync(tot.data(), totals, sizeof(int32_t) * tot.size(), hipMemcpyDeviceToHost, s));
    HIPCK(hipStreamSynchronize(s));
    h->edges.assign(T, 0);
    h->eoff.assign((size_t)T + 1, 0);
    h->scr_off.assign(T, 0);
    long long nnz_all = 0, scr = 0;
    for (int t = 0; t < T; ++t) {
        const long long nnz = tot[2 * t], up = tot[2 * t + 1];
        if (nnz != 2 * up) {
            h->E = -1;
            return fail("gnnx_xl_count: the sub-graph of target " + std::to_string(t) + " is not symmetric (" + std::to_string(nnz) + " directed entries, " +
                        std::to_string(up) + " above the diagonal)");
        }
        h->edges[t] = up;
        h->eoff[t + 1] = h->eoff[t] + up;
        h->csr_off[2 * t + 1] = nnz_all;
        nnz_all += nnz;
        h->scr_off[t] = scr;
        scr += xl_layout(h->meta[t].n, h->meta[t].ld, (int)nnz).total;
        if (edges_host) edges_host[t] = up;
    }
    h->E = h->eoff[T];
    h->NNZ = nnz_all;
    h->scr_floats = scr;
    h->weighted = weights != nullptr;
    size_t o = 0;
    h->e_col = xl_take(o, sizeof(int32_t) * (size_t)std::max<long long>(nnz_all, 1));
    h->e_row = xl_take(o, sizeof(int32_t) * (size_t)std::max<long long>(nnz_all, 1));
    h->e_w = h->weighted ? xl_take(o, sizeof(float) * (size_t)std::max<long long>(nnz_all, 1)) : 0;
    h->e_scr = xl_take(o, sizeof(float) * (size_t)scr);
    h->entries_bytes = o;
    // csr_off | eoff | scr_off: one block
    if (h->block2) (void)pool_free(h->block2);
    h->block2 = nullptr;
    std::vector<char> host;
    auto add = [&](const void* src, size_t bytes) {
        const size_t off = align_up(host.size(), 256);
        host.resize(off + bytes);
        std::memcpy(host.data() + off, src, bytes);
        return off;
    };
    const size_t o_c = add(h->csr_off.data(), sizeof(long long) * h->csr_off.size()), o_e = add(h->eoff.data(), sizeof(long long) * h->eoff.size()),
                 o_s = add(h->scr_off.data(), sizeof(long long) * h->scr_off.size());
    HIPCK(pool_malloc(&h->block2, host.size()));
    HIPCK(upload_sync(h->block2, host.data(), host.size()));
    char* b = static_cast<char*>(h->block2);
    h->d_csr_off = reinterpret_cast<long long*>(b + o_c);
    h->d_eoff = reinterpret_cast<long long*>(b + o_e);
    h->d_scr_off = reinterpret_cast<long long*>(b + o_s);
    h->built = false;
    return 0;
}

extern "C" int gnnx_xl_build(gnnx_xl_handle h, const int64_t* indptr, const int32_t* indices, const float* weights, const int32_t* nb,
                             const int64_t* nb_off, void* ws_rows, void* ws_entries, size_t ws_entries_bytes, int32_t* rc, void* stream) {
    if (!h || !indptr || !nb || !nb_off || !ws_rows || !ws_entries || !rc) return fail("null argument");
    if (h->E < 0) return fail("gnnx_xl_build: call gnnx_xl_count first");
    if (ws_entries_bytes < h->entries_bytes) return fail("entry workspace too small");
    if ((weights != nullptr) != h->weighted) return fail("gnnx_xl_build: pass the weights gnnx_xl_count saw");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(ws_rows);
    char* e = static_cast<char*>(ws_entries);
    hipLaunchKernelGGL(k_xl_emit, dim3((unsigned)h->blocks.size()), dim3(XL_ROWS_PER_BLOCK), 0, s, h->d_meta, h->d_blocks, indptr, indices, weights, nb,
                       nb_off, h->d_csr_off, reinterpret_cast<const int32_t*>(w + h->r_rowptr), reinterpret_cast<const int32_t*>(w + h->r_uprow), h->d_eoff,
                       reinterpret_cast<int32_t*>(e + h->e_col), reinterpret_cast<int32_t*>(e + h->e_row),
                       h->weighted ? reinterpret_cast<float*>(e + h->e_w) : nullptr, rc);
    HIPCK(hipGetLastError());
    h->built = true;
    return 0;
}

extern "C" int gnnx_xl_set_trace(gnnx_xl_handle h, uint32_t* gates) {
    if (!h) return fail("null argument");
    h->trace_gates = gates;
    return 0;
}

extern "C" int gnnx_xl_set_clocks(gnnx_xl_handle h, int64_t* ticks) {
    if (!h) return fail("null argument");
    h->clk = reinterpret_cast<long long*>(ticks);
    return 0;
}

extern "C" int gnnx_xl_run(gnnx_xl_handle h, const gnnx_hyper* hy, const gnnx_xl_state* st, float* abar_e, float* feat_mask, void* ws_rows,
                           void* ws_entries, void* stream) {
    if (!h || !hy || !st || !st->M_e || !abar_e || !feat_mask || !ws_rows || !ws_entries) return fail("null argument");
    if (!h->built) return fail("gnnx_xl_run: call gnnx_xl_count and gnnx_xl_build first");
    if (hy->num_iters < 1) return fail("num_iters must be >= 1");
    if (hy->opt < 0 || hy->opt > 3) return fail("opt must be 0 (adam), 1 (sgd), 2 (rmsprop) or 3 (adagrad)");
    if (hy->record_loss) return fail("the XL route has no loss logging (the decision trace: gnnx_xl_set_trace)");
    if (st->first_iter < 0) return fail("first_iter must be >= 0");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T = h->prob.num_targets;
    if (!h->d_adam || hy->lr_schedule || h->adam_first != st->first_iter || std::memcmp(&h->adam_for, hy, sizeof(gnnx_hyper)) != 0) {
        if (h->d_adam && !h->adam_shared) {
            HIPCK(hipStreamSynchronize(s));
            (void)pool_free(h->d_adam);
        }
        h->d_adam = nullptr;
        h->adam_shared = false;
        std::vector<float> tab(2 * (size_t)hy->num_iters);
        for (int it = 0; it < hy->num_iters; ++it) adam_scalars(hy, st->first_iter + it, &tab[2 * it], &tab[2 * it + 1], it);
        if (!hy->lr_schedule) h->d_adam = shared_adam_table(hy, st->first_iter, tab);
        if (h->d_adam) {
            h->adam_shared = true;
        } else {
            HIPCK(pool_malloc(&h->d_adam, sizeof(float) * tab.size()));
            HIPCK(upload_sync(h->d_adam, tab.data(), sizeof(float) * tab.size()));
        }
        h->adam_for = *hy;
        h->adam_first = st->first_iter;
    }
    char* w = static_cast<char*>(ws_rows);
    char* e = static_cast<char*>(ws_entries);
    Params p{};
    p.meta = h->d_meta;
    p.X = reinterpret_cast<const float*>(w + h->r_X);
    p.yhat = reinterpret_cast<const float*>(w + h->r_yhat);
    p.f[0] = p.f[1] = feat_mask;
    p.fs_in = st->feat;
    p.fs_out = st->feat_out;
    p.wts = h->d_wts;
    p.D = h->prob.D;
    p.H = h->prob.H;
    p.O = h->prob.O;
    p.C = h->prob.C;
    p.num_iters = hy->num_iters;
    p.opt = hy->opt;
    p.edge_only = 1;
    p.lr = (float)hy->lr;
    p.beta2 = (float)(hy->opt == 2 ? hy->alpha : hy->beta2);
    p.omb1 = (float)(1.0 - hy->beta1);
    p.omb2 = (float)(1.0 - (hy->opt == 2 ? hy->alpha : hy->beta2));
    p.eps = (float)hy->eps;
    p.c_size = hy->c_size;
    p.c_feat_size = hy->c_feat_size;
    p.c_ent = hy->c_ent;
    p.c_lap = hy->c_lap;
    p.trace_gates = h->trace_gates;
    p.trace_rows = h->R;
    XlIo io{};
    io.w = h->weighted ? reinterpret_cast<const float*>(e + h->e_w) : nullptr;
    io.eoff = h->d_eoff;
    io.M_e = st->M_e;
    io.m_in_e = st->m_e;
    io.v_in_e = st->v_e;
    io.m_out_e = st->m_out_e;
    io.v_out_e = st->v_out_e;
    io.abar_e = abar_e;
    io.scr = reinterpret_cast<float*>(e + h->e_scr);
    io.scr_off = h->d_scr_off;
    io.clk = h->clk;
    const int32_t* rowptr = reinterpret_cast<const int32_t*>(w + h->r_rowptr);
    const void* col = e + h->e_col;
    const void* row = e + h->e_row;
    const dim3 grid(T), block(SPL_THREADS);
    const bool exact = h->prob.D == 10 && h->prob.H == 20 && h->prob.O == 20;
    if (h->trace_gates) {
        if (!exact) return fail("gnnx_xl_set_trace: the decision trace needs the reference's widths (D = 10, H = O = 20)");
        HIPCK(hipMemsetAsync(h->trace_gates, 0, sizeof(uint32_t) * 2 * (size_t)h->R * hy->num_iters, s));
        hipLaunchKernelGGL((k_sparse_large<5, 10, true, true, true>), grid, block, 0, s, p, h->d_ids, h->d_adam, rowptr, col, row, h->d_csr_off,
                           (const int32_t*)nullptr, io);
    } else if (exact) {
        hipLaunchKernelGGL((k_sparse_large<5, 10, false, true, true>), grid, block, 0, s, p, h->d_ids, h->d_adam, rowptr, col, row, h->d_csr_off,
                           (const int32_t*)nullptr, io);
    } else {
        hipLaunchKernelGGL((k_sparse_large<16, 16, false, true, true>), grid, block, 0, s, p, h->d_ids, h->d_adam, rowptr, col, row, h->d_csr_off,
                           (const int32_t*)nullptr, io);
    }
    HIPCK(hipGetLastError());
    return 0;
}

// the jump polynomial of the segmented engine walk (utils/mt_jump.py): process-wide, one device copy per device; jump = 0: serial walks
namespace {
struct MtJump { std::vector<uint32_t> poly; long long jump = 0; int levels = 1; uint32_t* d_poly[MAX_DEVICES] = {}; };
MtJump g_mtj;
std::mutex g_mtj_mu;
}  // namespace
// polys [levels][624]: the polynomials of the strides 4^l segments (l = 0 .. levels - 1; levels <= 3): x^(4^l J) mod phi
extern "C" int gnnx_set_mt_jump_polys(const uint32_t* polys, int64_t jump_draws, int32_t levels) {
    std::lock_guard<std::mutex> lk(g_mtj_mu);
    if (!polys || jump_draws <= 0) {
        g_mtj.jump = 0;
        return 0;
    }
    if (jump_draws % MT_N) return fail("gnnx_set_mt_jump_polys: the stride must be a whole number of 624-draw blocks");
    if (levels < 1 || levels > MTJ_MAX_LEVELS) return fail("gnnx_set_mt_jump_polys: one to three levels");
    g_mtj.poly.assign(polys, polys + (size_t)levels * MT_N);
    g_mtj.jump = jump_draws;
    g_mtj.levels = levels;
    for (auto& d : g_mtj.d_poly) d = nullptr;      // (device copies of an older polynomial are dropped: a few KB, once per process)
    return 0;
}
extern "C" int gnnx_set_mt_jump_poly(const uint32_t* poly, int64_t jump_draws) { return gnnx_set_mt_jump_polys(poly, jump_draws, 1); }

extern "C" int gnnx_xl_mt_edge_words(gnnx_xl_handle h, const int64_t* seeds, void* ws_rows, void* ws_entries, uint32_t* words, void* stream) {
    if (!h || !seeds || !ws_rows || !ws_entries || !words) return fail("null argument");
    if (!h->built) return fail("gnnx_xl_mt_edge_words: call gnnx_xl_count and gnnx_xl_build first");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* w = static_cast<char*>(ws_rows);
    char* e = static_cast<char*>(ws_entries);
    const int T = h->prob.num_targets;
    long long jump = 0;
    int levels = 1;
    const uint32_t* d_poly = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mtj_mu);
        jump = g_mtj.jump;
        levels = g_mtj.levels;
        if (jump > 0) {
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= MAX_DEVICES) dev = 0;
            if (!g_mtj.d_poly[dev]) {
                uint32_t* d = nullptr;
                HIPCK(hipMalloc(&d, sizeof(uint32_t) * g_mtj.poly.size()));
                HIPCK(upload_sync(d, g_mtj.poly.data(), sizeof(uint32_t) * g_mtj.poly.size()));
                g_mtj.d_poly[dev] = d;
            }
            d_poly = g_mtj.d_poly[dev];
        }
    }
    // segments of every target (host: from the sizes alone), their start states in a pooled scratch block
    std::vector<long long> seg_off((size_t)T + 1, 0);
    std::vector<MtSeg> segs;
    for (int t = 0; t < T; ++t) {
        const int K = mt_segments(h->meta[t].n, jump);
        seg_off[t + 1] = seg_off[t] + K;
    }
    segs.reserve((size_t)seg_off[T]);
    // longest segments first are all alike (`jump` draws); the LAST segment of a target is the long one (up to 2 jump): launch those first
    for (int t = 0; t < T; ++t)
        if (seg_off[t + 1] > seg_off[t]) segs.push_back({t, (int32_t)(seg_off[t + 1] - seg_off[t] - 1)});
    for (int t = 0; t < T; ++t)
        for (int k = 0; k + 1 < seg_off[t + 1] - seg_off[t]; ++k) segs.push_back({t, k});
    if (segs.empty()) return 0;
    if (h->mt_block) {
        HIPCK(hipStreamSynchronize(s));      // an earlier walk of this plan may still read the block
        (void)pool_free(h->mt_block);
        h->mt_block = nullptr;
    }
    // the chains of the segment starts, level by level (k_mt_segment_starts): level l jumps 4^l segments at a time; the top level's chain of a
    // target seeds it and runs from segment 0, every lower level starts one chain of up to three jumps at every multiple of 4^(l + 1)
    std::vector<MtJumpItem> items;
    size_t lvl_first[MTJ_MAX_LEVELS + 1] = {};
    if (jump <= 0) levels = 1;
    for (int l = levels - 1; l >= 0; --l) {
        lvl_first[levels - 1 - l] = items.size();
        int step = 1;
        for (int k = 0; k < l; ++k) step *= MTJ_RADIX;
        const bool top = l == levels - 1;
        for (int t = 0; t < T; ++t) {
            const int K = (int)(seg_off[t + 1] - seg_off[t]);
            if (K == 0) continue;
            if (top) {
                items.push_back({t, 0, step, (K - 1) / step, 1, 0});
            } else {
                for (int s0 = 0; s0 < K; s0 += step * MTJ_RADIX) {
                    const int cnt = std::min(MTJ_RADIX - 1, (K - 1 - s0) / step);
                    if (cnt > 0) items.push_back({t, s0, step, cnt, 0, 0});
                }
            }
        }
    }
    lvl_first[levels] = items.size();
    const size_t o_off = 0, o_seg = align_up(sizeof(long long) * seg_off.size(), 256), o_items = align_up(o_seg + sizeof(MtSeg) * segs.size(), 256),
                 o_state = align_up(o_items + sizeof(MtJumpItem) * items.size(), 256);
    const size_t bytes = o_state + sizeof(uint32_t) * MT_N * (size_t)seg_off[T];
    HIPCK(pool_malloc(&h->mt_block, bytes));
    std::vector<char> host(o_state);
    std::memcpy(host.data() + o_off, seg_off.data(), sizeof(long long) * seg_off.size());
    std::memcpy(host.data() + o_seg, segs.data(), sizeof(MtSeg) * segs.size());
    std::memcpy(host.data() + o_items, items.data(), sizeof(MtJumpItem) * items.size());
    HIPCK(upload_sync(h->mt_block, host.data(), host.size()));
    char* mb = static_cast<char*>(h->mt_block);
    const long long* d_seg_off = reinterpret_cast<const long long*>(mb + o_off);
    const MtSeg* d_segs = reinterpret_cast<const MtSeg*>(mb + o_seg);
    uint32_t* d_state = reinterpret_cast<uint32_t*>(mb + o_state);
    const MtJumpItem* d_items = reinterpret_cast<const MtJumpItem*>(mb + o_items);
    for (int k = 0; k < levels; ++k) {      // top level first; a level reads the states the one above it wrote (same stream)
        const size_t cnt = lvl_first[k + 1] - lvl_first[k];
        if (cnt)
            hipLaunchKernelGGL(k_mt_segment_starts, dim3((unsigned)cnt), dim3(MTX_THREADS), 0, s, h->d_meta, seeds, d_seg_off,
                               d_poly ? d_poly + (size_t)(levels - 1 - k) * MT_N : nullptr, jump, d_state, d_items + lvl_first[k]);
    }
    hipLaunchKernelGGL(k_mt_edge_words_seg, dim3((unsigned)segs.size()), dim3(MTX_THREADS), 0, s, h->d_meta, d_segs, d_seg_off, jump, (const uint32_t*)d_state,
                       h->d_csr_off, reinterpret_cast<const int32_t*>(w + h->r_rowptr), reinterpret_cast<const int32_t*>(w + h->r_uprow),
                       reinterpret_cast<const int32_t*>(e + h->e_col), reinterpret_cast<const int32_t*>(e + h->e_row), h->d_eoff, words);
    HIPCK(hipGetLastError());
    return 0;
}
