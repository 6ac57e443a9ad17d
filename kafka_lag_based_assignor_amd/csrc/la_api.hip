// la_api.hip -- the C ABI of include/lagassign.h over the HIP kernels.
//
// There is no CPU fallback here by design: if the device or a kernel is unavailable the
// call fails with a negative code and the caller decides what to do.
#include "../../include/lagassign.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "la_kernels.h"
#include "la_group_small.h"

#define LA_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_create_error;
// Worker threads of a host-buffer call report into their own slot (the first failing worker's text becomes the
// context's last error once they have joined); everything else writes the context's string directly.
thread_local std::string* t_err_sink = nullptr;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// Every entry point that switches devices leaves the calling thread's current HIP device as it found it (a caller that
// drives its own kernels next to the library must not find its device changed by a call on a multi-device context).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace

// Everything ONE in-flight batch needs besides the bulk arrays: a stream, the device status word + deferred-tile
// counters, and the per-call scratch of the three kernel paths.  A shard owns several lanes so that chunks of one
// host-buffer call can be in flight side by side (H2D of one chunk under the kernels / D2H of another); the device
// entry points use lane 0 with the caller's stream.
struct Lane {
    hipStream_t stream = nullptr;
    uint32_t* d_status = nullptr;        // 256 B: word 0 = status bits, words 16..19 = deferred-tile counters
    uint32_t* h_status = nullptr;        // pinned word the status is copied to (one stream sync per call)
    uint32_t* status_word = nullptr;     // small-batch call: the status word lives inside the call's one device staging
                                         // buffer (uploaded as zero with the inputs, downloaded with the results)
    uint32_t* status() const { return status_word ? status_word : d_status; }
    // tile path: list of tiles the packed kernel leaves to the wide kernel, and which of the two
    // counters (d_status + 16 / + 17 words) the next launch uses
    DevBuf defer;
    unsigned launches = 0;
    // block path: topic lists per size class.  Built on the host into a small ring of pinned slots (a slot
    // is reused only after the copy that read it has completed) and copied to block_list.
    DevBuf block_list;
    struct Stage {
        int32_t* p = nullptr;
        size_t cap = 0;          // in int32 entries
        hipEvent_t done = nullptr;
    } stage[4];
    unsigned stage_next = 0;
    la::LargeScratch large;              // scratch of the large-topic path and of la_group_by_member
    // a zero-copy small call asks the dispatcher to hang its tail (lists + completion word, la::TileTail) on the batch's tile
    // launch; honoured only when the batch IS one single-launch tile kernel -- tail_done says whether it was
    la::TileTail tail{};
    bool tail_wanted = false, tail_done = false;
    std::vector<uint8_t> topic_class;    // host scratch of the dispatcher: path / class of every topic
    std::vector<int64_t> host_offsets;   // offsets fetched from the device when the caller gave no host copy
};

struct HostBuf {                         // pinned, grow-only
    void* p = nullptr;
    size_t cap = 0;
};

// One shard = one device (or, for tests, one of several logical shards mapped to the same device): the bulk
// device arrays of the host-buffer entry points, and its lanes.
struct Shard {
    int device = 0;
    DevBuf part_off, pid, begin, end, committed, cons_off, cons_rank, out_pid, out_rank, out_total;
    DevBuf none_idx, none_val;           // la_assign_batch_sparse: the shard's slice of the (position, begin) list
    std::vector<Lane> lanes;
    hipEvent_t ready = nullptr;          // the shard's offsets are on the device (lane 0's stream)
    std::vector<int64_t> local_part_off, local_cons_off;   // offsets rebased to the shard's first topic (shards > 0)
    HostBuf g_off, g_topic, g_part;      // multi-shard la_group_last_by_member: this shard's CSR before the merge
    // results of the last host-buffer assign call, still on the device (la_group_last_by_member)
    int32_t last_t0 = 0, last_topics = 0;
    int64_t last_p0 = 0, last_n = 0;
    const int64_t* last_part_off = nullptr;
    const int32_t *last_out_pid = nullptr, *last_out_rank = nullptr;
    // small batches (what a real group leader sends): inputs and results of a call travel as ONE H2D and ONE D2H copy
    // through these two staging buffers (device, pinned host) instead of eight copies of caller arrays
    DevBuf small_d, small_g;
    HostBuf small_h, small_gh;
    HostBuf zc_h;                        // zero-copy small calls: coherent, device-mapped staging the kernels read and write in place
    // pinned caller arrays (la_host_alloc): one host thread, three streams -- every H2D of the call in order on copy_in,
    // the kernels on lane 0's stream, every D2H on copy_out, chained per chunk by events (run_shard_async)
    static constexpr int kCopyIn = 2;    // input streams, alternating per chunk: a copy costs ~20 us of engine latency before its
                                         // first byte moves; one chunk's gaps hide under the other stream's bytes.  (One stream
                                         // per ARRAY, five copies in flight at once, was measured far worse: 19-26 ms against 15.)
    hipStream_t copy_in[kCopyIn] = {}, copy_out = nullptr;
    std::vector<hipEvent_t> chunk_ev;    // two per chunk: inputs landed, results ready
};

// one polite spin step of a host thread: the x86 pause where there is one, the architecture's yield elsewhere (the host side
// of the library builds on aarch64 / ppc64 hosts too)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    __asm__ __volatile__("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

// The host threads of the host-buffer calls, PARKED between calls.  The copies of a pageable caller array block the thread that
// issues them, which is why a shard's lanes are threads; spawning and joining them on every call cost ~50 us per thread
// (VERDICT r4 weak #5).  The pool belongs to the context (created with its first multi-lane call, joined by la_destroy); a call
// hands it a job, runs index 0 itself and waits for the others.  The context is single-threaded for its caller, so one job at
// a time.  A worker that has just finished spins briefly before it parks on the condition variable: the next job of the same
// call (grouping after the assignment, the second phase of a multi-shard merge) then starts without a futex round trip.
struct WorkerPool {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    std::function<void(int)> job;
    int n_jobs = 0;
    std::atomic<int> next{0};
    int running = 0;                      // workers inside the current job
    std::atomic<uint64_t> gen{0};
    bool stop = false;

    void worker() {
        uint64_t seen = 0;
        for (;;) {
            // a short spin first: jobs of one call follow each other within microseconds
            for (int spin = 0; spin < 2000 && gen.load(std::memory_order_acquire) == seen; ++spin) cpu_relax();
            const std::function<void(int)>* my_job = nullptr;
            int my_n = 0;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return stop || gen.load(std::memory_order_relaxed) != seen; });
                if (stop) return;
                seen = gen.load(std::memory_order_relaxed);
                if (n_jobs == 0) continue;           // a late riser: that job is over (run() zeroes n_jobs under this lock)
                ++running;                           // registered under the lock: run() does not return before this worker is done
                my_job = &job;
                my_n = n_jobs;
            }
            for (;;) {
                const int i = next.fetch_add(1, std::memory_order_relaxed);
                if (i >= my_n) break;
                (*my_job)(i);
            }
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) cv_done.notify_all();
            }
        }
    }

    // fn(i) for i in [0, n): index 0 on the calling thread, the rest on the pool (grown to n - 1 threads; where no thread is to
    // be had the caller runs what is left itself).
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        while ((int)threads.size() < n - 1) {
            try { threads.emplace_back([this] { worker(); }); } catch (...) { break; }
        }
        {
            std::lock_guard<std::mutex> lk(m);
            job = fn;
            n_jobs = n;
            next.store(1, std::memory_order_relaxed);
            gen.fetch_add(1, std::memory_order_release);
        }
        cv_work.notify_all();
        fn(0);
        for (;;) {                                   // whatever the pool has not taken (no threads, or fewer than n - 1)
            const int i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) break;
            fn(i);
        }
        // every index has been TAKEN; wait until the workers that registered for this job are done.  A worker reads the job only
        // under the lock and only while n_jobs != 0: one that wakes up after this point finds nothing and parks again.
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return running == 0; });
        n_jobs = 0;
    }

    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_work.notify_all();
        for (std::thread& t : threads) if (t.joinable()) t.join();
    }
};

struct la_ctx {
    WorkerPool pool;
    std::vector<Shard> shards;
    std::string err;
    bool split_always = false;           // LA_CREATE_SPLIT_ALWAYS: shard and chunk even tiny batches (tests)
    int64_t chunk_partitions = 0;        // LA_CHUNK_PARTITIONS override (0: automatic)
    bool last_valid = false;
    int last_pipeline = 0;               // how the last host-buffer call moved its data: 0 = one copy each way (small batch),
                                         // 1 = lanes (a host thread per stream; pageable arrays), 2 = three streams, no threads (pinned)
    std::vector<void*> comms;            // la_allgather_results: one ncclComm_t per shard, created on first use
    size_t zero_copy_bytes = 0;          // calls whose staging layout is at most this large run zero-copy (assign_small_zc)
    size_t small_bytes = 0;              // ... and up to this large as ONE staging buffer at all (zero-copy or one copy)
    int last_shards = 0;                 // shards the last call used
    int32_t last_bounds[65] = {};        // their topic ranges
    la_call_hints hints{};               // la_hint_next_call: what the caller knows about its next host-buffer assign call
    bool hints_set = false;              // (one-shot: assign_host takes them and clears the flag)
    int64_t last_launches = 0;           // kernel launches of the last call (la_last_launches)
};

namespace {

// Kernel launches between construction and destruction, left in the context for la_last_launches (the outermost span of a
// call wins: the grouped calls run an assign call inside).
struct LaunchSpan {
    la_ctx* ctx;
    uint64_t at;
    explicit LaunchSpan(la_ctx* c) : ctx(c), at(la::g_kernel_launches.load(std::memory_order_relaxed)) {}
    ~LaunchSpan() { if (ctx) ctx->last_launches = (int64_t)(la::g_kernel_launches.load(std::memory_order_relaxed) - at); }
    LaunchSpan(const LaunchSpan&) = delete;
    LaunchSpan& operator=(const LaunchSpan&) = delete;
};

int fail(la_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (t_err_sink) *t_err_sink = buf;
    else if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

#define LA_HIP(ctx, expr)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(ctx, e_ == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "%s: %s", #expr, \
                        hipGetErrorString(e_));                                              \
    } while (0)

int reserve(la_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return LA_OK;
    if (b.p) { LA_HIP(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 4 + 256;
    LA_HIP(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return LA_OK;
}

int reserve_host(la_ctx* ctx, HostBuf& b, size_t bytes);

void release(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

struct Shape {
    int64_t n = 0, k = 0, max_p = 0, max_c = 0;
};

// Validates offsets (and, if given, the ascending-rank contract) on the host.
int scan_shape(la_ctx* ctx, int32_t T, const int64_t* part_off, const int64_t* cons_off,
               const int32_t* cons_rank, Shape* s) {
    if (part_off[0] != 0 || cons_off[0] != 0) return fail(ctx, LA_EINVAL, "part_off[0] and cons_off[0] must be 0");
    for (int32_t t = 0; t < T; ++t) {
        const int64_t p = part_off[t + 1] - part_off[t], c = cons_off[t + 1] - cons_off[t];
        if (p < 0 || c < 0) return fail(ctx, LA_EINVAL, "offsets of topic %d decrease", t);
        if (p > s->max_p) s->max_p = p;
        if (c > s->max_c) s->max_c = c;
        if (cons_rank)
            for (int64_t k = cons_off[t] + 1; k < cons_off[t + 1]; ++k)
                if (cons_rank[k - 1] >= cons_rank[k])
                    return fail(ctx, LA_EINVAL,
                                "cons_rank of topic %d is not strictly ascending at %lld", t, (long long)k);
    }
    s->n = part_off[T];
    s->k = cons_off[T];
    return LA_OK;
}

// ---- dispatcher: which path every topic of a batch takes --------------------------------------------------
//   * tile-sized topics (<= 1 024 partitions, <= 64 consumers): one wave-tile launch over the whole batch that
//     skips the others -- or, when the shapes are ragged enough to pay for it, one launch per shape class over a
//     topic list, so that a few wide topics do not make every topic pay for the widest tile;
//   * up to 8 192 partitions x 2 048 consumers: the block path, one workgroup per topic, one launch per size
//     class over a topic list;
//   * beyond: the large path, topic by topic.
constexpr int kTileClasses = 3;
constexpr int64_t kTileClsP[kTileClasses] = {64, 256, la::kTileMaxPartitions};
constexpr int64_t kTileClsC[kTileClasses] = {8, 32, la::kTileMaxConsumers};
constexpr uint8_t kLargeCode = kTileClasses + la::kBlockClasses;      // codes: tile classes, block classes, large

struct BatchPlan {
    const uint8_t* code = nullptr;                                    // per topic
    struct { int64_t n = 0, mp = 0, mc = 0; } tile[kTileClasses];     // after merging: what each launch holds
    int merged[kTileClasses] = {0, 1, 2};                             // tile class -> the class it is launched with
    bool classed = false;                                             // tile topics go by shape class (lists)
    int64_t n_tile = 0, tile_mp = 0, tile_mc = 0;
    int64_t n_block[la::kBlockClasses] = {};
    int64_t n_block_all = 0, n_large = 0;
    // offsets of the lists in the staged array: block classes first, then tile classes
    int64_t block_at[la::kBlockClasses] = {}, tile_at[kTileClasses] = {};
    int64_t n_lists = 0;
};

// One pass over the host offsets: a class code per topic, counts and maxima; then the tile plan.
int plan_batch(la_ctx* ctx, Lane& ln, const la_device_batch* b, int tile_mode, bool use_block, BatchPlan* plan) {
    const int64_t T = b->n_topics;
    ln.topic_class.resize((size_t)T);
    uint8_t* code = ln.topic_class.data();
    int64_t cnt[kLargeCode + 1] = {};
    int64_t mp[kTileClasses] = {}, mc[kTileClasses] = {};
    bool decreasing = false;
    const int64_t* po = b->h_part_off;
    const int64_t* co = b->h_cons_off;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t p = po[t + 1] - po[t], c = co[t + 1] - co[t];
        decreasing |= (p < 0) | (c < 0);
        uint8_t k;
        if (p <= kTileClsP[0] && c <= kTileClsC[0]) k = 0;
        else if (p <= kTileClsP[1] && c <= kTileClsC[1]) k = 1;
        else if (p <= kTileClsP[2] && c <= kTileClsC[2]) k = 2;
        else if (use_block && la::block_fits(p, c)) k = (uint8_t)(kTileClasses + la::block_class(p, c));
        else k = kLargeCode;
        code[t] = k;
        ++cnt[k];
        if (k < kTileClasses) {
            if (p > mp[k]) mp[k] = p;
            if (c > mc[k]) mc[k] = c;
        }
    }
    if (decreasing) return fail(ctx, LA_EINVAL, "part_off / cons_off decrease");
    plan->code = code;
    for (int k = 0; k < kTileClasses; ++k) {
        plan->tile[k].n = cnt[k]; plan->tile[k].mp = mp[k]; plan->tile[k].mc = mc[k];
        plan->n_tile += cnt[k];
        if (mp[k] > plan->tile_mp) plan->tile_mp = mp[k];
        if (mc[k] > plan->tile_mc) plan->tile_mc = mc[k];
    }
    for (int k = 0; k < la::kBlockClasses; ++k) {
        plan->n_block[k] = cnt[kTileClasses + k];
        plan->n_block_all += cnt[kTileClasses + k];
    }
    plan->n_large = cnt[kLargeCode];

    // tile plan: the sort slots a launch spends = topics x lanes x records per lane of its tile shape
    auto work = [](int64_t n, int64_t p, int64_t c) {
        int L = 0, E = 0;
        la::wave_tile_pick(p, c, &L, &E);
        return (double)n * L * E;
    };
    const bool force = (b->flags & LA_FLAG_SHAPE_CLASSES) != 0;
    if (tile_mode == 0 && (plan->n_tile >= 4096 || force)) {
        for (int k = 0; k + 1 < kTileClasses; ++k) {                    // a launch is not worth a handful of topics
            auto& lo = plan->tile[k];
            auto& hi = plan->tile[k + 1];
            if (lo.n > 0 && lo.n < 1024 && !force) {
                hi.n += lo.n;
                if (lo.mp > hi.mp) hi.mp = lo.mp;
                if (lo.mc > hi.mc) hi.mc = lo.mc;
                lo.n = 0;
                for (int q = 0; q <= k; ++q) if (plan->merged[q] == k) plan->merged[q] = k + 1;
            }
        }
        double split = 0;
        int launches = 0;
        for (const auto& k : plan->tile) if (k.n > 0) { split += work(k.n, k.mp, k.mc); ++launches; }
        // worth it when the sort slots saved (~12 ps each on the device) outweigh the second host pass over
        // the topics (~2 ns each) and the extra launches
        plan->classed = launches >= 2 &&
                        (force || work(plan->n_tile, plan->tile_mp, plan->tile_mc) - split > 170.0 * (double)T + 8e6);
    }
    for (int k = 1; k < la::kBlockClasses; ++k) plan->block_at[k] = plan->block_at[k - 1] + plan->n_block[k - 1];
    plan->tile_at[0] = plan->n_block_all;
    for (int k = 1; k < kTileClasses; ++k) plan->tile_at[k] = plan->tile_at[k - 1] + plan->tile[k - 1].n;
    plan->n_lists = plan->n_block_all + (plan->classed ? plan->n_tile : 0);
    return LA_OK;
}

// Second host pass: the topic lists, built in a pinned slot of the context's ring and copied to the device.
int stage_topic_lists(la_ctx* ctx, Lane& ln, const BatchPlan& plan, int64_t T, hipStream_t stream, const int32_t** d_lists) {
    *d_lists = nullptr;
    if (plan.n_lists == 0 || (!plan.classed && plan.n_block_all <= 8)) return LA_OK;   // few block topics go inline
    Lane::Stage& sg = ln.stage[ln.stage_next++ & 3u];
    if (sg.done) LA_HIP(ctx, hipEventSynchronize(sg.done));          // the copy that last read this slot
    else LA_HIP(ctx, hipEventCreateWithFlags(&sg.done, hipEventDisableTiming));
    if (sg.cap < (size_t)plan.n_lists) {
        if (sg.p) { LA_HIP(ctx, hipHostFree(sg.p)); sg.p = nullptr; sg.cap = 0; }
        const size_t want = (size_t)plan.n_lists + (size_t)plan.n_lists / 2 + 64;
        LA_HIP(ctx, hipHostMalloc((void**)&sg.p, want * sizeof(int32_t), hipHostMallocDefault));
        sg.cap = want;
    }
    int64_t block_fill[la::kBlockClasses], tile_fill[kTileClasses];
    for (int k = 0; k < la::kBlockClasses; ++k) block_fill[k] = plan.block_at[k];
    for (int k = 0; k < kTileClasses; ++k) tile_fill[k] = plan.tile_at[k];
    for (int64_t t = 0; t < T; ++t) {
        const uint8_t k = plan.code[t];
        if (k < kTileClasses) {
            if (plan.classed) sg.p[tile_fill[plan.merged[k]]++] = (int32_t)t;
        } else if (k < kLargeCode) {
            sg.p[block_fill[k - kTileClasses]++] = (int32_t)t;
        }
    }
    // calls of one context are stream-ordered (lagassign.h), so the device copy of the lists is free again by
    // the time this copy runs
    if (int rc = reserve(ctx, ln.block_list, (size_t)plan.n_lists * sizeof(int32_t))) return rc;
    LA_HIP(ctx, hipMemcpyAsync(ln.block_list.p, sg.p, (size_t)plan.n_lists * sizeof(int32_t), hipMemcpyHostToDevice,
                               stream));
    LA_HIP(ctx, hipEventRecord(sg.done, stream));
    *d_lists = (const int32_t*)ln.block_list.p;
    return LA_OK;
}

int launch_block_topics(la_ctx* ctx, Lane& ln, const la_device_batch* b, const BatchPlan& plan, const int32_t* d_lists,
                        hipStream_t stream) {
    la::BlockArgs g{};
    g.part_off = b->d_part_off;
    g.cons_off = b->d_cons_off;
    g.pid = b->d_partition_id;
    g.begin = b->d_begin_off;
    g.end = b->d_end_off;
    g.committed = b->d_committed_off;
    g.lag = b->d_lag;
    g.cons_rank = b->d_cons_rank;
    g.out_pid = b->d_out_partition;
    g.out_rank = b->d_out_member_rank;
    g.out_total = b->d_out_total_lag;
    g.status = ln.status();
    g.reset_latest = (b->reset_mode == LA_RESET_LATEST) ? 1 : 0;
    if (!d_lists) {
        // a handful of block topics: their indices travel in the kernel arguments (no copy, no event)
        for (int cls = 0; cls < la::kBlockClasses; ++cls) {
            if (plan.n_block[cls] == 0) continue;
            int n = 0;
            for (int64_t t = 0; t < b->n_topics && n < (int)plan.n_block[cls]; ++t)
                if (plan.code[t] == kTileClasses + cls) g.inline_list[n++] = (int32_t)t;
            g.list = nullptr;
            g.n_list = n;
            LA_HIP(ctx, la::block_launch(g, cls, stream));
        }
        return LA_OK;
    }
    for (int cls = 0; cls < la::kBlockClasses; ++cls) {
        g.list = d_lists + plan.block_at[cls];
        g.n_list = (int32_t)plan.n_block[cls];
        LA_HIP(ctx, la::block_launch(g, cls, stream));
    }
    return LA_OK;
}

int launch_large_topics(la_ctx* ctx, Lane& ln, const la_device_batch* b, const BatchPlan& plan, bool argmin, hipStream_t stream) {
    // The per-topic loop of assign(Map,Map) (Main.java:177-184) is independent across topics: all large topics of the batch
    // run SIDE BY SIDE (la::large_topics_launch: every phase one launch over all of them, one greedy workgroup per topic),
    // in groups bounded by scratch memory.  Topics with more consumers than the one-workgroup greedy holds take the
    // bins-in-HBM form, one after another; the literal-argmin test hook stays serial.
    std::vector<la::LargeArgs> group;
    int64_t group_n = 0;
    auto flush = [&]() -> int {
        if (group.empty()) return LA_OK;
        const hipError_t e = la::large_topics_launch(ln.large, group.data(), (int)group.size(), stream);
        group.clear();
        group_n = 0;
        if (e != hipSuccess)
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "large topics: %s", hipGetErrorString(e));
        return LA_OK;
    };
    constexpr int64_t kGroupPartitions = (int64_t)1 << 30;          // ~26 GB of sort buffers
    constexpr size_t kGroupTopics = 8192;
    for (int64_t t = 0; t < b->n_topics; ++t) {
        if (plan.code[t] != kLargeCode) continue;
        const int64_t p = b->h_part_off[t + 1] - b->h_part_off[t], c = b->h_cons_off[t + 1] - b->h_cons_off[t];
        if (p > 0x7FFFFFFF || c > 0x7FFFFFFF)
            return fail(ctx, LA_ESHAPE, "topic %lld has %lld partitions and %lld consumers; at most 2^31-1 of each are supported",
                        (long long)t, (long long)p, (long long)c);
        la::LargeArgs g{};
        g.p0 = b->h_part_off[t];
        g.n_part = p;
        g.c0 = b->h_cons_off[t];
        g.n_cons = c;
        g.pid = b->d_partition_id;
        g.begin = b->d_begin_off;
        g.end = b->d_end_off;
        g.committed = b->d_committed_off;
        g.lag = b->d_lag;
        g.cons_rank = b->d_cons_rank;
        g.out_pid = b->d_out_partition;
        g.out_rank = b->d_out_member_rank;
        g.out_total = b->d_out_total_lag;
        g.reset_latest = (b->reset_mode == LA_RESET_LATEST) ? 1 : 0;
        g.no_sample_sort = (b->flags & LA_FLAG_NO_SAMPLE_SORT) ? 1 : ((b->flags & LA_FLAG_SAMPLE_TIGHT) ? 2 : 0);
        g.sort_multi_kernel = (b->flags & LA_FLAG_SORT_MULTIKERNEL) ? 1 : 0;
        g.no_run_merge = (b->flags & LA_FLAG_NO_RUN_MERGE) ? 1 : 0;
        g.no_moved_sort = (b->flags & LA_FLAG_NO_MOVED_SORT) ? 1 : 0;
        const bool bounded_large = (b->flags & LA_FLAG_BOUNDS) && b->max_lag_hint >= 0 && b->max_partition_id_hint >= 0;
        g.max_lag_hint = bounded_large ? b->max_lag_hint : -1;
        g.max_id_hint = bounded_large ? b->max_partition_id_hint : -1;
        g.status = ln.status();
        hipError_t e = hipSuccess;
        if (c > la::kLargeMaxConsumers) {
            if (int rc = flush()) return rc;
            if (argmin)
                return fail(ctx, LA_ESHAPE, "topic %lld has %lld consumers; LA_ALGO_ARGMIN (a test hook) holds at most %lld",
                            (long long)t, (long long)c, (long long)la::kLargeMaxConsumers);
            e = la::huge_topic_launch(ln.large, g, stream);
        } else if (argmin || p == 0 || (b->flags & LA_FLAG_SERIAL_LARGE)) {
            if (int rc = flush()) return rc;
            e = la::large_topic_launch(ln.large, g, argmin, stream);
        } else {
            if (group.size() >= kGroupTopics || (group_n > 0 && group_n + p > kGroupPartitions))
                if (int rc = flush()) return rc;
            group.push_back(g);
            group_n += p;
        }
        if (e != hipSuccess)
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "large topic %lld: %s", (long long)t,
                        hipGetErrorString(e));
    }
    return flush();
}

// The dispatcher shared by the host and device entry points.
int enqueue_batch(la_ctx* ctx, Lane& ln, const la_device_batch* b, hipStream_t stream) {
    if (b->n_topics < 0 || b->n_partitions < 0 || b->n_consumers < 0)
        return fail(ctx, LA_EINVAL, "negative size");
    if (b->n_topics == 0) return LA_OK;
    if (!b->d_part_off || !b->d_cons_off) return fail(ctx, LA_EINVAL, "null offsets");
    // (a batch whose topics have no partitions at all has nothing to write: its [0]-sized outputs may be anything)
    const bool wire_out = (b->flags & LA_FLAG_WIRE_OUT) != 0;
    if (b->n_partitions > 0 && !wire_out && (!b->d_out_partition || !b->d_out_member_rank)) return fail(ctx, LA_EINVAL, "null outputs");
    if (wire_out) {
        // the results in the all-gather's wire format straight from the kernels: the tile path's one-launch form only
        if (!b->d_out_wire || (b->wire_elem_bytes != 2 && b->wire_elem_bytes != 4) ||
            !la::wire_format_valid(b->wire_elem_bytes, b->wire_id_bits))
            return fail(ctx, LA_EINVAL, "LA_FLAG_WIRE_OUT: d_out_wire and a 2- or 4-byte wire format are required");
        if (b->algo != LA_ALGO_AUTO || (b->flags & LA_FLAG_RAGGED) ||
            !la::wave_tile_fits(b->max_partitions_per_topic, b->max_consumers_per_topic) || b->n_partitions >= ((int64_t)1 << 29) ||
            b->n_consumers >= ((int64_t)1 << 30) || (b->flags & LA_FLAG_INDEX64) || !(b->flags & LA_FLAG_BOUNDS) ||
            !la::wave_tile_always_packs(b->max_partitions_per_topic, b->max_consumers_per_topic, b->max_lag_hint,
                                        b->max_partition_id_hint))
            return fail(ctx, LA_EINVAL, "LA_FLAG_WIRE_OUT needs a batch of tile-sized topics (shape hint within 1024 x 64, no LA_FLAG_RAGGED), "
                                        "LA_FLAG_BOUNDS that prove every tile packs, and fewer than 2^29 partitions: run this batch "
                                        "without the flag and pack the results with la_pack_results_on");
    }
    if (b->n_partitions > 0 && (!b->d_partition_id || (!b->d_lag && (!b->d_end_off || !b->d_committed_off))))
        return fail(ctx, LA_EINVAL, "null per-partition input");
    if (b->n_consumers > 0 && !b->d_cons_rank) return fail(ctx, LA_EINVAL, "null cons_rank");
    if (!b->d_lag && b->reset_mode != LA_RESET_LATEST && !b->d_begin_off && b->n_partitions > 0)
        return fail(ctx, LA_EINVAL, "begin_off is required unless reset_mode is LA_RESET_LATEST");
    if (b->algo != LA_ALGO_AUTO && b->algo != LA_ALGO_ROUNDS && b->algo != LA_ALGO_ARGMIN && b->algo != LA_ALGO_ROUNDS_WIDE)
        return fail(ctx, LA_EINVAL, "unknown algo %d", b->algo);

    la::TileArgs a{};
    a.n_topics = b->n_topics;
    a.part_off = b->d_part_off;
    a.pid = b->d_partition_id;
    a.begin = b->d_begin_off;
    a.end = b->d_end_off;
    a.committed = b->d_committed_off;
    a.lag = b->d_lag;
    a.cons_off = b->d_cons_off;
    a.cons_rank = b->d_cons_rank;
    a.out_pid = b->d_out_partition;
    a.out_rank = b->d_out_member_rank;
    a.out_total = b->d_out_total_lag;
    a.status = ln.status();
    a.reset_latest = (b->reset_mode == LA_RESET_LATEST) ? 1 : 0;
    a.n_total = b->n_partitions;
    a.flags = b->flags & (LA_FLAG_INDEX64 | LA_FLAG_DEFER_WIDE);
    if (getenv("LA_TILE_ALWAYS_STAGE2")) a.flags |= la::kTileAlwaysStage2;       // (A/B hook, read at every call)
    if (wire_out) {
        a.flags |= la::kTileWireOut;
        a.out_wire = b->d_out_wire;
        a.wire_bytes = b->wire_elem_bytes;
        a.wire_id_bits = b->wire_id_bits;
    }
    a.topic_list = nullptr;
    a.k_total = b->n_consumers;
    if (int rc = reserve(ctx, ln.defer, la::wave_tile_defer_bytes(b->n_topics))) return rc;
    a.defer_list = (int32_t*)ln.defer.p;
    const bool argmin = (b->algo == LA_ALGO_ARGMIN);
    const int tile_mode = argmin ? 2 : (b->algo == LA_ALGO_ROUNDS_WIDE ? 1 : 0);

    // The counter pair alternates per LA_ALGO_AUTO launch: such a launch counts into one and its wide
    // kernel zeroes the other (idle by stream order), so no memset node sits between launches.
    // Under stream capture the arguments are frozen into the graph, so the alternation cannot work on replay: a
    // captured launch counts into a third word that a memset node clears first (its wide kernel "resets" a dummy).
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (tile_mode == 0) LA_HIP(ctx, hipStreamIsCapturing(stream, &capture));
    hipError_t counter_err = hipSuccess;
    // LA_FLAG_BOUNDS: when the caller's bounds prove that every tile of a launch packs, that launch defers nothing, needs no
    // wide kernel behind it and leaves the counter pair alone (it neither counts nor has a wide kernel to zero the idle one)
    const bool bounded = tile_mode == 0 && (b->flags & LA_FLAG_BOUNDS) && b->max_lag_hint >= 0 && b->max_partition_id_hint >= 0;
    auto proven = [&](int64_t mp, int64_t mc) {
        return bounded && la::wave_tile_always_packs(mp, mc, b->max_lag_hint, b->max_partition_id_hint);
    };
    auto next_counters = [&](la::TileArgs& t) {
        int32_t* pair = (int32_t*)(ln.d_status + 16);
        if (t.flags & la::kTileNoDefer) {
            t.defer_count = pair + (ln.launches & 1u);
            t.defer_count_next = pair + ((ln.launches + 1u) & 1u);
            return;
        }
        if (capture != hipStreamCaptureStatusNone) {
            t.defer_count = pair + 2;
            t.defer_count_next = pair + 3;
            const hipError_t e = hipMemsetAsync(pair + 2, 0, sizeof(int32_t), stream);
            if (e != hipSuccess) counter_err = e;
            return;
        }
        t.defer_count = pair + (ln.launches & 1u);
        t.defer_count_next = pair + ((ln.launches + 1u) & 1u);
        if (tile_mode == 0) ++ln.launches;
    };
    if (b->n_partitions == 0) {
        // nothing to assign; consumers of partition-less topics still report a total of 0
        if (b->d_out_total_lag && b->n_consumers > 0)
            LA_HIP(ctx, hipMemsetAsync(b->d_out_total_lag, 0, (size_t)b->n_consumers * sizeof(int64_t), stream));
        return LA_OK;
    }
    ln.large.prof.armed = (b->flags & LA_FLAG_PROFILE) != 0;
    if (ln.large.prof.armed) ln.large.prof.recorded = false;
    const bool have_host = b->h_part_off && b->h_cons_off;
    const bool fits_hint = la::wave_tile_fits(b->max_partitions_per_topic, b->max_consumers_per_topic);
    if (fits_hint && !(have_host && (b->flags & LA_FLAG_RAGGED) && tile_mode == 0)) {
        // the plain case: every topic fits a wave tile, one shape for all, nothing read on the host
        if (proven(b->max_partitions_per_topic, b->max_consumers_per_topic)) a.flags |= la::kTileNoDefer;
        next_counters(a);
        LA_HIP(ctx, counter_err);
        if (ln.tail_wanted && tile_mode == 0) a.tail = ln.tail;           // the batch is this one launch
        LA_HIP(ctx, la::wave_tile_launch(a, b->max_partitions_per_topic, b->max_consumers_per_topic, tile_mode, stream,
                                         &ln.tail_done));
        return LA_OK;
    }
    la_device_batch with_host;
    if (!have_host) {
        // The shape hint exceeds one wave tile and the caller kept no host copy of the offsets: fetch them (two
        // small copies and a wait on `stream` -- this call is then neither asynchronous nor capturable).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        LA_HIP(ctx, hipStreamIsCapturing(stream, &cap));
        if (cap != hipStreamCaptureStatusNone)
            return fail(ctx, LA_EINVAL, "shape hint exceeds one wave tile: h_part_off and h_cons_off are required "
                                        "while the stream is being captured");
        const size_t words = (size_t)b->n_topics + 1;
        ln.host_offsets.resize(2 * words);
        LA_HIP(ctx, hipMemcpyAsync(ln.host_offsets.data(), b->d_part_off, words * 8, hipMemcpyDeviceToHost, stream));
        LA_HIP(ctx, hipMemcpyAsync(ln.host_offsets.data() + words, b->d_cons_off, words * 8, hipMemcpyDeviceToHost, stream));
        LA_HIP(ctx, hipStreamSynchronize(stream));
        with_host = *b;
        with_host.h_part_off = ln.host_offsets.data();
        with_host.h_cons_off = ln.host_offsets.data() + words;
        b = &with_host;
    }

    // mixed or ragged shapes (see the dispatcher notes above)
    const bool use_block = !argmin && b->algo != LA_ALGO_ROUNDS_WIDE;   // the test-hook algos keep to tile + large
    BatchPlan plan;
    if (int rc = plan_batch(ctx, ln, b, tile_mode, use_block, &plan)) return rc;
    const int32_t* d_lists = nullptr;
    if (int rc = stage_topic_lists(ctx, ln, plan, b->n_topics, stream, &d_lists)) return rc;
    if (plan.classed) {
        for (int k = 0; k < kTileClasses; ++k) {
            if (plan.tile[k].n == 0) continue;
            la::TileArgs run = a;
            run.n_topics = plan.tile[k].n;
            run.topic_list = d_lists + plan.tile_at[k];
            if (proven(plan.tile[k].mp, plan.tile[k].mc)) run.flags |= la::kTileNoDefer;
            next_counters(run);
            LA_HIP(ctx, counter_err);
            LA_HIP(ctx, la::wave_tile_launch(run, plan.tile[k].mp, plan.tile[k].mc, tile_mode, stream));
        }
    } else if (plan.n_tile > 0) {
        la::TileArgs run = a;
        run.flags |= la::kTileSkipOversize;
        if (proven(plan.tile_mp, plan.tile_mc)) run.flags |= la::kTileNoDefer;
        next_counters(run);
        LA_HIP(ctx, counter_err);
        // (a tail only when nothing else of the batch runs behind this launch)
        if (ln.tail_wanted && tile_mode == 0 && plan.n_block_all == 0 && plan.n_large == 0) run.tail = ln.tail;
        LA_HIP(ctx, la::wave_tile_launch(run, plan.tile_mp, plan.tile_mc, tile_mode, stream, &ln.tail_done));
    }
    if (plan.n_block_all > 0)
        if (int rc = launch_block_topics(ctx, ln, b, plan, d_lists, stream)) return rc;
    if (plan.n_large > 0)
        if (int rc = launch_large_topics(ctx, ln, b, plan, argmin, stream)) return rc;
    return LA_OK;
}

// The device status word as an error of the call (most severe first).
int status_error(la_ctx* ctx, uint32_t st) {
    if (st & la::kStatusInternal)
        return fail(ctx, LA_EHIP, "internal error: a radix-sort look-back gave up waiting for an earlier tile");
    if (st & la::kStatusOrder)
        return fail(ctx, LA_EHIP, "internal error: a device radix sort left its result out of order");
    if (st & la::kStatusUnsorted)
        return fail(ctx, LA_EINVAL, "a topic's cons_rank segment is not strictly ascending");
    if (st & la::kStatusBounds)
        return fail(ctx, LA_EINVAL, "a lag or a partition id lies outside the bounds the caller gave (la_hint_next_call / LA_FLAG_BOUNDS)");
    if (st & la::kStatusSparse)
        return fail(ctx, LA_EINVAL, "none_index must hold ascending positions inside the batch");
    if (st & la::kStatusWire)
        return fail(ctx, LA_EINVAL, "a partition id or member rank does not fit the wire format given to la_pack_results_on");
    return fail(ctx, LA_ESHAPE, "a topic exceeds the batch's shape hint");
}

// Waits for `stream` and reports what the kernels flagged on this lane (one copy + one sync).
int sync_status(la_ctx* ctx, Lane& ln, hipStream_t stream) {
    LA_HIP(ctx, hipMemcpyAsync(ln.h_status, ln.d_status, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LA_HIP(ctx, hipStreamSynchronize(stream));
    const uint32_t st = *(volatile uint32_t*)ln.h_status;
    if (st) {
        LA_HIP(ctx, hipMemsetAsync(ln.d_status, 0, sizeof st, stream));
        LA_HIP(ctx, hipStreamSynchronize(stream));
        return status_error(ctx, st);
    }
    return LA_OK;
}

// ---- planner: contiguous topic ranges balanced by partition count -----------------------------------------
// The shard axis is the per-topic loop of assign(Map,Map) (Main.java:177-184): assignTopic touches only its own
// topic's bins, so any split at topic boundaries is valid; the kernels' cost is per partition, so the split
// balances partitions.  bounds[r] = first topic boundary at or after r/parts of the partitions.  Every topic
// lands in exactly one range; ranges may be empty.  Used for shards (devices) and for the chunks of one shard.
void plan_ranges(const int64_t* po, int32_t t_begin, int32_t t_end, int parts, int32_t* bounds) {
    const int64_t base = po[t_begin], total = po[t_end] - base;
    bounds[0] = t_begin;
    for (int r = 1; r < parts; ++r) {
        const int64_t target = base + (total / parts) * r + (total % parts) * r / parts;
        const int64_t* lo = std::lower_bound(po + bounds[r - 1], po + t_end + 1, target);
        int64_t t = lo - po;
        if (t > t_end) t = t_end;
        bounds[r] = (int32_t)t;
    }
    bounds[parts] = t_end;
}

constexpr int kMaxShards = 64;
constexpr int64_t kMinShardPartitions = 1 << 16;   // below this a second device costs more than it saves
constexpr int64_t kMinChunkPartitions = 1 << 19;   // a chunk's copies must be long enough to hide a kernel
constexpr int64_t kMidChunkPartitions = 100000;    // pageable arrays: from two such chunks on, a mid-size shard runs on two lanes
constexpr int kMaxChunks = 64;

// One host-buffer assign call (all pointers are the caller's host arrays).
struct HostCall {
    int32_t T = 0;
    const int64_t* part_off = nullptr;
    const int32_t* pid = nullptr;
    const int64_t *begin = nullptr, *end = nullptr, *committed = nullptr, *lag = nullptr;
    int32_t reset_mode = 0;
    const int64_t* cons_off = nullptr;
    const int32_t* cons_rank = nullptr;
    int32_t *out_pid = nullptr, *out_rank = nullptr;
    int64_t* out_total = nullptr;
    bool use_begin = false;
    // la_assign_batch_sparse: `begin` arrives as (position, value) pairs for the partitions without a committed offset; the
    // dense array the kernels read is rebuilt on the device (zero + scatter), chunk by chunk
    bool mapped = false;                 // every array of the call is pinned AND device-mapped: the kernels read / write it in place
    bool sparse = false;
    int64_t n_none = 0;
    const int64_t *none_index = nullptr, *none_begin = nullptr;
    Shape shape;
    // la_assign_batch_grouped: every member's list is wanted instead of the ungrouped arrays.  A small call builds it in the same
    // staging buffer and brings it back in its one D2H (grouped_done = true); any other call leaves it to group_last_impl.
    int32_t g_members = -1;
    int64_t* g_off = nullptr;
    int32_t *g_topic = nullptr, *g_part = nullptr;
    bool* grouped_done = nullptr;
    // la_hint_next_call: the caller's bounds on every lag and partition id of the call (LA_FLAG_BOUNDS of every chunk's batch)
    bool bounded = false;
    int64_t max_lag = 0, max_id = 0;
};

// The caller's hints on a chunk's / shard's device batch: bounds that hold for the whole call hold for every part of it.
void apply_hints(const HostCall& c, la_device_batch& b) {
    if (!c.bounded) return;
    b.flags |= LA_FLAG_BOUNDS;
    b.max_lag_hint = c.max_lag;
    b.max_partition_id_hint = c.max_id;
}

struct ShardPlan {
    int32_t t0 = 0, t1 = 0;                 // the shard's topics in the caller's numbering
    int64_t P0 = 0, K0 = 0, n = 0, k = 0;   // its partitions / consumer entries: first position, count
    const int64_t *lpo = nullptr, *lco = nullptr;   // offsets rebased to the shard (host)
    std::vector<int32_t> chunk;             // chunk boundaries, shard-local topic indices
    std::vector<int64_t> none_at;           // sparse begin: first list entry of every chunk (+ the end), caller's numbering
    std::atomic<int> next{0};
};

// Buffers + offsets of one shard, on the calling thread, before its lanes start.
int prepare_shard(la_ctx* ctx, const HostCall& c, Shard& sh, ShardPlan& sp) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    const int32_t Ts = sp.t1 - sp.t0;
    if (sp.P0 == 0 && sp.K0 == 0) {
        sp.lpo = c.part_off + sp.t0;
        sp.lco = c.cons_off + sp.t0;
    } else {
        sh.local_part_off.resize((size_t)Ts + 1);
        sh.local_cons_off.resize((size_t)Ts + 1);
        for (int32_t t = 0; t <= Ts; ++t) {
            sh.local_part_off[t] = c.part_off[sp.t0 + t] - sp.P0;
            sh.local_cons_off[t] = c.cons_off[sp.t0 + t] - sp.K0;
        }
        sp.lpo = sh.local_part_off.data();
        sp.lco = sh.local_cons_off.data();
    }
    const size_t nb8 = (size_t)sp.n * 8, nb4 = (size_t)sp.n * 4, kb8 = (size_t)sp.k * 8, kb4 = (size_t)sp.k * 4;
    const size_t tb = (size_t)(Ts + 1) * 8;
    int rc;
    if ((rc = reserve(ctx, sh.part_off, tb)) || (rc = reserve(ctx, sh.cons_off, tb))) return rc;
    if (c.mapped) {
        // the kernels work on the caller's arrays in place: only what is NOT the caller's lives here -- the results of a call
        // that keeps them on the device, and the dense `begin` rebuilt from a sparse list (indexed by the caller's positions,
        // begin[0] included: the tile kernels park idle lanes there)
        if (!c.out_pid && ((rc = reserve(ctx, sh.out_pid, nb4 + 16)) || (rc = reserve(ctx, sh.out_rank, nb4 + 16)))) return rc;
        if (!c.out_total && (rc = reserve(ctx, sh.out_total, kb8 + 16))) return rc;
        if (c.sparse && (rc = reserve(ctx, sh.begin, (size_t)c.shape.n * 8 + 16))) return rc;
    } else if ((rc = reserve(ctx, sh.pid, nb4 + 16)) || (rc = reserve(ctx, sh.end, nb8 + 16)) ||
               (rc = reserve(ctx, sh.committed, c.lag ? 16 : nb8 + 16)) ||
               (rc = reserve(ctx, sh.begin, c.use_begin ? nb8 + 16 : 16)) ||
               (rc = reserve(ctx, sh.cons_rank, kb4 + 16)) || (rc = reserve(ctx, sh.out_pid, nb4 + 16)) ||
               (rc = reserve(ctx, sh.out_rank, nb4 + 16)) || (rc = reserve(ctx, sh.out_total, kb8 + 16))) {
        return rc;
    }
    hipStream_t st = sh.lanes[0].stream;
    LA_HIP(ctx, hipMemcpyAsync(sh.part_off.p, sp.lpo, tb, hipMemcpyHostToDevice, st));
    LA_HIP(ctx, hipMemcpyAsync(sh.cons_off.p, sp.lco, tb, hipMemcpyHostToDevice, st));
    LA_HIP(ctx, hipEventRecord(sh.ready, st));

    // chunks: enough of them that the copies of one overlap the kernels and the opposite copies of another
    int n_chunks = 1;
    if (c.mapped) {
        n_chunks = 1;                                    // the kernels pull the shard's bytes themselves: nothing to overlap
    } else if (ctx->split_always) {
        n_chunks = Ts < 3 ? (Ts > 0 ? Ts : 1) : 3;
    } else if ((int)sh.lanes.size() > 1 || ctx->last_pipeline == 2) {
        // (pinned arrays, one enqueueing thread: finer chunks -- the call ends one chunk's kernels + result copy after
        //  the last input byte has landed)
        int64_t target = ctx->chunk_partitions > 0 ? ctx->chunk_partitions : sp.n / 16;
        if (ctx->chunk_partitions <= 0 && target < kMinChunkPartitions) target = kMinChunkPartitions;
        // three-stream form: chunks of ~1 M partitions (28 MB of input) measured best on the 25.6 M-partition batch -- 13.6 ms
        // against 18-20 ms at 512 K, 14.4 at 2 M, 14.5 with few chunks that are small at both ends (profiles/archive/r03_host_probe.txt)
        // (by bytes: 1 M partitions of the dense offsets form = 28 MB of input; the sparse-begin and the lags forms move fewer
        //  bytes per partition and take proportionally more partitions per chunk -- a chunk's copies cost ~20 us each before their
        //  first byte moves, whatever they carry)
        if (ctx->chunk_partitions <= 0 && ctx->last_pipeline == 2)
            target = ((int64_t)28 << 20) / (c.lag ? 12 : (c.use_begin && !c.sparse ? 28 : 20));
        int64_t want = (sp.n + target - 1) / target;
        // pageable arrays, a mid-size shard (one chunk by the rule above): every copy of a pageable array costs its issuing thread
        // ~25 us before the first byte moves, whatever it carries, and a chunk is ten of them -- two lanes side by side still
        // halve the bytes behind each (256 000 partitions: 386 -> 326 us), more chunks than that lose (4: 421 us, 8: 501;
        // profiles/r05_b_latency_chunks.txt).  The lanes' threads are parked in the context, so the second one costs a wake-up.
        if (ctx->chunk_partitions <= 0 && ctx->last_pipeline == 1 && want == 1 && sp.n >= 2 * kMidChunkPartitions) want = 2;
        n_chunks = (int)(want < 1 ? 1 : (want > kMaxChunks ? kMaxChunks : want));
        if (n_chunks > Ts) n_chunks = Ts > 0 ? Ts : 1;
    }
    sp.chunk.resize((size_t)n_chunks + 1);
    plan_ranges(sp.lpo, 0, Ts, n_chunks, sp.chunk.data());
    sp.next.store(0);
    if (c.sparse) {
        // the list is ascending by contract, so a chunk's entries are one slice of it.  The first chunk of the call starts at
        // entry 0 and the last ends at n_none whatever the values say: every entry is then looked at by exactly one chunk, and
        // one that is not inside its chunk's positions (an unsorted or out-of-range list) is caught by the scatter kernel.
        sp.none_at.resize((size_t)n_chunks + 1);
        const int64_t* nb = c.none_index;
        const int64_t* ne = c.none_index + c.n_none;
        for (int ci = 0; ci <= n_chunks; ++ci) {
            const int64_t pos = sp.P0 + sp.lpo[sp.chunk[(size_t)ci]];
            sp.none_at[(size_t)ci] = std::lower_bound(nb, ne, pos) - nb;
        }
        if (sp.P0 == 0) sp.none_at[0] = 0;
        if (sp.P0 + sp.n == c.shape.n) sp.none_at[(size_t)n_chunks] = c.n_none;
        const size_t m = (size_t)(sp.none_at[(size_t)n_chunks] - sp.none_at[0]);
        if ((rc = reserve(ctx, sh.none_idx, m * 8 + 16)) || (rc = reserve(ctx, sh.none_val, m * 8 + 16))) return rc;
    }
    return LA_OK;
}

// Sparse begin offsets of chunk ci: `copy` (non-null) uploads the chunk's slice of the (position, begin) list, `kern`
// (non-null) zeroes the chunk's part of the dense array and scatters the slice into it.  One stream may play both roles.
int sparse_begin_chunk(la_ctx* ctx, const HostCall& c, Shard& sh, const ShardPlan& sp, int ci, Lane& ln, hipStream_t copy,
                       hipStream_t kern) {
    const int64_t j0 = sp.none_at[(size_t)ci], j1 = sp.none_at[(size_t)ci + 1], at = j0 - sp.none_at[0];
    const size_t m = (size_t)(j1 - j0);
    const int32_t a = sp.chunk[(size_t)ci], z = sp.chunk[(size_t)ci + 1];
    const int64_t p0 = sp.lpo[a], p1 = sp.lpo[z];
    int64_t* d_idx = (int64_t*)sh.none_idx.p + at;
    int64_t* d_val = (int64_t*)sh.none_val.p + at;
    if (copy && m) {
        LA_HIP(ctx, hipMemcpyAsync(d_idx, c.none_index + j0, m * 8, hipMemcpyHostToDevice, copy));
        LA_HIP(ctx, hipMemcpyAsync(d_val, c.none_begin + j0, m * 8, hipMemcpyHostToDevice, copy));
    }
    if (kern) {
        if (p1 > p0) LA_HIP(ctx, hipMemsetAsync((int64_t*)sh.begin.p + p0, 0, (size_t)(p1 - p0) * 8, kern));
        LA_HIP(ctx, la::sparse_begin_launch((int64_t)m, d_idx, d_val, sp.P0, sp.P0 + p0, sp.P0 + p1, (int64_t*)sh.begin.p,
                                            ln.d_status, kern));
    }
    return LA_OK;
}

// One lane of one shard: takes the shard's chunks in order until none is left.  Per chunk: H2D of its slices,
// the kernels over its topics, D2H of its results -- all on the lane's stream; the lanes of a shard overlap
// each other (the copies of a pageable buffer block their host thread, which is why lanes are threads).
int run_lane(la_ctx* ctx, const HostCall& c, Shard& sh, ShardPlan& sp, int lane_idx, std::atomic<bool>& stop) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    Lane& ln = sh.lanes[(size_t)lane_idx];
    hipStream_t st = ln.stream;
    if (lane_idx > 0) LA_HIP(ctx, hipStreamWaitEvent(st, sh.ready, 0));
    const int n_chunks = (int)sp.chunk.size() - 1;
    auto run_chunk = [&](int ci) -> int {
        const int32_t a = sp.chunk[(size_t)ci], z = sp.chunk[(size_t)ci + 1];
        const int64_t p0 = sp.lpo[a], p1 = sp.lpo[z], k0 = sp.lco[a], k1 = sp.lco[z];
        const int64_t gp = sp.P0 + p0, gk = sp.K0 + k0;              // the chunk in the caller's arrays
        const size_t np = (size_t)(p1 - p0), nk = (size_t)(k1 - k0);
        if (np) {
            LA_HIP(ctx, hipMemcpyAsync((int32_t*)sh.pid.p + p0, c.pid + gp, np * 4, hipMemcpyHostToDevice, st));
            if (c.lag) {
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.end.p + p0, c.lag + gp, np * 8, hipMemcpyHostToDevice, st));
            } else {
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.end.p + p0, c.end + gp, np * 8, hipMemcpyHostToDevice, st));
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.committed.p + p0, c.committed + gp, np * 8,
                                           hipMemcpyHostToDevice, st));
                if (c.use_begin && !c.sparse)
                    LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.begin.p + p0, c.begin + gp, np * 8, hipMemcpyHostToDevice, st));
                if (c.sparse)
                    if (int rc = sparse_begin_chunk(ctx, c, sh, sp, ci, ln, st, st)) return rc;
            }
        }
        if (c.sparse && !np && sp.none_at[(size_t)ci + 1] > sp.none_at[(size_t)ci])      // entries where there are no partitions:
            if (int rc = sparse_begin_chunk(ctx, c, sh, sp, ci, ln, st, st)) return rc;       // the kernel reports them
        if (nk) {
            LA_HIP(ctx, hipMemcpyAsync((int32_t*)sh.cons_rank.p + k0, c.cons_rank + gk, nk * 4, hipMemcpyHostToDevice, st));
            // the ascending-rank contract of cons_rank is checked on the device (one pass over K entries there
            // instead of ~1 ns per entry of host time), reported by sync_status as LA_EINVAL
            LA_HIP(ctx, la::check_consumers_launch(z - a, (const int64_t*)sh.cons_off.p + a,
                                                   (const int32_t*)sh.cons_rank.p, ln.d_status, st));
        }
        // The chunk is a topic sub-range of the shard's batch: same arrays, offsets advanced to its first topic.
        // n_partitions / n_consumers stay the shard's totals -- they bound the kernels' clamped loads, which may
        // touch (and ignore) elements of neighbouring chunks.
        la_device_batch b{};
        b.n_topics = z - a;
        b.reset_mode = c.reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
        b.algo = LA_ALGO_AUTO;
        b.n_partitions = sp.n;
        b.n_consumers = sp.k;
        b.max_partitions_per_topic = c.shape.max_p;
        b.max_consumers_per_topic = c.shape.max_c;
        b.d_part_off = (const int64_t*)sh.part_off.p + a;
        b.d_partition_id = (const int32_t*)sh.pid.p;
        b.d_begin_off = c.use_begin ? (const int64_t*)sh.begin.p : nullptr;
        b.d_end_off = (const int64_t*)sh.end.p;
        b.d_committed_off = (const int64_t*)sh.committed.p;
        b.d_lag = c.lag ? (const int64_t*)sh.end.p : nullptr;
        b.d_cons_off = (const int64_t*)sh.cons_off.p + a;
        b.d_cons_rank = (const int32_t*)sh.cons_rank.p;
        b.d_out_partition = (int32_t*)sh.out_pid.p;
        b.d_out_member_rank = (int32_t*)sh.out_rank.p;
        b.d_out_total_lag = c.out_total ? (int64_t*)sh.out_total.p : nullptr;
        b.h_part_off = sp.lpo + a;
        b.h_cons_off = sp.lco + a;
        b.flags = LA_FLAG_RAGGED;        // the offsets are on the host anyway: let the dispatcher look at the shapes
        apply_hints(c, b);
        if (np == 0) {
            // nothing to assign in this chunk; its consumers still report a total of 0
            if (c.out_total && nk)
                LA_HIP(ctx, hipMemsetAsync((int64_t*)sh.out_total.p + k0, 0, nk * 8, st));
        } else if (int rc = enqueue_batch(ctx, ln, &b, st)) {
            return rc;
        }
        if (np && c.out_pid) {
            LA_HIP(ctx, hipMemcpyAsync(c.out_pid + gp, (const int32_t*)sh.out_pid.p + p0, np * 4, hipMemcpyDeviceToHost, st));
            LA_HIP(ctx, hipMemcpyAsync(c.out_rank + gp, (const int32_t*)sh.out_rank.p + p0, np * 4, hipMemcpyDeviceToHost, st));
        }
        if (c.out_total && nk)
            LA_HIP(ctx, hipMemcpyAsync(c.out_total + gk, (const int64_t*)sh.out_total.p + k0, nk * 8, hipMemcpyDeviceToHost, st));
        return LA_OK;
    };
    int rc = LA_OK;
    while (!stop.load(std::memory_order_relaxed)) {
        const int ci = sp.next.fetch_add(1);
        if (ci >= n_chunks) break;
        if ((rc = run_chunk(ci))) break;
    }
    const int rs = sync_status(ctx, ln, st);     // also when stopping early: nothing of this lane stays in flight
    return rc ? rc : rs;
}

// true when [p, p + bytes) is pinned host memory the runtime knows (la_host_alloc, hipHostMalloc, hipHostRegister): copies
// to and from it are plain DMA and hipMemcpyAsync returns at once.  NULL / zero bytes count as pinned.
bool is_pinned(const void* p, size_t bytes) {
    if (!p || bytes == 0) return true;
    // first AND last byte: a buffer registered only in part (hipHostRegister of a sub-range), or one that runs past its
    // allocation into another, must not take the thread-less pipeline, which assumes plain DMA over the whole range
    hipPointerAttribute_t at{}, az{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess ||
        hipPointerGetAttributes(&az, (const char*)p + (bytes - 1)) != hipSuccess) {
        (void)hipGetLastError();                          // an ordinary malloc'ed pointer: not an error of ours
        return false;
    }
    if (at.type != hipMemoryTypeHost || az.type != hipMemoryTypeHost) return false;
    // the same registration: where the runtime reports the host / device views, they advance together from first to last byte
    const ptrdiff_t span = (ptrdiff_t)(bytes - 1);
    if (at.hostPointer && az.hostPointer && (const char*)az.hostPointer - (const char*)at.hostPointer != span) return false;
    if ((at.devicePointer == nullptr) != (az.devicePointer == nullptr)) return false;
    if (at.devicePointer && (const char*)az.devicePointer - (const char*)at.devicePointer != span) return false;
    return true;
}

bool call_is_pinned(const HostCall& c) {
    const size_t n = (size_t)c.shape.n, k = (size_t)c.shape.k;
    return is_pinned(c.pid, n * 4) && is_pinned(c.end, n * 8) && is_pinned(c.committed, n * 8) && is_pinned(c.lag, n * 8) &&
           (!c.use_begin || c.sparse || is_pinned(c.begin, n * 8)) &&
           (!c.sparse || (is_pinned(c.none_index, (size_t)c.n_none * 8) && is_pinned(c.none_begin, (size_t)c.n_none * 8))) &&
           is_pinned(c.cons_rank, k * 4) && is_pinned(c.out_pid, n * 4) &&
           is_pinned(c.out_rank, n * 4) && is_pinned(c.out_total, k * 8);
}

// One shard of a call whose arrays are all pinned: no worker threads.  The calling thread enqueues, per chunk,
//   copy_in[]: H2D of the chunk's slices, chunks alternating over two streams -> event "in"
//   lane 0   : wait "in"; check_consumers; the kernels over its topics     -> event "out"
//   copy_out : wait "out"; D2H of the chunk's results straight into the caller's arrays
// so the link carries input bytes back to back from the first chunk to the last (the H2D leg IS the floor of the call:
// 717 MB at the 57 GB/s this link sustains = 12.5 ms for the 25.6 M-partition batch, profiles/archive/r03_pcie_probe.txt) while
// kernels and result copies of earlier chunks run beside it.  Returns after enqueueing; finish_shard_async waits.
int run_shard_async(la_ctx* ctx, const HostCall& c, Shard& sh, ShardPlan& sp) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    for (hipStream_t& st : sh.copy_in)
        if (!st) LA_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (!sh.copy_out) LA_HIP(ctx, hipStreamCreateWithFlags(&sh.copy_out, hipStreamNonBlocking));
    const int n_chunks = (int)sp.chunk.size() - 1;
    while ((int)sh.chunk_ev.size() < 2 * n_chunks + 1) {
        hipEvent_t e = nullptr;
        LA_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sh.chunk_ev.push_back(e);
    }
    Lane& ln = sh.lanes[0];
    hipStream_t sk = ln.stream, so = sh.copy_out;
    // the consumer ranks of the whole shard go up once (4 B per entry: too small to be worth a copy per chunk)
    // ... and so does the shard's slice of a sparse begin list (16 B per listed partition: one copy pair, not one per chunk --
    // every copy costs ~20 us of engine latency before its first byte moves)
    const size_t m_all = c.sparse ? (size_t)(sp.none_at[(size_t)n_chunks] - sp.none_at[0]) : 0;
    if (sp.k || m_all) {
        if (sp.k)
            LA_HIP(ctx, hipMemcpyAsync(sh.cons_rank.p, c.cons_rank + sp.K0, (size_t)sp.k * 4, hipMemcpyHostToDevice, sh.copy_in[1]));
        if (m_all) {
            LA_HIP(ctx, hipMemcpyAsync(sh.none_idx.p, c.none_index + sp.none_at[0], m_all * 8, hipMemcpyHostToDevice, sh.copy_in[1]));
            LA_HIP(ctx, hipMemcpyAsync(sh.none_val.p, c.none_begin + sp.none_at[0], m_all * 8, hipMemcpyHostToDevice, sh.copy_in[1]));
        }
        LA_HIP(ctx, hipEventRecord(sh.chunk_ev[(size_t)(2 * n_chunks)], sh.copy_in[1]));
        LA_HIP(ctx, hipStreamWaitEvent(sk, sh.chunk_ev[(size_t)(2 * n_chunks)], 0));
    }
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int32_t a = sp.chunk[(size_t)ci], z = sp.chunk[(size_t)ci + 1];
        const int64_t p0 = sp.lpo[a], p1 = sp.lpo[z], k0 = sp.lco[a], k1 = sp.lco[z];
        const int64_t gp = sp.P0 + p0, gk = sp.K0 + k0;
        const size_t np = (size_t)(p1 - p0), nk = (size_t)(k1 - k0);
        hipEvent_t ev_in = sh.chunk_ev[(size_t)(2 * ci)], ev_out = sh.chunk_ev[(size_t)(2 * ci + 1)];
        hipStream_t si = sh.copy_in[ci % Shard::kCopyIn];
        if (np) {
            LA_HIP(ctx, hipMemcpyAsync((int32_t*)sh.pid.p + p0, c.pid + gp, np * 4, hipMemcpyHostToDevice, si));
            if (c.lag) {
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.end.p + p0, c.lag + gp, np * 8, hipMemcpyHostToDevice, si));
            } else {
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.end.p + p0, c.end + gp, np * 8, hipMemcpyHostToDevice, si));
                LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.committed.p + p0, c.committed + gp, np * 8, hipMemcpyHostToDevice, si));
                if (c.use_begin && !c.sparse)
                    LA_HIP(ctx, hipMemcpyAsync((int64_t*)sh.begin.p + p0, c.begin + gp, np * 8, hipMemcpyHostToDevice, si));
            }
            LA_HIP(ctx, hipEventRecord(ev_in, si));
            LA_HIP(ctx, hipStreamWaitEvent(sk, ev_in, 0));
            if (c.sparse)
                if (int rc = sparse_begin_chunk(ctx, c, sh, sp, ci, ln, nullptr, sk)) return rc;
        }
        if (c.sparse && !np && sp.none_at[(size_t)ci + 1] > sp.none_at[(size_t)ci])      // entries where there are no partitions:
            if (int rc = sparse_begin_chunk(ctx, c, sh, sp, ci, ln, nullptr, sk)) return rc;  // the kernel reports them
        if (nk)
            LA_HIP(ctx, la::check_consumers_launch(z - a, (const int64_t*)sh.cons_off.p + a, (const int32_t*)sh.cons_rank.p,
                                                   ln.d_status, sk));
        la_device_batch b{};
        b.n_topics = z - a;
        b.reset_mode = c.reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
        b.algo = LA_ALGO_AUTO;
        b.n_partitions = sp.n;
        b.n_consumers = sp.k;
        b.max_partitions_per_topic = c.shape.max_p;
        b.max_consumers_per_topic = c.shape.max_c;
        b.d_part_off = (const int64_t*)sh.part_off.p + a;
        b.d_partition_id = (const int32_t*)sh.pid.p;
        b.d_begin_off = c.use_begin ? (const int64_t*)sh.begin.p : nullptr;
        b.d_end_off = (const int64_t*)sh.end.p;
        b.d_committed_off = (const int64_t*)sh.committed.p;
        b.d_lag = c.lag ? (const int64_t*)sh.end.p : nullptr;
        b.d_cons_off = (const int64_t*)sh.cons_off.p + a;
        b.d_cons_rank = (const int32_t*)sh.cons_rank.p;
        b.d_out_partition = (int32_t*)sh.out_pid.p;
        b.d_out_member_rank = (int32_t*)sh.out_rank.p;
        b.d_out_total_lag = c.out_total ? (int64_t*)sh.out_total.p : nullptr;
        b.h_part_off = sp.lpo + a;
        b.h_cons_off = sp.lco + a;
        b.flags = LA_FLAG_RAGGED;
        apply_hints(c, b);
        if (np == 0) {
            if (c.out_total && nk) LA_HIP(ctx, hipMemsetAsync((int64_t*)sh.out_total.p + k0, 0, nk * 8, sk));
        } else if (int rc = enqueue_batch(ctx, ln, &b, sk)) {
            return rc;
        }
        LA_HIP(ctx, hipEventRecord(ev_out, sk));
        const bool any_out = (np && c.out_pid) || (c.out_total && nk);
        if (any_out) LA_HIP(ctx, hipStreamWaitEvent(so, ev_out, 0));
        if (np && c.out_pid) {
            LA_HIP(ctx, hipMemcpyAsync(c.out_pid + gp, (const int32_t*)sh.out_pid.p + p0, np * 4, hipMemcpyDeviceToHost, so));
            LA_HIP(ctx, hipMemcpyAsync(c.out_rank + gp, (const int32_t*)sh.out_rank.p + p0, np * 4, hipMemcpyDeviceToHost, so));
        }
        if (c.out_total && nk)
            LA_HIP(ctx, hipMemcpyAsync(c.out_total + gk, (const int64_t*)sh.out_total.p + k0, nk * 8, hipMemcpyDeviceToHost, so));
    }
    return LA_OK;
}

// ---- pinned AND mapped caller arrays: no copies at all ------------------------------------------------------------------
// la_host_alloc memory (hipHostMalloc: portable, mapped) has a device address.  The tile kernels issue all the loads of a tile
// back to back and touch every input byte exactly once, so letting THEM pull the batch across PCIe needs no staging, no
// chunks and no copy engine: measured on the 25.6 M-partition target, 10.1 ms per call against 13.6 ms for the three-stream
// copy pipeline (tools/mapped_probe.py: 52-55 GB/s of kernel loads over the link, whose one-copy floor is 57 GB/s) -- and
// `begin` crosses the link only where a partition has no committed offset, because that is all the kernels read of it.
// Results are written straight into the caller's arrays.  Every shard works on the SAME arrays (positions are the caller's:
// part_off stays absolute), on its own range of topics.
template <typename T>
T* mapped_ptr(T* host) {
    if (!host) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, (void*)host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (T*)d;
}

bool call_is_mapped(const HostCall& c) {
    // first AND last byte of every array: an array registered only in part (hipHostRegister of a sub-range), or one
    // that runs past its registration, must not be handed to kernels that read and write it in place -- its device view has to
    // exist at both ends and advance with the host view in between; anything else takes a copying pipeline.
    auto ok = [](const void* p, size_t bytes) {
        if (p == nullptr || bytes == 0) return true;
        const char* d0 = mapped_ptr((const char*)p);
        const char* d1 = mapped_ptr((const char*)p + (bytes - 1));
        return d0 != nullptr && d1 != nullptr && (size_t)(d1 - d0) == bytes - 1;
    };
    const size_t n = (size_t)c.shape.n, k = (size_t)c.shape.k, tb = ((size_t)c.T + 1) * 8, m = (size_t)c.n_none * 8;
    return ok(c.part_off, tb) && ok(c.cons_off, tb) && ok(c.pid, n * 4) && ok(c.end, n * 8) && ok(c.committed, n * 8) &&
           ok(c.lag, n * 8) && ok(c.begin, n * 8) && ok(c.none_index, m) && ok(c.none_begin, m) && ok(c.cons_rank, k * 4) &&
           ok(c.out_pid, n * 4) && ok(c.out_rank, n * 4) && ok(c.out_total, k * 8);
}

int run_shard_mapped(la_ctx* ctx, const HostCall& c, Shard& sh, ShardPlan& sp) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    Lane& ln = sh.lanes[0];
    hipStream_t st = ln.stream;
    const int32_t Ts = sp.t1 - sp.t0;
    const int64_t* m_part_off = mapped_ptr(c.part_off);
    const int64_t* m_cons_off = mapped_ptr(c.cons_off);
    const int32_t* m_cons_rank = mapped_ptr(c.cons_rank);
    if (sp.k)
        LA_HIP(ctx, la::check_consumers_launch(Ts, m_cons_off + sp.t0, m_cons_rank, ln.d_status, st));
    const int64_t* d_begin = c.use_begin ? mapped_ptr(c.begin) : nullptr;
    if (c.sparse) {
        int64_t* db = (int64_t*)sh.begin.p;
        const int64_t j0 = sp.none_at.front(), j1 = sp.none_at.back();
        LA_HIP(ctx, hipMemsetAsync(db, 0, 8, st));                               // begin[0]: where idle lanes park
        if (sp.n) LA_HIP(ctx, hipMemsetAsync(db + sp.P0, 0, (size_t)sp.n * 8, st));
        LA_HIP(ctx, la::sparse_begin_launch(j1 - j0, mapped_ptr(c.none_index) + j0, mapped_ptr(c.none_begin) + j0, 0, sp.P0,
                                            sp.P0 + sp.n, db, ln.d_status, st));
        d_begin = db;
    }
    la_device_batch b{};
    b.n_topics = Ts;
    b.reset_mode = c.reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
    b.algo = LA_ALGO_AUTO;
    b.n_partitions = c.shape.n;                                                  // (positions are the caller's: the bound is too)
    b.n_consumers = c.shape.k;
    b.max_partitions_per_topic = c.shape.max_p;
    b.max_consumers_per_topic = c.shape.max_c;
    b.d_part_off = m_part_off + sp.t0;
    b.d_partition_id = mapped_ptr(c.pid);
    b.d_begin_off = d_begin;
    b.d_end_off = mapped_ptr(c.lag ? c.lag : c.end);
    b.d_committed_off = mapped_ptr(c.committed);
    b.d_lag = c.lag ? mapped_ptr(c.lag) : nullptr;
    b.d_cons_off = m_cons_off + sp.t0;
    b.d_cons_rank = m_cons_rank;
    // results the caller asked for land in its arrays; those it leaves on the device go to the shard's own buffers, addressed
    // by the caller's positions (only positions of this shard's topics are ever stored)
    // (the shard's own buffers hold positions [P0, P0 + n) only; their "position 0" is an address computed in integers, never
    //  dereferenced below P0)
    auto rebased = [&](void* base) { return (int32_t*)((uintptr_t)base - (uintptr_t)sp.P0 * sizeof(int32_t)); };
    b.d_out_partition = c.out_pid ? mapped_ptr(c.out_pid) : rebased(sh.out_pid.p);
    b.d_out_member_rank = c.out_pid ? mapped_ptr(c.out_rank) : rebased(sh.out_rank.p);
    b.d_out_total_lag = c.out_total ? mapped_ptr(c.out_total) : nullptr;
    b.h_part_off = c.part_off + sp.t0;
    b.h_cons_off = c.cons_off + sp.t0;
    b.flags = LA_FLAG_RAGGED;
    apply_hints(c, b);
    if (sp.n == 0) {
        if (c.out_total && sp.k) LA_HIP(ctx, hipMemsetAsync(mapped_ptr(c.out_total) + sp.K0, 0, (size_t)sp.k * 8, st));
        return LA_OK;
    }
    return enqueue_batch(ctx, ln, &b, st);
}

// Waits for everything run_shard_async enqueued on this shard (also after an error elsewhere: nothing stays in flight).
int finish_shard_async(la_ctx* ctx, Shard& sh) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    const int rs = sync_status(ctx, sh.lanes[0], sh.lanes[0].stream);
    hipError_t e1 = hipSuccess;
    for (hipStream_t st : sh.copy_in)
        if (st) { const hipError_t e = hipStreamSynchronize(st); if (e != hipSuccess) e1 = e; }
    hipError_t e2 = sh.copy_out ? hipStreamSynchronize(sh.copy_out) : hipSuccess;
    if (rs) return rs;
    if (e1 != hipSuccess || e2 != hipSuccess)
        return fail(ctx, LA_EHIP, "copy stream: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return LA_OK;
}

struct WorkerResult {
    int rc = LA_OK;
    std::string msg;
};

// Runs fn(i) for i in [0, n): i = 0 on the calling thread, the rest on the context's parked threads.  Each call reports
// into its own slot; the first failure (by index) becomes the context's error.
template <typename F>
int run_workers(la_ctx* ctx, int n, F&& fn) {
    std::vector<WorkerResult> res((size_t)n);
    auto body = [&](int i) {
        t_err_sink = &res[(size_t)i].msg;
        try {
            res[(size_t)i].rc = fn(i);
        } catch (...) {
            res[(size_t)i].rc = LA_ENOMEM;
            res[(size_t)i].msg = "exception in a worker thread";
        }
        t_err_sink = nullptr;
    };
    ctx->pool.run(n, body);                      // parked threads of the context: no spawn, no join per call
    for (int i = 0; i < n; ++i)
        if (res[(size_t)i].rc != LA_OK) {
            ctx->err = res[(size_t)i].msg;
            return res[(size_t)i].rc;
        }
    return LA_OK;
}

// ---- small batches: one H2D, the kernels, one D2H ----------------------------------------------------------------
// A real group leader's rebalance is ONE call over a few dozen topics.  The chunked pipeline above is built for bandwidth;
// at this size the call is API latency: 7 H2D copies of caller arrays, 3 D2H copies, the status word, ~8 us each -- 105 us
// for a three-partition batch.  Here the inputs are packed into one pinned staging buffer (a memcpy of kilobytes), go up
// in ONE copy together with a zeroed status word, and status + results come back in ONE copy.
constexpr size_t kSmallBytes = 12u << 20;                   // (~340 000 partitions)
constexpr size_t kGroupStageBytes = 2u << 20;               // la_group_last_by_member: CSRs up to this size cross in one copy
constexpr size_t kMappedSmallBytes = 1280u << 10;          // mapped caller arrays: layouts beyond this are read in place instead
constexpr size_t kMappedSmallGroupedBytes = 3u << 20;      // ... and beyond this when the call wants every member's list too
constexpr size_t kPinnedSmallBytes = 2u << 20;             // pinned, not mapped: beyond this the three-stream pipeline
constexpr size_t kSmallHostCheck = 16384;      // consumer entries up to which the small path validates the ranks on the host

struct SmallLayout {
    size_t po, co, pid, end, com, beg, cr, status, ot, goff, gt, gp, op, orank, total;
};

SmallLayout small_layout(const HostCall& c) {
    const size_t T = (size_t)c.T, n = (size_t)c.shape.n, k = (size_t)c.shape.k;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 16 + 255) & ~(size_t)255; return o; };
    SmallLayout L{};
    L.po = carve((T + 1) * 8);
    L.co = carve((T + 1) * 8);
    L.pid = carve(n * 4);
    L.end = carve(n * 8);                                     // the lags, for la_assign_batch_lags
    L.com = carve(c.lag ? 0 : n * 8);
    L.beg = carve(c.use_begin ? n * 8 : 0);
    L.cr = carve(k * 4);
    L.status = carve(256);                                    // last word of the upload, first of the download
    L.ot = carve(k * 8);
    const bool grouped = c.g_members >= 0;                    // la_assign_batch_grouped: the CSR rides in the same download
    L.goff = carve(grouped ? ((size_t)c.g_members + 1) * 8 : 0);
    L.gt = carve(grouped && c.g_topic ? n * 4 : 0);
    L.gp = carve(grouped ? n * 4 : 0);
    L.op = carve(n * 4);
    L.orank = carve(n * 4);
    L.total = off;
    return L;
}

// ---- the smallest calls: zero copies ----------------------------------------------------------------------------------
// For a rebalance of a few dozen topics even the two DMA submissions of assign_small are most of the call (~8 us each of ~35).
// Here the staging buffer is coherent host memory mapped into the device: the kernels READ the inputs in place over PCIe (all
// loads of a tile go out back to back: a couple of ~1.5 us round trips), write what the caller wants back -- totals, results
// or every member's list -- straight into it, and the call's last launch stores `done | status` where this thread is spinning.
// The ungrouped result of a grouped call never leaves the device.  No hipMemcpy, no stream synchronize.
// Up to which staging layout: rounds 3-4 stopped at 128 KB, where the form's three dependent launches + PCIe reads at kernel
// rate lost to one DMA copy each way.  With the call's end (and the small lists) fused into the tile kernel (round 5) it wins
// at every size measured up to where packing the inputs loses to the lanes' overlapped copies
// (profiles/r05_n_latency_probe.txt, r05_o_, r05_q_: 5 000 partitions 49 -> 35 us, 10 000: 56 -> 38, 30 000: 84 -> 65, 100 000:
// 242 (lanes) -> 160; 256 000: 335 (lanes) vs 402 packed by one thread, 267-310 with the parked threads' help (copy_pieces);
// 512 000: 546 vs 552) -- so every staged call is zero-copy now, and the one-copy form
// (assign_small) is what LA_ZERO_COPY_BYTES=<smaller> still selects (A/B, tests).
constexpr size_t kZeroCopyBytes = 12u << 20;

int reserve_host_coherent(la_ctx* ctx, HostBuf& b, size_t bytes) {
    if (bytes <= b.cap) return LA_OK;
    if (b.p) { LA_HIP(ctx, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    LA_HIP(ctx, hipHostMalloc(&b.p, want, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
    b.cap = want;
    return LA_OK;
}

int small_fill_inputs(la_ctx* ctx, const HostCall& c, const SmallLayout& L, char* h);

// The staged forms pack the caller's arrays into one buffer and unpack the results: plain memcpy.  From a few megabytes on that
// is most of the call on ONE thread (~35 GB/s: 100 us for the inputs of 100 000 partitions), so the pieces are handed out to
// the context's parked threads as well -- the calling thread starts on the first piece at once and simply does all of them
// when nobody else turns up in time (a parked thread needs 30-50 us to be back).  Measured, serial / with help
// (profiles/r05_q_latency_probe.txt): 100 000 partitions 161 / 139-166 us, 256 000: 402 / 267-310 us (lanes: 335).
struct CopyPiece { void* dst; const void* src; size_t bytes; };
constexpr size_t kParallelCopyBytes = 3u << 20, kCopyPieceBytes = 256u << 10;

void copy_pieces(la_ctx* ctx, const CopyPiece* cp, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += cp[i].bytes;
    static const bool serial = getenv("LA_NO_PARALLEL_COPY") != nullptr;                // (A/B hook)
    if (total < kParallelCopyBytes || serial) {
        for (int i = 0; i < n; ++i)
            if (cp[i].bytes) memcpy(cp[i].dst, cp[i].src, cp[i].bytes);
        return;
    }
    std::vector<CopyPiece> pieces;
    pieces.reserve(total / kCopyPieceBytes + (size_t)n);
    for (int i = 0; i < n; ++i)
        for (size_t o = 0; o < cp[i].bytes; o += kCopyPieceBytes)
            pieces.push_back({(char*)cp[i].dst + o, (const char*)cp[i].src + o,
                              cp[i].bytes - o < kCopyPieceBytes ? cp[i].bytes - o : kCopyPieceBytes});
    std::atomic<size_t> next{0};
    ctx->pool.run(total >= (4u << 20) ? 4 : 3, [&](int) {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pieces.size()) break;
            memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes);
        }
    });
}

int assign_small_zc(la_ctx* ctx, const HostCall& c, Shard& sh, const SmallLayout& L) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    Lane& ln = sh.lanes[0];
    hipStream_t st = ln.stream;
    const size_t n = (size_t)c.shape.n, k = (size_t)c.shape.k;
    if (int rc = reserve_host_coherent(ctx, sh.zc_h, L.total)) return rc;
    if (int rc = reserve(ctx, sh.small_d, L.total)) return rc;           // the ungrouped result (same layout, device side)
    char* h = (char*)sh.zc_h.p;
    char* d = (char*)sh.small_d.p;
    void* hd = nullptr;                                                  // the device's view of the staging buffer
    LA_HIP(ctx, hipHostGetDevicePointer(&hd, h, 0));
    char* m = (char*)hd;
    if (int rc = small_fill_inputs(ctx, c, L, h)) return rc;
    volatile uint32_t* flag = (volatile uint32_t*)(h + L.status);
    *flag = 0;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (k > kSmallHostCheck)
        LA_HIP(ctx, la::check_consumers_launch(c.T, (const int64_t*)(m + L.co), (const int32_t*)(m + L.cr), ln.d_status, st));
    const bool grouped = c.g_members >= 0;
    la_device_batch b{};
    b.n_topics = c.T;
    b.reset_mode = c.reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
    b.algo = LA_ALGO_AUTO;
    b.flags = LA_FLAG_RAGGED;
    apply_hints(c, b);
    b.n_partitions = c.shape.n;
    b.n_consumers = c.shape.k;
    b.max_partitions_per_topic = c.shape.max_p;
    b.max_consumers_per_topic = c.shape.max_c;
    b.d_part_off = (const int64_t*)(m + L.po);
    b.d_partition_id = (const int32_t*)(m + L.pid);
    b.d_begin_off = c.use_begin ? (const int64_t*)(m + L.beg) : nullptr;
    b.d_end_off = (const int64_t*)(m + L.end);
    b.d_committed_off = (const int64_t*)(m + L.com);
    b.d_lag = c.lag ? (const int64_t*)(m + L.end) : nullptr;
    b.d_cons_off = (const int64_t*)(m + L.co);
    b.d_cons_rank = (const int32_t*)(m + L.cr);
    // results the caller reads go to the host's memory; those only the grouping reads stay on the device
    char* res = (c.out_pid && !grouped) ? m : d;
    b.d_out_partition = (int32_t*)(res + L.op);
    b.d_out_member_rank = (int32_t*)(res + L.orank);
    b.d_out_total_lag = c.out_total ? (int64_t*)(m + L.ot) : nullptr;
    b.h_part_off = c.part_off;
    b.h_cons_off = c.cons_off;
    // ONE launch for the whole rebalance where the batch is one resident tile launch: its last workgroup builds the lists and
    // stores the completion word -- otherwise the grouping / finishing launches below.  Where that pays was measured at the C
    // ABI (profiles/r05_ae_fused_tail.txt): with lists, 100 / 500 partitions 22-24 / 25 us fused against 25 / 26.5 us with the
    // one-workgroup grouping as its own launch, but 2 000 partitions 37-38 against 34 -- the tail is 256 threads of the tile
    // kernel, the grouping kernel 1 024, and every workgroup of a fused launch pays a system-scope fence -- so the lists are fused
    // up to kTailMaxEntries entries; a call WITHOUT lists is faster with the plain finishing launch at every size (100
    // partitions 16.6 against 18.6 us, 10 000: 31 against 38) and never takes the tail.  LA_FUSED_TAIL=all restores round
    // 5's first form (every staged call that can), LA_NO_FUSED_TAIL=1 never fuses.
    static const bool no_tail = getenv("LA_NO_FUSED_TAIL") != nullptr;                 // (A/B hooks)
    static const bool tail_all = [] { const char* e = getenv("LA_FUSED_TAIL"); return e && !strcmp(e, "all"); }();
    const bool one_wg_lists = grouped ? (c.shape.n <= (tail_all ? la::kSmallGroupN : la::kTailMaxEntries) &&
                                         (int64_t)c.g_members + 2 <= la::kTailGroupM)
                                      : tail_all;
    ln.tail_done = false;
    ln.tail_wanted = one_wg_lists && !no_tail && k <= kSmallHostCheck;
    if (ln.tail_wanted) {
        ln.tail = la::TileTail{};
        ln.tail.enabled = 1;
        ln.tail.n_members = grouped ? c.g_members : 0;
        ln.tail.n = (int32_t)c.shape.n;
        ln.tail.n_topics = c.T;
        ln.tail.part_off = (const int64_t*)(m + L.po);
        ln.tail.out_pid = b.d_out_partition;
        ln.tail.out_rank = b.d_out_member_rank;
        ln.tail.member_off = grouped ? (int64_t*)(m + L.goff) : nullptr;
        ln.tail.grouped_topic = grouped && c.g_topic ? (int32_t*)(m + L.gt) : nullptr;
        ln.tail.grouped_partition = grouped ? (int32_t*)(m + L.gp) : nullptr;
        ln.tail.counter = ln.d_status + 24;
        ln.tail.fin_flag = (uint32_t*)(m + L.status);
    }
    int rc = enqueue_batch(ctx, ln, &b, st);
    ln.tail_wanted = false;
    bool finished = ln.tail_done;                                        // (the one-workgroup grouping also finishes the call)
    ln.tail_done = false;
    if (rc == LA_OK && grouped && !finished) {
        const hipError_t e = la::group_by_member_launch(ln.large, c.shape.n, c.g_members, c.T, (const int64_t*)(m + L.po),
                                                        (const int32_t*)(d + L.op), (const int32_t*)(d + L.orank),
                                                        (int64_t*)(m + L.goff), c.g_topic ? (int32_t*)(m + L.gt) : nullptr,
                                                        (int32_t*)(m + L.gp), nullptr, ln.d_status, st,
                                                        (uint32_t*)(m + L.status), &finished);
        if (e != hipSuccess) rc = fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "group_by_member: %s", hipGetErrorString(e));
    }
    if (rc == LA_OK && !finished) {
        const hipError_t e = la::finish_status_launch(ln.d_status, (uint32_t*)(m + L.status), st);
        if (e != hipSuccess) rc = fail(ctx, LA_EHIP, "finish launch: %s", hipGetErrorString(e));
    }
    if (rc != LA_OK) {
        // nothing of this call stays in flight or behind: kernels already enqueued may have OR-ed bits into the lane's status word
        // (and counted themselves done on the tail's counter) -- the next call must not inherit them
        (void)hipStreamSynchronize(st);
        (void)hipMemsetAsync(ln.d_status, 0, sizeof(uint32_t), st);
        (void)hipMemsetAsync(ln.d_status + 24, 0, sizeof(uint32_t), st);
        (void)hipStreamSynchronize(st);
        return rc;
    }
    // the wait: a spin on the flag word; should the flag never come (a fault on the stream), the runtime will say why
    uint32_t f = 0;
    for (uint64_t spins = 0;; ++spins) {
        f = *flag;
        if (f & 0x80000000u) break;
        if ((spins & 0xFFFu) == 0xFFFu) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) { f = *flag; if (f & 0x80000000u) break; LA_HIP(ctx, hipStreamSynchronize(st)); f = *flag | 0x80000000u; break; }
            if (q != hipErrorNotReady) {
                (void)hipMemsetAsync(ln.d_status, 0, sizeof(uint32_t), st);          // (best effort after a stream error)
                (void)hipMemsetAsync(ln.d_status + 24, 0, sizeof(uint32_t), st);
                return fail(ctx, LA_EHIP, "zero-copy call: %s", hipGetErrorString(q));
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const uint32_t status = f & 0x7FFFFFFFu;
    if (status) {
        LA_HIP(ctx, hipMemsetAsync(ln.d_status, 0, sizeof(uint32_t), st));
        LA_HIP(ctx, hipStreamSynchronize(st));
        return status_error(ctx, status);
    }
    if (c.out_total && k) memcpy(c.out_total, h + L.ot, k * 8);
    if (c.out_pid && n) {
        if (grouped) {
            // (no caller asks for both forms today: the ungrouped arrays would need their own trip)
            LA_HIP(ctx, hipMemcpyAsync(c.out_pid, d + L.op, n * 4, hipMemcpyDeviceToHost, st));
            LA_HIP(ctx, hipMemcpyAsync(c.out_rank, d + L.orank, n * 4, hipMemcpyDeviceToHost, st));
            LA_HIP(ctx, hipStreamSynchronize(st));
        } else {
            const CopyPiece out[2] = {{c.out_pid, h + L.op, n * 4}, {c.out_rank, h + L.orank, n * 4}};
            copy_pieces(ctx, out, 2);
        }
    }
    if (grouped) {
        memcpy(c.g_off, h + L.goff, ((size_t)c.g_members + 1) * 8);
        if (n) {
            const CopyPiece out[2] = {{c.g_part, h + L.gp, n * 4}, {c.g_topic, h + L.gt, c.g_topic ? n * 4 : 0}};
            copy_pieces(ctx, out, 2);
        }
        if (c.grouped_done) *c.grouped_done = true;
    }
    // what la_group_last_by_member would read: valid only where the results stayed on the device
    sh.last_part_off = (const int64_t*)(m + L.po);
    sh.last_out_pid = (const int32_t*)(res + L.op);
    sh.last_out_rank = (const int32_t*)(res + L.orank);
    return LA_OK;
}

// The inputs of a small call, packed into its staging buffer (and validated where that is cheaper here than on the device).
int small_fill_inputs(la_ctx* ctx, const HostCall& c, const SmallLayout& L, char* h) {
    const size_t T = (size_t)c.T, n = (size_t)c.shape.n, k = (size_t)c.shape.k;
    memcpy(h + L.po, c.part_off, (T + 1) * 8);
    memcpy(h + L.co, c.cons_off, (T + 1) * 8);
    if (n) {
        const CopyPiece in[4] = {{h + L.pid, c.pid, n * 4},
                                 {h + L.end, c.lag ? c.lag : c.end, n * 8},
                                 {h + L.com, c.committed, c.lag ? 0 : n * 8},
                                 {h + L.beg, c.begin, (c.use_begin && !c.sparse) ? n * 8 : 0}};
        copy_pieces(ctx, in, 4);
        if (c.sparse) {
            // a small call rebuilds the dense array right here in the staging buffer: unlisted partitions have begin 0
            int64_t* hb = (int64_t*)(h + L.beg);
            memset(hb, 0, n * 8);
            int64_t prev = -1;
            for (int64_t j = 0; j < c.n_none; ++j) {
                const int64_t g = c.none_index[j];
                if (g <= prev || g >= (int64_t)n)
                    return fail(ctx, LA_EINVAL, "none_index must hold ascending positions inside the batch");
                hb[g] = c.none_begin[j];
                prev = g;
            }
        }
    }
    if (k) memcpy(h + L.cr, c.cons_rank, k * 4);
    if (k && k <= kSmallHostCheck) {
        // a topic's member ranks must be strictly ascending (the kernels compare ranks, never strings).  A few thousand
        // entries are checked here in less time than the ~5 us a dependent launch of the check kernel puts in front of
        // the assignment kernels -- and a real rebalance is this size.
        for (size_t t = 0; t < T; ++t)
            for (int64_t j = c.cons_off[t] + 1; j < c.cons_off[t + 1]; ++j)
                if (c.cons_rank[j - 1] >= c.cons_rank[j])
                    return fail(ctx, LA_EINVAL, "a topic's cons_rank segment is not strictly ascending");
    }
    return LA_OK;
}

int assign_small(la_ctx* ctx, const HostCall& c, Shard& sh, const SmallLayout& L) {
    LA_HIP(ctx, hipSetDevice(sh.device));
    Lane& ln = sh.lanes[0];
    hipStream_t st = ln.stream;
    const size_t n = (size_t)c.shape.n, k = (size_t)c.shape.k;
    if (int rc = reserve(ctx, sh.small_d, L.total)) return rc;
    if (int rc = reserve_host(ctx, sh.small_h, L.total)) return rc;
    char* h = (char*)sh.small_h.p;
    char* d = (char*)sh.small_d.p;
    if (int rc = small_fill_inputs(ctx, c, L, h)) return rc;
    memset(h + L.status, 0, 256);
    LA_HIP(ctx, hipMemcpyAsync(d, h, L.status + 256, hipMemcpyHostToDevice, st));
    ln.status_word = (uint32_t*)(d + L.status);
    struct Restore { Lane& l; ~Restore() { l.status_word = nullptr; } } restore{ln};
    if (k > kSmallHostCheck) {
        LA_HIP(ctx, la::check_consumers_launch(c.T, (const int64_t*)(d + L.co), (const int32_t*)(d + L.cr),
                                               ln.status_word, st));
    }
    la_device_batch b{};
    b.n_topics = c.T;
    b.reset_mode = c.reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
    b.algo = LA_ALGO_AUTO;
    b.flags = LA_FLAG_RAGGED;
    apply_hints(c, b);
    b.n_partitions = c.shape.n;
    b.n_consumers = c.shape.k;
    b.max_partitions_per_topic = c.shape.max_p;
    b.max_consumers_per_topic = c.shape.max_c;
    b.d_part_off = (const int64_t*)(d + L.po);
    b.d_partition_id = (const int32_t*)(d + L.pid);
    b.d_begin_off = c.use_begin ? (const int64_t*)(d + L.beg) : nullptr;
    b.d_end_off = (const int64_t*)(d + L.end);
    b.d_committed_off = (const int64_t*)(d + L.com);
    b.d_lag = c.lag ? (const int64_t*)(d + L.end) : nullptr;
    b.d_cons_off = (const int64_t*)(d + L.co);
    b.d_cons_rank = (const int32_t*)(d + L.cr);
    b.d_out_partition = (int32_t*)(d + L.op);
    b.d_out_member_rank = (int32_t*)(d + L.orank);
    b.d_out_total_lag = (int64_t*)(d + L.ot);                 // always: it sits between the status and the results
    b.h_part_off = c.part_off;
    b.h_cons_off = c.cons_off;
    if (int rc = enqueue_batch(ctx, ln, &b, st)) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    if (c.g_members >= 0) {
        // every member's list, built where the results are: one more launch on the same stream, no second round trip
        const hipError_t e = la::group_by_member_launch(ln.large, c.shape.n, c.g_members, c.T, (const int64_t*)(d + L.po),
                                                        (const int32_t*)(d + L.op), (const int32_t*)(d + L.orank),
                                                        (int64_t*)(d + L.goff), c.g_topic ? (int32_t*)(d + L.gt) : nullptr,
                                                        (int32_t*)(d + L.gp), nullptr, ln.status_word, st);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(st);
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "group_by_member: %s", hipGetErrorString(e));
        }
    }
    // status | totals | (member_off | grouped topic | grouped partition) | partition order | member ranks: as far as the caller wants them
    const size_t upto = c.out_pid ? L.total : L.op;
    LA_HIP(ctx, hipMemcpyAsync(h + L.status, d + L.status, upto - L.status, hipMemcpyDeviceToHost, st));
    LA_HIP(ctx, hipStreamSynchronize(st));
    const uint32_t status = *(const uint32_t*)(h + L.status);
    if (status) return status_error(ctx, status);
    if (c.out_total && k) memcpy(c.out_total, h + L.ot, k * 8);
    if (c.out_pid && n) {
        memcpy(c.out_pid, h + L.op, n * 4);
        memcpy(c.out_rank, h + L.orank, n * 4);
    }
    if (c.g_members >= 0) {
        memcpy(c.g_off, h + L.goff, ((size_t)c.g_members + 1) * 8);
        if (n) {
            memcpy(c.g_part, h + L.gp, n * 4);
            if (c.g_topic) memcpy(c.g_topic, h + L.gt, n * 4);
        }
        if (c.grouped_done) *c.grouped_done = true;
    }
    sh.last_part_off = (const int64_t*)(d + L.po);
    sh.last_out_pid = (const int32_t*)(d + L.op);
    sh.last_out_rank = (const int32_t*)(d + L.orank);
    return LA_OK;
}

// la_hint_next_call is one-shot: "forgotten when the next host-buffer assign call returns, whatever it returns".  assign_host
// takes the hints when it starts; every host-buffer entry point also holds one of these, so that a call that returns BEFORE
// it gets there (a NULL result array, n_topics <= 0, a NULL lag array) spends them too and a stale bound cannot fail the
// next, unrelated call.
struct HintSpender {
    la_ctx* ctx;
    explicit HintSpender(la_ctx* c) : ctx(c) {}
    ~HintSpender() { if (ctx) ctx->hints_set = false; }
    HintSpender(const HintSpender&) = delete;
    HintSpender& operator=(const HintSpender&) = delete;
};

int assign_host(la_ctx* ctx, int32_t T, const int64_t* part_off, const int32_t* pid, const int64_t* begin,
                const int64_t* end, const int64_t* committed, const int64_t* lag, int32_t reset_mode,
                const int64_t* cons_off, const int32_t* cons_rank, int32_t* out_pid, int32_t* out_rank,
                int64_t* out_total, const HostCall* grouped = nullptr, const HostCall* sparse = nullptr) {
    if (!ctx) return LA_EINVAL;
    ctx->last_valid = false;
    const bool hinted = ctx->hints_set;                       // one-shot: whatever this call returns, the hints are spent
    ctx->hints_set = false;
    if (T < 0) return fail(ctx, LA_EINVAL, "n_topics < 0");
    if (T == 0) return LA_OK;
    if (!part_off || !cons_off) return fail(ctx, LA_EINVAL, "null offsets");
    HostCall c;
    if (hinted && (ctx->hints.flags & LA_HINT_BOUNDS) && ctx->hints.max_lag >= 0 && ctx->hints.max_partition_id >= 0) {
        c.bounded = true;
        c.max_lag = ctx->hints.max_lag;
        c.max_id = ctx->hints.max_partition_id;
    }
    // offsets are checked here; cons_rank's order on the device (run_lane)
    if (int rc = scan_shape(ctx, T, part_off, cons_off, nullptr, &c.shape)) return rc;
    const Shape& s = c.shape;
    if (s.n > 0 && (!pid || (!lag && (!end || !committed)))) return fail(ctx, LA_EINVAL, "null per-partition buffer");
    if ((out_pid == nullptr) != (out_rank == nullptr))
        return fail(ctx, LA_EINVAL, "out_partition and out_member_rank must both be given or both be NULL");
    if (s.k > 0 && !cons_rank) return fail(ctx, LA_EINVAL, "null cons_rank");
    const bool is_sparse = sparse && !lag && reset_mode != LA_RESET_LATEST;      // (`latest` never reads begin: the list is ignored)
    if (is_sparse && (sparse->n_none < 0 || (sparse->n_none > 0 && (!sparse->none_index || !sparse->none_begin))))
        return fail(ctx, LA_EINVAL, "n_none < 0 or a null sparse-begin array");
    if (!lag && reset_mode != LA_RESET_LATEST && !begin && !is_sparse && s.n > 0)
        return fail(ctx, LA_EINVAL, "begin_off is required unless reset_mode is LA_RESET_LATEST");
    c.T = T; c.part_off = part_off; c.pid = pid; c.begin = begin; c.end = end; c.committed = committed; c.lag = lag;
    c.reset_mode = reset_mode; c.cons_off = cons_off; c.cons_rank = cons_rank;
    c.out_pid = out_pid; c.out_rank = out_rank; c.out_total = out_total;
    c.use_begin = !lag && (begin || is_sparse) && reset_mode != LA_RESET_LATEST;
    if (is_sparse) {
        c.sparse = true; c.begin = nullptr;
        c.n_none = sparse->n_none; c.none_index = sparse->none_index; c.none_begin = sparse->none_begin;
    }
    if (grouped) {
        c.g_members = grouped->g_members; c.g_off = grouped->g_off; c.g_topic = grouped->g_topic; c.g_part = grouped->g_part;
        c.grouped_done = grouped->grouped_done;
    }

    // shards: all devices for a batch worth splitting, fewer (down to one) for a small one
    int S = (int)ctx->shards.size();
    if (!ctx->split_always) {
        const int64_t by_size = s.n / kMinShardPartitions;
        if (by_size < S) S = by_size < 1 ? 1 : (int)by_size;
    }
    if (S > T) S = T;
    plan_ranges(part_off, 0, T, S, ctx->last_bounds);
    ctx->last_shards = S;
    if (S == 1 && !ctx->split_always && s.n > 0) {
        SmallLayout L = small_layout(c);
        // How large a layout still travels through ONE staging buffer.  Pageable caller arrays: ctx->small_bytes (12 MB) -- beyond,
        // packing the inputs costs as much as the lanes' overlapped copies.  Caller arrays that are pinned
        // need no packing at all on the other side of the comparison: device-mapped ones (la_host_alloc: the Java host's direct
        // buffers) are read in place by the kernels from 1.25 MB on (3 MB with the lists aboard, which the staged form brings
        // back in the same round trip); pinned but not mapped ones keep round 4's 2 MB.  profiles/r05_o_latency_probe.txt.
        const bool no_pinned_rule = getenv("LA_NO_MAPPED_SMALL") || getenv("LA_NO_ASYNC_PIPELINE");
        auto staged_upto = [&](bool with_lists) {
            size_t upto = ctx->small_bytes;
            if (L.total > kMappedSmallBytes && !no_pinned_rule && call_is_pinned(c)) {
                const bool mapped = !getenv("LA_NO_MAPPED_PIPELINE") && call_is_mapped(c);
                const size_t lim = mapped ? (with_lists ? kMappedSmallGroupedBytes : kMappedSmallBytes) : kPinnedSmallBytes;
                if (lim < upto) upto = lim;
            }
            return upto;
        };
        size_t upto = staged_upto(c.g_members >= 0);
        if (L.total > upto && c.g_members >= 0) {            // too large with the lists aboard: they take their own round trip
            c.g_members = -1;
            L = small_layout(c);
            upto = staged_upto(false);
        }
        if (L.total <= upto) {
            Shard& sh = ctx->shards[0];
            sh.last_t0 = 0; sh.last_topics = T; sh.last_p0 = 0; sh.last_n = s.n;
            const bool zc = L.total <= ctx->zero_copy_bytes;
            ctx->last_pipeline = zc ? LA_PIPELINE_ZERO_COPY : LA_PIPELINE_ONE_COPY;
            if (int rc = zc ? assign_small_zc(ctx, c, sh, L) : assign_small(ctx, c, sh, L)) return rc;
            ctx->last_valid = true;
            return LA_OK;
        }
    }
    std::vector<ShardPlan> plans((size_t)S);
    struct Work { int shard, lane; };
    std::vector<Work> work;
    const bool pinned = call_is_pinned(c) && !getenv("LA_NO_ASYNC_PIPELINE");
    c.mapped = pinned && !getenv("LA_NO_MAPPED_PIPELINE") && call_is_mapped(c);
    ctx->last_pipeline = c.mapped ? LA_PIPELINE_MAPPED : (pinned ? 2 : 1);
    for (int i = 0; i < S; ++i) {
        ShardPlan& sp = plans[(size_t)i];
        Shard& sh = ctx->shards[(size_t)i];
        sp.t0 = ctx->last_bounds[i];
        sp.t1 = ctx->last_bounds[i + 1];
        sp.P0 = part_off[sp.t0]; sp.n = part_off[sp.t1] - sp.P0;
        sp.K0 = cons_off[sp.t0]; sp.k = cons_off[sp.t1] - sp.K0;
        sh.last_t0 = sp.t0; sh.last_topics = sp.t1 - sp.t0; sh.last_p0 = sp.P0; sh.last_n = sp.n;
        if (sp.t1 == sp.t0) continue;
        const int rc_prepare = prepare_shard(ctx, c, sh, sp);
        sh.last_part_off = (const int64_t*)sh.part_off.p;       // (after the reserves of prepare_shard)
        sh.last_out_pid = (const int32_t*)sh.out_pid.p;
        sh.last_out_rank = (const int32_t*)sh.out_rank.p;
        if (int rc = rc_prepare) {
            for (int j = 0; j <= i; ++j)                         // nothing of this call stays in flight
                if (hipSetDevice(ctx->shards[(size_t)j].device) == hipSuccess)
                    (void)hipStreamSynchronize(ctx->shards[(size_t)j].lanes[0].stream);
            return rc;
        }
        if (c.mapped && c.out_pid) {
            // the kernels wrote the ungrouped results into the caller's own (mapped) arrays: that is where a following
            // la_group_last_by_member reads them, in place
            sh.last_out_pid = mapped_ptr(c.out_pid) + sp.P0;
            sh.last_out_rank = mapped_ptr(c.out_rank) + sp.P0;
        }
        const int n_chunks = (int)sp.chunk.size() - 1;
        const int lanes = n_chunks < (int)sh.lanes.size() ? n_chunks : (int)sh.lanes.size();
        if (!pinned)
            for (int l = 0; l < lanes; ++l) work.push_back({i, l});
    }
    if (pinned) {
        // pinned caller arrays: this thread enqueues every shard's chunks (nothing blocks), then waits shard by shard
        int rc = LA_OK;
        for (int i = 0; i < S && rc == LA_OK; ++i)
            if (plans[(size_t)i].t1 > plans[(size_t)i].t0)
                rc = c.mapped ? run_shard_mapped(ctx, c, ctx->shards[(size_t)i], plans[(size_t)i])
                              : run_shard_async(ctx, c, ctx->shards[(size_t)i], plans[(size_t)i]);
        for (int i = 0; i < S; ++i) {
            if (plans[(size_t)i].t1 == plans[(size_t)i].t0) continue;
            const int rf = finish_shard_async(ctx, ctx->shards[(size_t)i]);
            if (rc == LA_OK) rc = rf;
        }
        if (rc) return rc;
        ctx->last_valid = true;
        return LA_OK;
    }
    std::atomic<bool> stop{false};
    const int rc = run_workers(ctx, (int)work.size(), [&](int w) {
        const Work& wk = work[(size_t)w];
        const int r = run_lane(ctx, c, ctx->shards[(size_t)wk.shard], plans[(size_t)wk.shard], wk.lane, stop);
        if (r != LA_OK) stop.store(true, std::memory_order_relaxed);
        return r;
    });
    if (rc) return rc;
    ctx->last_valid = true;
    return LA_OK;
}

// ---- RCCL, loaded at run time (la_allgather_results) ------------------------------------------------------------------
// Only the five entry points the all-gather needs; prototypes as in rccl.h (ncclResult_t is an int enum, 0 = success;
// ncclInt32 = 2).  A copy of the library that the process has already loaded wins: a process that carries its own ROCm
// runtime (PyTorch does) must not get a second one through /opt/rocm's librccl.
struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t stream) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);       // already in the process?
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!api.handle) { api.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char* name) { return dlsym(api.handle, name); };
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!api.CommInitAll || !api.CommDestroy || !api.AllGather || !api.GroupStart || !api.GroupEnd) {
            api.why = "librccl lacks ncclCommInitAll / ncclAllGather / ncclGroupStart";
            api.handle = nullptr;
        }
    });
    return api;
}

int create_lane(Lane& ln) {
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void**)&ln.d_status, 256)) != hipSuccess ||
        // zeroed ON the lane's stream and waited for: the stream is non-blocking, so a null-stream hipMemset is not ordered
        // before the first kernel launched on it -- a context whose very first call came right behind la_create could find
        // the status word or the fused tail's workgroup counter stale (a small grouped call then came back with empty lists;
        // found by la_wake's test in round 6: hipMalloc hands back memory a destroyed context had used)
        (e = hipMemsetAsync(ln.d_status, 0, 256, ln.stream)) != hipSuccess ||
        (e = hipStreamSynchronize(ln.stream)) != hipSuccess ||
        (e = hipHostMalloc((void**)&ln.h_status, 64, hipHostMallocDefault)) != hipSuccess)
        return fail(nullptr, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "context setup: %s", hipGetErrorString(e));
    *ln.h_status = 0;
    return LA_OK;
}

void destroy_lane(Lane& ln) {
    if (ln.stream) (void)hipStreamSynchronize(ln.stream);
    release(ln.defer);
    release(ln.block_list);
    for (Lane::Stage& sg : ln.stage) {
        if (sg.done) { (void)hipEventSynchronize(sg.done); (void)hipEventDestroy(sg.done); }
        if (sg.p) (void)hipHostFree(sg.p);
    }
    la::large_scratch_release(ln.large);
    if (ln.d_status) (void)hipFree(ln.d_status);
    if (ln.h_status) (void)hipHostFree(ln.h_status);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
}

int reserve_host(la_ctx* ctx, HostBuf& b, size_t bytes) {
    if (bytes <= b.cap) return LA_OK;
    if (b.p) { LA_HIP(ctx, hipHostFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 4 + 256;
    LA_HIP(ctx, hipHostMalloc(&b.p, want, hipHostMallocPortable));
    b.cap = want;
    return LA_OK;
}

// grouping of one shard's last results, on its lane 0, into the given device arrays (null: the shard's own pid =
// grouped partition, cons_rank = grouped topic with shard-local indices, out_total = member_off -- their old contents
// are no longer needed)
int group_shard_device(la_ctx* ctx, Shard& sh, int32_t n_members, bool want_topic, int64_t* d_off = nullptr,
                       int32_t* d_topic = nullptr, int32_t* d_part = nullptr) {
    const int64_t n = sh.last_n;
    if (n > 0x7FFFFFFF) return fail(ctx, LA_ESHAPE, "at most 2^31-1 entries per shard are supported");
    if (!d_off) {
        const size_t nb4 = (size_t)n * 4, mb = ((size_t)n_members + 1) * 8;
        int rc;
        if ((rc = reserve(ctx, sh.pid, nb4 + 16)) || (rc = reserve(ctx, sh.cons_rank, nb4 + 16)) ||
            (rc = reserve(ctx, sh.out_total, mb + 16)))
            return rc;
        d_off = (int64_t*)sh.out_total.p;
        d_topic = (int32_t*)sh.cons_rank.p;
        d_part = (int32_t*)sh.pid.p;
    }
    hipError_t e = la::group_by_member_launch(sh.lanes[0].large, n, n_members, sh.last_topics, sh.last_part_off,
                                              sh.last_out_pid, sh.last_out_rank, d_off, want_topic ? d_topic : nullptr,
                                              d_part, nullptr, sh.lanes[0].d_status, sh.lanes[0].stream);
    if (e != hipSuccess)
        return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "group_by_member: %s", hipGetErrorString(e));
    return LA_OK;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------
LA_API int la_version(void) { return LA_VERSION; }

LA_API const char* la_last_error(const la_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

LA_API int la_device_count(void) {
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) return fail(nullptr, LA_ENODEV, "no HIP device (%s)", hipGetErrorString(e));
    return count;
}

LA_API int la_create_multi(la_ctx** out, int n_devices, const int* device_ids, unsigned flags) {
    DeviceGuard restore_device;
    if (!out) return fail(nullptr, LA_EINVAL, "out is NULL");
    *out = nullptr;
    try {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            return fail(nullptr, LA_ENODEV, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        std::vector<int> ids;
        if (n_devices < 0) return fail(nullptr, LA_EINVAL, "n_devices < 0");
        if (n_devices == 0 || !device_ids) {
            if (n_devices != 0) return fail(nullptr, LA_EINVAL, "device_ids is NULL");
            for (int d = 0; d < count; ++d) ids.push_back(d);          // every device of the node
        } else {
            ids.assign(device_ids, device_ids + n_devices);
        }
        if ((int)ids.size() > kMaxShards) return fail(nullptr, LA_EINVAL, "at most %d shards", kMaxShards);
        for (int d : ids) {
            if (d < 0 || d >= count) return fail(nullptr, LA_ENODEV, "device %d of %d", d, count);
            hipDeviceProp_t prop;
            if ((e = hipGetDeviceProperties(&prop, d)) != hipSuccess)
                return fail(nullptr, LA_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
            if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
                return fail(nullptr, LA_ENODEV, "device %d is %s; this library is built for gfx950 only", d,
                            prop.gcnArchName);
        }
        int lanes = (int)(flags & LA_CREATE_LANES_MASK);
        if (const char* env = getenv("LA_LANES")) lanes = atoi(env);
        if (lanes <= 0) lanes = ids.size() <= 2 ? 3 : 2;
        if (lanes > 8) lanes = 8;
        la_ctx* ctx = new (std::nothrow) la_ctx();
        if (!ctx) return fail(nullptr, LA_ENOMEM, "out of host memory");
        ctx->split_always = (flags & LA_CREATE_SPLIT_ALWAYS) != 0;
        if (const char* env = getenv("LA_CHUNK_PARTITIONS")) ctx->chunk_partitions = atoll(env);
        ctx->zero_copy_bytes = kZeroCopyBytes;
        if (const char* env = getenv("LA_ZERO_COPY_BYTES")) ctx->zero_copy_bytes = (size_t)atoll(env);
        ctx->small_bytes = kSmallBytes;
        if (const char* env = getenv("LA_SMALL_BYTES")) ctx->small_bytes = (size_t)atoll(env);      // (lab: where the forms hand over)
        ctx->shards.resize(ids.size());
        for (size_t i = 0; i < ids.size(); ++i) {
            Shard& sh = ctx->shards[i];
            sh.device = ids[i];
            sh.lanes.resize((size_t)lanes);
            int rc = LA_OK;
            if ((e = hipSetDevice(sh.device)) != hipSuccess ||
                (e = hipEventCreateWithFlags(&sh.ready, hipEventDisableTiming)) != hipSuccess)
                rc = fail(nullptr, LA_EHIP, "context setup: %s", hipGetErrorString(e));
            if (rc == LA_OK && (e = la::large_init_device()) != hipSuccess)
                rc = fail(nullptr, LA_EHIP, "context setup (device check): %s", hipGetErrorString(e));
            for (Lane& ln : sh.lanes)
                if (rc == LA_OK) rc = create_lane(ln);
            if (rc != LA_OK) {
                la_destroy(ctx);
                return rc;
            }
        }
        *out = ctx;
        return LA_OK;
    } catch (...) {
        return fail(nullptr, LA_ENOMEM, "exception in la_create_multi");
    }
}

LA_API int la_create(la_ctx** out, int device_id, unsigned flags) { return la_create_multi(out, 1, &device_id, flags); }

LA_API void la_destroy(la_ctx* ctx) {
    DeviceGuard restore_device;
    if (!ctx) return;
    if (!ctx->comms.empty()) {
        for (Shard& sh : ctx->shards) {                                   // nothing of an all-gather stays in flight
            (void)hipSetDevice(sh.device);
            if (!sh.lanes.empty() && sh.lanes[0].stream) (void)hipStreamSynchronize(sh.lanes[0].stream);
        }
        RcclApi& r = rccl_api();
        for (void* c : ctx->comms)
            if (c && r.CommDestroy) (void)r.CommDestroy(c);
        ctx->comms.clear();
    }
    for (Shard& sh : ctx->shards) {
        (void)hipSetDevice(sh.device);
        for (Lane& ln : sh.lanes) destroy_lane(ln);
        for (DevBuf* b : {&sh.part_off, &sh.pid, &sh.begin, &sh.end, &sh.committed, &sh.cons_off, &sh.cons_rank,
                          &sh.out_pid, &sh.out_rank, &sh.out_total, &sh.small_d, &sh.small_g, &sh.none_idx, &sh.none_val})
            release(*b);
        for (HostBuf* h : {&sh.g_off, &sh.g_topic, &sh.g_part, &sh.small_h, &sh.small_gh, &sh.zc_h})
            if (h->p) (void)hipHostFree(h->p);
        if (sh.ready) (void)hipEventDestroy(sh.ready);
        for (hipEvent_t e : sh.chunk_ev) (void)hipEventDestroy(e);
        for (hipStream_t st : sh.copy_in)
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        if (sh.copy_out) { (void)hipStreamSynchronize(sh.copy_out); (void)hipStreamDestroy(sh.copy_out); }
    }
    delete ctx;
}

LA_API int la_shard_count(const la_ctx* ctx) { return ctx ? (int)ctx->shards.size() : LA_EINVAL; }

LA_API int la_shard_device(const la_ctx* ctx, int shard) {
    if (!ctx || shard < 0 || shard >= (int)ctx->shards.size()) return LA_EINVAL;
    return ctx->shards[(size_t)shard].device;
}

LA_API int la_device_features(const la_ctx* ctx, int shard) {
    DeviceGuard restore_device;
    if (!ctx || shard < 0 || shard >= (int)ctx->shards.size()) return LA_EINVAL;
    if (hipSetDevice(ctx->shards[(size_t)shard].device) != hipSuccess) return LA_EHIP;
    return la::large_atomic_rank_supported() ? LA_FEATURE_ATOMIC_RANK : 0;
}

LA_API int la_plan_shards(int32_t n_topics, const int64_t* part_off, int32_t n_shards, int32_t* bounds) {
    if (n_topics < 0 || n_shards < 1 || !part_off || !bounds) return LA_EINVAL;
    for (int32_t t = 0; t < n_topics; ++t)
        if (part_off[t + 1] < part_off[t]) return LA_EINVAL;
    try {
        std::vector<int32_t> tmp((size_t)n_shards + 1);
        plan_ranges(part_off, 0, n_topics, n_shards, tmp.data());
        memcpy(bounds, tmp.data(), tmp.size() * sizeof(int32_t));
        return LA_OK;
    } catch (...) {
        return LA_ENOMEM;
    }
}

LA_API int la_last_shard_bounds(const la_ctx* ctx, int32_t* bounds, int32_t capacity) {
    if (!ctx) return LA_EINVAL;
    const int S = ctx->last_shards;
    if (bounds)
        for (int i = 0; i <= S && i < capacity; ++i) bounds[i] = ctx->last_bounds[i];
    return S;
}

// One ncclAllGather per shard inside one group, `bytes` bytes each (ncclUint8: an all-gather does no arithmetic, the type only
// sizes the elements), on the shards' own streams.
static int allgather_bytes(la_ctx* ctx, size_t bytes, const void* const* d_send, void* const* d_recv) {
    const int S = (int)ctx->shards.size();
    if (!d_send || !d_recv) return fail(ctx, LA_EINVAL, "null buffer list");
    for (int i = 0; i < S; ++i)
        if (bytes > 0 && (!d_send[i] || !d_recv[i])) return fail(ctx, LA_EINVAL, "null buffer of shard %d", i);
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < i; ++j)
            if (ctx->shards[(size_t)i].device == ctx->shards[(size_t)j].device)
                return fail(ctx, LA_EINVAL, "shards %d and %d share device %d: RCCL needs one distinct device per rank", j, i,
                            ctx->shards[(size_t)i].device);
    if (bytes == 0) return LA_OK;
    RcclApi& r = rccl_api();
    if (!r.handle) return fail(ctx, LA_ENODEV, "%s", r.why.c_str());
    auto text = [&](int rc) { return r.GetErrorString ? r.GetErrorString(rc) : "rccl error"; };
    if (ctx->comms.empty()) {
        std::vector<int> devs;
        for (const Shard& sh : ctx->shards) devs.push_back(sh.device);
        std::vector<void*> comms((size_t)S, nullptr);
        const int rc = r.CommInitAll(comms.data(), S, devs.data());
        if (rc != 0) return fail(ctx, LA_EHIP, "ncclCommInitAll over %d device(s): %s", S, text(rc));
        ctx->comms = comms;
    }
    int rc = r.GroupStart();
    if (rc != 0) return fail(ctx, LA_EHIP, "ncclGroupStart: %s", text(rc));
    int first_bad = 0;
    for (int i = 0; i < S; ++i) {
        Shard& sh = ctx->shards[(size_t)i];
        if (hipSetDevice(sh.device) != hipSuccess) { first_bad = -1; break; }
        rc = r.AllGather(d_send[i], d_recv[i], bytes, /* ncclUint8 */ 1, ctx->comms[(size_t)i], sh.lanes[0].stream);
        if (rc != 0 && first_bad == 0) first_bad = rc;
    }
    rc = r.GroupEnd();
    if (first_bad == -1) return fail(ctx, LA_EHIP, "hipSetDevice failed while enqueueing the all-gather");
    if (first_bad != 0) return fail(ctx, LA_EHIP, "ncclAllGather: %s", text(first_bad));
    if (rc != 0) return fail(ctx, LA_EHIP, "ncclGroupEnd: %s", text(rc));
    return LA_OK;
}

LA_API int la_allgather_results(la_ctx* ctx, int64_t count, const int32_t* const* d_send, int32_t* const* d_recv) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (count < 0) return fail(ctx, LA_EINVAL, "negative count");
        return allgather_bytes(ctx, (size_t)count * 4, (const void* const*)d_send, (void* const*)d_recv);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_allgather_results");
    }
}

LA_API int la_allgather_packed(la_ctx* ctx, int64_t count, int32_t elem_bytes, const void* const* d_send, void* const* d_recv) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (count < 0) return fail(ctx, LA_EINVAL, "negative count");
        if (elem_bytes != 2 && elem_bytes != 4 && elem_bytes != 8) return fail(ctx, LA_EINVAL, "elem_bytes must be 2, 4 or 8");
        return allgather_bytes(ctx, (size_t)count * (size_t)elem_bytes, d_send, d_recv);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_allgather_packed");
    }
}

LA_API int la_wire_format_for(int64_t max_partition_id, int64_t n_members, la_wire_format* out) {
    if (!out) return LA_EINVAL;
    int eb = 8, ib = 32;
    la::wire_format_for(max_partition_id, n_members, &eb, &ib);
    out->elem_bytes = eb;
    out->id_bits = ib;
    return LA_OK;
}

LA_API int la_pack_results_on(la_ctx* ctx, int shard, int64_t n, const int32_t* d_out_partition, const int32_t* d_out_member_rank,
                              const la_wire_format* fmt, void* d_packed, void* stream) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (shard < 0 || shard >= (int)ctx->shards.size()) return fail(ctx, LA_EINVAL, "shard %d of %d", shard, (int)ctx->shards.size());
        if (n < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!fmt || !la::wire_format_valid(fmt->elem_bytes, fmt->id_bits)) return fail(ctx, LA_EINVAL, "bad wire format");
        if (n > 0 && (!d_out_partition || !d_out_member_rank || !d_packed)) return fail(ctx, LA_EINVAL, "null buffer");
        Shard& sh = ctx->shards[(size_t)shard];
        LA_HIP(ctx, hipSetDevice(sh.device));
        LA_HIP(ctx, la::wire_pack_launch(n, d_out_partition, d_out_member_rank, fmt->elem_bytes, fmt->id_bits, d_packed,
                                         sh.lanes[0].d_status, (hipStream_t)stream));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_pack_results_on");
    }
}

LA_API int la_unpack_results_on(la_ctx* ctx, int shard, int64_t n, const void* d_packed, const la_wire_format* fmt,
                                int32_t* d_out_partition, int32_t* d_out_member_rank, void* stream) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (shard < 0 || shard >= (int)ctx->shards.size()) return fail(ctx, LA_EINVAL, "shard %d of %d", shard, (int)ctx->shards.size());
        if (n < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!fmt || !la::wire_format_valid(fmt->elem_bytes, fmt->id_bits)) return fail(ctx, LA_EINVAL, "bad wire format");
        if (n > 0 && (!d_out_partition || !d_out_member_rank || !d_packed)) return fail(ctx, LA_EINVAL, "null buffer");
        Shard& sh = ctx->shards[(size_t)shard];
        LA_HIP(ctx, hipSetDevice(sh.device));
        LA_HIP(ctx, la::wire_unpack_launch(n, d_packed, fmt->elem_bytes, fmt->id_bits, d_out_partition, d_out_member_rank,
                                           (hipStream_t)stream));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_unpack_results_on");
    }
}

LA_API int la_last_pipeline(const la_ctx* ctx) { return ctx ? ctx->last_pipeline : LA_EINVAL; }

LA_API int64_t la_last_launches(const la_ctx* ctx) { return ctx ? ctx->last_launches : (int64_t)LA_EINVAL; }

LA_API int la_hint_next_call(la_ctx* ctx, const la_call_hints* hints) {
    if (!ctx) return LA_EINVAL;
    ctx->hints_set = false;
    if (!hints) return LA_OK;
    // the struct may grow: take what the caller's header knows of it (the four fields of ABI 0.4.0 at least)
    if (hints->struct_size < (int32_t)sizeof(la_call_hints))
        return fail(ctx, LA_EINVAL, "la_call_hints.struct_size %d is smaller than the %zu bytes of ABI 0.4.0", hints->struct_size,
                    sizeof(la_call_hints));
    if ((hints->flags & LA_HINT_BOUNDS) && (hints->max_lag < 0 || hints->max_partition_id < 0))
        return fail(ctx, LA_EINVAL, "LA_HINT_BOUNDS: max_lag and max_partition_id must be >= 0");
    ctx->hints = *hints;
    ctx->hints_set = true;
    return LA_OK;
}

LA_API void* la_host_alloc(la_ctx* ctx, size_t bytes) {
    DeviceGuard restore_device;
    if (!ctx || ctx->shards.empty()) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(ctx->shards[0].device) != hipSuccess) return nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable);
    if (e != hipSuccess) {
        (void)fail(ctx, LA_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}

LA_API void la_host_free(la_ctx* ctx, void* p) {
    (void)ctx;
    if (p) (void)hipHostFree(p);
}

LA_API int la_compute_lag(la_ctx* ctx, int64_t n, const int64_t* begin_off, const int64_t* end_off,
                          const int64_t* committed_off, int32_t reset_mode, int64_t* out_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        if (n < 0) return fail(ctx, LA_EINVAL, "n < 0");
        if (n == 0) return LA_OK;
        if (!end_off || !committed_off || !out_lag) return fail(ctx, LA_EINVAL, "null buffer");
        const bool latest = reset_mode == LA_RESET_LATEST;
        if (!latest && !begin_off) return fail(ctx, LA_EINVAL, "begin_off is required unless reset_mode is LA_RESET_LATEST");
        ctx->last_valid = false;                       // reuses the scratch the last results live in
        // elementwise and PCIe-bound: contiguous element ranges over the shards (every device has its own link), each on
        // its shard's first lane
        int S = (int)ctx->shards.size();
        if (!ctx->split_always) {
            const int64_t by_size = n / (4 * kMinShardPartitions);
            if (by_size < S) S = by_size < 1 ? 1 : (int)by_size;
        }
        if ((int64_t)S > n) S = (int)n;
        auto run_range = [&](int i) -> int {
            Shard& sh = ctx->shards[(size_t)i];
            const int64_t e0 = n / S * i + (n % S) * i / S, e1 = n / S * (i + 1) + (n % S) * (i + 1) / S;
            const int64_t m = e1 - e0;
            if (m <= 0) return LA_OK;
            LA_HIP(ctx, hipSetDevice(sh.device));
            const size_t nb = (size_t)m * 8;
            int rc;
            if ((rc = reserve(ctx, sh.end, nb)) || (rc = reserve(ctx, sh.committed, nb)) ||
                (rc = reserve(ctx, sh.begin, latest ? 16 : nb)) || (rc = reserve(ctx, sh.out_total, nb)))
                return rc;
            hipStream_t st = sh.lanes[0].stream;
            LA_HIP(ctx, hipMemcpyAsync(sh.end.p, end_off + e0, nb, hipMemcpyHostToDevice, st));
            LA_HIP(ctx, hipMemcpyAsync(sh.committed.p, committed_off + e0, nb, hipMemcpyHostToDevice, st));
            if (!latest) LA_HIP(ctx, hipMemcpyAsync(sh.begin.p, begin_off + e0, nb, hipMemcpyHostToDevice, st));
            LA_HIP(ctx, la::lag_launch(m, latest ? nullptr : (const int64_t*)sh.begin.p, (const int64_t*)sh.end.p,
                                       (const int64_t*)sh.committed.p, latest, (int64_t*)sh.out_total.p, st));
            LA_HIP(ctx, hipMemcpyAsync(out_lag + e0, sh.out_total.p, nb, hipMemcpyDeviceToHost, st));
            LA_HIP(ctx, hipStreamSynchronize(st));
            return LA_OK;
        };
        if (S == 1) return run_range(0);
        return run_workers(ctx, S, run_range);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_compute_lag");
    }
}

LA_API int la_assign_batch(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                           const int64_t* begin_off, const int64_t* end_off, const int64_t* committed_off,
                           int32_t reset_mode, const int64_t* cons_off, const int32_t* cons_rank,
                           int32_t* out_partition, int32_t* out_member_rank, int64_t* out_total_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    HintSpender spend_hints(ctx);
    try {
        return assign_host(ctx, n_topics, part_off, partition_id, begin_off, end_off, committed_off, nullptr,
                           reset_mode, cons_off, cons_rank, out_partition, out_member_rank, out_total_lag);
    } catch (...) {
        return ctx ? fail(ctx, LA_ENOMEM, "exception in la_assign_batch") : LA_EINVAL;
    }
}

LA_API int la_assign_batch_lags(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                                const int64_t* lag, const int64_t* cons_off, const int32_t* cons_rank,
                                int32_t* out_partition, int32_t* out_member_rank, int64_t* out_total_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    HintSpender spend_hints(ctx);
    try {
        if (ctx && !lag && n_topics > 0 && part_off && part_off[n_topics] > 0)
            return fail(ctx, LA_EINVAL, "lag is NULL");
        static const int64_t dummy = 0;
        return assign_host(ctx, n_topics, part_off, partition_id, nullptr, nullptr, nullptr, lag ? lag : &dummy,
                           LA_RESET_LATEST, cons_off, cons_rank, out_partition, out_member_rank, out_total_lag);
    } catch (...) {
        return ctx ? fail(ctx, LA_ENOMEM, "exception in la_assign_batch_lags") : LA_EINVAL;
    }
}

LA_API int la_assign_batch_device_on(la_ctx* ctx, int shard, const la_device_batch* batch, void* stream) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        if (shard < 0 || shard >= (int)ctx->shards.size()) return fail(ctx, LA_EINVAL, "shard %d of %d", shard, (int)ctx->shards.size());
        if (!batch) return fail(ctx, LA_EINVAL, "batch is NULL");
        Shard& sh = ctx->shards[(size_t)shard];        // device buffers belong to one device: this shard's
        LA_HIP(ctx, hipSetDevice(sh.device));
        return enqueue_batch(ctx, sh.lanes[0], batch, (hipStream_t)stream);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_assign_batch_device");
    }
}

LA_API int la_assign_batch_device(la_ctx* ctx, const la_device_batch* batch, void* stream) {
    return la_assign_batch_device_on(ctx, 0, batch, stream);
}

LA_API void* la_shard_stream(la_ctx* ctx, int shard) {
    if (!ctx || shard < 0 || shard >= (int)ctx->shards.size()) return nullptr;
    return (void*)ctx->shards[(size_t)shard].lanes[0].stream;
}

LA_API void* la_stream(la_ctx* ctx) { return la_shard_stream(ctx, 0); }

LA_API int la_group_by_member_device(la_ctx* ctx, int32_t n_topics, int64_t n_partitions, const int64_t* d_part_off,
                                     const int32_t* d_out_partition, const int32_t* d_out_member_rank,
                                     int32_t n_members, int64_t* d_member_off, int32_t* d_grouped_topic,
                                     int32_t* d_grouped_partition, void* stream) {
    return la_group_by_member_device_on(ctx, 0, n_topics, n_partitions, d_part_off, d_out_partition, d_out_member_rank,
                                        n_members, d_member_off, d_grouped_topic, d_grouped_partition, stream);
}

LA_API int la_group_by_member_device_on(la_ctx* ctx, int shard, int32_t n_topics, int64_t n_partitions,
                                        const int64_t* d_part_off, const int32_t* d_out_partition,
                                        const int32_t* d_out_member_rank, int32_t n_members, int64_t* d_member_off,
                                        int32_t* d_grouped_topic, int32_t* d_grouped_partition, void* stream) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (shard < 0 || shard >= (int)ctx->shards.size()) return fail(ctx, LA_EINVAL, "shard %d of %d", shard, (int)ctx->shards.size());
        if (n_topics < 0 || n_partitions < 0 || n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!d_member_off) return fail(ctx, LA_EINVAL, "member_off is NULL");
        if (n_partitions > 0 && (!d_out_partition || !d_out_member_rank || !d_grouped_partition ||
                                 (d_grouped_topic && !d_part_off)))
            return fail(ctx, LA_EINVAL, "null buffer");
        if (n_partitions > 0x7FFFFFFF) return fail(ctx, LA_ESHAPE, "at most 2^31-1 entries are supported");
        Shard& sh = ctx->shards[(size_t)shard];
        LA_HIP(ctx, hipSetDevice(sh.device));
        hipError_t e = la::group_by_member_launch(sh.lanes[0].large, n_partitions, n_members, n_topics, d_part_off,
                                                  d_out_partition, d_out_member_rank, d_member_off,
                                                  d_grouped_topic, d_grouped_partition, nullptr, sh.lanes[0].d_status,
                                                  (hipStream_t)stream);
        if (e != hipSuccess)
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "group_by_member: %s", hipGetErrorString(e));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_by_member_device");
    }
}

// Every member's list from the results the shards hold (Shard::last_*): one shard -> its CSR straight into the caller's
// arrays; several -> grouped per shard on its device, merged by offset on the host.
static int group_last_impl(la_ctx* ctx, int32_t n_members, int64_t* member_off, int32_t* grouped_topic,
                           int32_t* grouped_partition) {
    const int S = ctx->last_shards;
    int64_t n = 0;
    for (int i = 0; i < S; ++i) n += ctx->shards[(size_t)i].last_n;
    if (!member_off || (n > 0 && !grouped_partition)) return fail(ctx, LA_EINVAL, "null buffer");
    const size_t mb = ((size_t)n_members + 1) * 8;
    if (S == 1) {
        // one shard: its CSR is the answer, straight into the caller's arrays
        Shard& sh = ctx->shards[0];
        LA_HIP(ctx, hipSetDevice(sh.device));
        hipStream_t st = sh.lanes[0].stream;
        const size_t nb4 = (size_t)n * 4;
        const size_t o_topic = (mb + 16 + 255) & ~(size_t)255, o_part = (o_topic + nb4 + 16 + 255) & ~(size_t)255;
        const size_t g_total = o_part + nb4 + 16;
        if (g_total <= kGroupStageBytes) {
            // small batch: the CSR is built in one staging buffer and crosses in one copy (see assign_small)
            int rc;
            if ((rc = reserve(ctx, sh.small_g, g_total)) || (rc = reserve_host(ctx, sh.small_gh, g_total))) return rc;
            char* d = (char*)sh.small_g.p;
            char* h = (char*)sh.small_gh.p;
            if ((rc = group_shard_device(ctx, sh, n_members, grouped_topic != nullptr, (int64_t*)d,
                                         (int32_t*)(d + o_topic), (int32_t*)(d + o_part))))
                return rc;
            LA_HIP(ctx, hipMemcpyAsync(h, d, g_total, hipMemcpyDeviceToHost, st));
            if ((rc = sync_status(ctx, sh.lanes[0], st))) return rc;    // the grouping's sort reports through the lane's status word
            memcpy(member_off, h, mb);
            if (n) {
                memcpy(grouped_partition, h + o_part, nb4);
                if (grouped_topic) memcpy(grouped_topic, h + o_topic, nb4);
            }
            return LA_OK;
        }
        if (int rc = group_shard_device(ctx, sh, n_members, grouped_topic != nullptr)) return rc;
        LA_HIP(ctx, hipMemcpyAsync(member_off, sh.out_total.p, mb, hipMemcpyDeviceToHost, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(grouped_partition, sh.pid.p, nb4, hipMemcpyDeviceToHost, st));
            if (grouped_topic) LA_HIP(ctx, hipMemcpyAsync(grouped_topic, sh.cons_rank.p, nb4, hipMemcpyDeviceToHost, st));
        }
        return sync_status(ctx, sh.lanes[0], st);
    }
    // Several shards.  A member's list is its per-topic appends in topic order (Main.java:177-184, :264), and the
    // shards are contiguous topic ranges: the list is the concatenation, in shard order, of the shards' lists.
    // Phase 1, per shard in parallel: group on the device, CSR into pinned staging.  Then the global offsets
    // (a scan over members x shards on the host), then phase 2, per shard in parallel: every member's slice to
    // its place in the caller's arrays, topic indices moved from shard-local to the caller's numbering.
    int rc = run_workers(ctx, S, [&](int i) -> int {
        Shard& sh = ctx->shards[(size_t)i];
        if (sh.last_topics == 0) return LA_OK;
        LA_HIP(ctx, hipSetDevice(sh.device));
        const size_t nb4 = (size_t)sh.last_n * 4;
        int r;
        if ((r = reserve_host(ctx, sh.g_off, mb)) || (r = reserve_host(ctx, sh.g_part, nb4 + 16)) ||
            (grouped_topic && (r = reserve_host(ctx, sh.g_topic, nb4 + 16))))
            return r;
        if ((r = group_shard_device(ctx, sh, n_members, grouped_topic != nullptr))) return r;
        hipStream_t st = sh.lanes[0].stream;
        LA_HIP(ctx, hipMemcpyAsync(sh.g_off.p, sh.out_total.p, mb, hipMemcpyDeviceToHost, st));
        if (sh.last_n) {
            LA_HIP(ctx, hipMemcpyAsync(sh.g_part.p, sh.pid.p, nb4, hipMemcpyDeviceToHost, st));
            if (grouped_topic) LA_HIP(ctx, hipMemcpyAsync(sh.g_topic.p, sh.cons_rank.p, nb4, hipMemcpyDeviceToHost, st));
        }
        return sync_status(ctx, sh.lanes[0], st);
    });
    if (rc) return rc;
    // group g = 0 is "no consumer" (rank -1: the entries before member_off[0]), g = r + 1 is member r
    const size_t G = (size_t)n_members + 1;
    std::vector<int64_t> base((size_t)S * G);          // destination of shard i's slice of group g
    {
        int64_t run = 0;
        for (size_t g = 0; g < G; ++g) {
            if (g >= 1) member_off[g - 1] = run;
            for (int i = 0; i < S; ++i) {
                const Shard& sh = ctx->shards[(size_t)i];
                base[(size_t)i * G + g] = run;
                if (sh.last_topics == 0) continue;
                const int64_t* so = (const int64_t*)sh.g_off.p;
                const int64_t lo = g == 0 ? 0 : so[g - 1], hi = g == G - 1 ? sh.last_n : so[g];
                run += hi - lo;
            }
        }
        member_off[n_members] = run;
    }
    rc = run_workers(ctx, S, [&](int i) -> int {
        const Shard& sh = ctx->shards[(size_t)i];
        if (sh.last_topics == 0 || sh.last_n == 0) return LA_OK;
        const int64_t* so = (const int64_t*)sh.g_off.p;
        const int32_t* sp = (const int32_t*)sh.g_part.p;
        const int32_t* stp = (const int32_t*)sh.g_topic.p;
        for (size_t g = 0; g < G; ++g) {
            const int64_t lo = g == 0 ? 0 : so[g - 1], hi = g == G - 1 ? sh.last_n : so[g];
            if (hi <= lo) continue;
            const int64_t dst = base[(size_t)i * G + g];
            memcpy(grouped_partition + dst, sp + lo, (size_t)(hi - lo) * 4);
            if (grouped_topic) {
                int32_t* gt = grouped_topic + dst;
                const int32_t t0 = sh.last_t0;
                for (int64_t j = lo; j < hi; ++j) gt[j - lo] = stp[j] + t0;
            }
        }
        return LA_OK;
    });
    return rc;
}

LA_API int la_group_by_member(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* out_partition,
                              const int32_t* out_member_rank, int32_t n_members, int64_t* member_off,
                              int32_t* grouped_topic, int32_t* grouped_partition) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        ctx->last_valid = false;                       // this call reuses the scratch the last results live in
        if (n_topics < 0 || n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!member_off || (n_topics > 0 && !part_off)) return fail(ctx, LA_EINVAL, "null buffer");
        const int64_t n = n_topics > 0 ? part_off[n_topics] : 0;
        if (n < 0) return fail(ctx, LA_EINVAL, "part_off decreases");
        if (n > 0 && (!out_partition || !out_member_rank || !grouped_partition)) return fail(ctx, LA_EINVAL, "null buffer");
        // A large input is split like an assign call's: contiguous topic ranges over the shards (la_plan_shards), each
        // grouped on its own device, the lists concatenated in shard order (a member's list is its per-topic appends in
        // topic order, Main.java:177-184, :264) -- the code la_group_last_by_member runs on results already resident.
        int S = (int)ctx->shards.size();
        if (!ctx->split_always) {
            const int64_t by_size = n / (4 * kMinShardPartitions);
            if (by_size < S) S = by_size < 1 ? 1 : (int)by_size;
        }
        if (S > n_topics) S = n_topics;
        if (S > 1) {
            for (int32_t t = 0; t < n_topics; ++t)
                if (part_off[t + 1] < part_off[t]) return fail(ctx, LA_EINVAL, "part_off decreases");
            plan_ranges(part_off, 0, n_topics, S, ctx->last_bounds);
            ctx->last_shards = S;
            int rc = run_workers(ctx, S, [&](int i) -> int {
                Shard& sh = ctx->shards[(size_t)i];
                const int32_t t0 = ctx->last_bounds[i], t1 = ctx->last_bounds[i + 1];
                const int64_t p0 = part_off[t0], m = part_off[t1] - p0;
                sh.last_t0 = t0; sh.last_topics = t1 - t0; sh.last_p0 = p0; sh.last_n = m;
                if (t1 == t0) return LA_OK;
                LA_HIP(ctx, hipSetDevice(sh.device));
                const size_t tb = (size_t)(t1 - t0 + 1) * 8, nb4 = (size_t)m * 4;
                int r;
                if ((r = reserve(ctx, sh.part_off, tb + 16)) || (r = reserve(ctx, sh.out_pid, nb4 + 16)) ||
                    (r = reserve(ctx, sh.out_rank, nb4 + 16)))
                    return r;
                sh.local_part_off.resize((size_t)(t1 - t0) + 1);
                for (int32_t t = 0; t <= t1 - t0; ++t) sh.local_part_off[(size_t)t] = part_off[t0 + t] - p0;
                hipStream_t st = sh.lanes[0].stream;
                LA_HIP(ctx, hipMemcpyAsync(sh.part_off.p, sh.local_part_off.data(), tb, hipMemcpyHostToDevice, st));
                if (m) {
                    LA_HIP(ctx, hipMemcpyAsync(sh.out_pid.p, out_partition + p0, nb4, hipMemcpyHostToDevice, st));
                    LA_HIP(ctx, hipMemcpyAsync(sh.out_rank.p, out_member_rank + p0, nb4, hipMemcpyHostToDevice, st));
                }
                LA_HIP(ctx, hipStreamSynchronize(st));           // local_part_off is read by the copy until here
                sh.last_part_off = (const int64_t*)sh.part_off.p;
                sh.last_out_pid = (const int32_t*)sh.out_pid.p;
                sh.last_out_rank = (const int32_t*)sh.out_rank.p;
                return LA_OK;
            });
            if (rc) return rc;
            return group_last_impl(ctx, n_members, member_off, grouped_topic, grouped_partition);
        }
        Shard& sh = ctx->shards[0];                    // one stable sort of the whole array: the first device
        LA_HIP(ctx, hipSetDevice(sh.device));
        const size_t nb4 = (size_t)n * 4, tb = (size_t)(n_topics + 1) * 8, mb = ((size_t)n_members + 1) * 8;
        int rc;
        // scratch reuse: out_pid <- out_partition, out_rank <- member ranks, pid <- grouped_partition,
        // cons_rank <- grouped_topic, out_total <- member_off
        if ((rc = reserve(ctx, sh.part_off, tb + 16)) || (rc = reserve(ctx, sh.out_pid, nb4 + 16)) ||
            (rc = reserve(ctx, sh.out_rank, nb4 + 16)) || (rc = reserve(ctx, sh.pid, nb4 + 16)) ||
            (rc = reserve(ctx, sh.cons_rank, nb4 + 16)) || (rc = reserve(ctx, sh.out_total, mb + 16)))
            return rc;
        hipStream_t st = sh.lanes[0].stream;
        if (n_topics > 0) LA_HIP(ctx, hipMemcpyAsync(sh.part_off.p, part_off, tb, hipMemcpyHostToDevice, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(sh.out_pid.p, out_partition, nb4, hipMemcpyHostToDevice, st));
            LA_HIP(ctx, hipMemcpyAsync(sh.out_rank.p, out_member_rank, nb4, hipMemcpyHostToDevice, st));
        }
        rc = la_group_by_member_device(ctx, n_topics, n, (const int64_t*)sh.part_off.p, (const int32_t*)sh.out_pid.p,
                                       (const int32_t*)sh.out_rank.p, n_members, (int64_t*)sh.out_total.p,
                                       grouped_topic ? (int32_t*)sh.cons_rank.p : nullptr, (int32_t*)sh.pid.p, st);
        if (rc) return rc;
        LA_HIP(ctx, hipMemcpyAsync(member_off, sh.out_total.p, mb, hipMemcpyDeviceToHost, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(grouped_partition, sh.pid.p, nb4, hipMemcpyDeviceToHost, st));
            if (grouped_topic) LA_HIP(ctx, hipMemcpyAsync(grouped_topic, sh.cons_rank.p, nb4, hipMemcpyDeviceToHost, st));
        }
        return sync_status(ctx, sh.lanes[0], st);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_by_member");
    }
}

LA_API int la_group_last_by_member(la_ctx* ctx, int32_t n_members, int64_t* member_off, int32_t* grouped_topic,
                                   int32_t* grouped_partition) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        if (!ctx->last_valid)
            return fail(ctx, LA_EINVAL, "no result of la_assign_batch / la_assign_batch_lags is held on the device");
        if (n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        return group_last_impl(ctx, n_members, member_off, grouped_topic, grouped_partition);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_last_by_member");
    }
}

static int assign_grouped(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                          const int64_t* begin_off, const int64_t* end_off, const int64_t* committed_off, int32_t reset_mode,
                          const int64_t* cons_off, const int32_t* cons_rank, int32_t n_members, int64_t* member_off,
                          int32_t* grouped_topic, int32_t* grouped_partition, int64_t* out_total_lag, const HostCall* sparse) {
    if (n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
    if (!member_off) return fail(ctx, LA_EINVAL, "member_off is NULL");
    if (n_topics > 0 && part_off && part_off[n_topics] > 0 && !grouped_partition)
        return fail(ctx, LA_EINVAL, "grouped_partition is NULL");
    if (n_topics <= 0) {                                        // nothing assigned: every member's list is empty
        if (n_topics < 0) return fail(ctx, LA_EINVAL, "n_topics < 0");
        ctx->last_valid = false;
        for (int32_t r = 0; r <= n_members; ++r) member_off[r] = 0;
        return LA_OK;
    }
    bool done = false;
    HostCall g;
    g.g_members = n_members; g.g_off = member_off; g.g_topic = grouped_topic; g.g_part = grouped_partition;
    g.grouped_done = &done;
    if (int rc = assign_host(ctx, n_topics, part_off, partition_id, begin_off, end_off, committed_off, nullptr, reset_mode,
                             cons_off, cons_rank, nullptr, nullptr, out_total_lag, &g, sparse))
        return rc;
    if (done) return LA_OK;                                      // a small call: the lists came back with the totals
    if (!ctx->last_valid) {                                      // (no partitions at all)
        for (int32_t r = 0; r <= n_members; ++r) member_off[r] = 0;
        return LA_OK;
    }
    return group_last_impl(ctx, n_members, member_off, grouped_topic, grouped_partition);
}

// la_wake: ONE-PARTITION REBALANCE through the path a real small call takes (zero-copy staging, the tile kernel with its fused
// end, the spin on the completion word), waited for, and one empty kernel on every other stream of the context.  Why a whole
// call and why synchronous (tools/cold_probe.py through ctypes, profiles/r06_j_cold_probe.txt, one box, a 100-partition grouped
// call: 30 us back to back, 135-175 us after 1 s of idle): an empty asynchronous launch leaves it at 107-124 us, a 1-element
// la_compute_lag (copies, a kernel, a stream wait) at 69-74, a dummy rebalance through the real path at 43-45.  At the C ABI
// (tools/cold_c.c, profiles/r06_p_cold_c.txt): 23 us back to back, 85 us after 1 s of idle, 34 us with la_wake 1-20 ms before
// it.  What is cold after an idle second is not one thing (the queue, the link, the mapped pages' translations, the runtime's
// and this library's own code and data in the host's caches): the cheapest way to warm all of it is to do the thing once.  It
// costs the caller the cold call it spares the rebalance (~95 us) -- at the top of assign(), where milliseconds of broker round
// trips follow anyway.  The pending hint and the diagnostics of the last real call survive; results kept on the device do not.
LA_API int la_wake(la_ctx* ctx) {
    if (!ctx) return LA_EINVAL;
    DeviceGuard restore_device;
    const la_call_hints saved_hints = ctx->hints;
    const bool saved_set = ctx->hints_set;
    const int saved_pipeline = ctx->last_pipeline;
    const int64_t saved_launches = ctx->last_launches;
    ctx->hints_set = false;
    int rc = LA_OK;
    try {
        static const int64_t off[2] = {0, 1}, begin[1] = {0}, end[1] = {1}, committed[1] = {0};
        static const int32_t pid[1] = {0}, rank[1] = {0};
        int64_t member_off[2] = {0, 0};
        int32_t grouped_partition[1] = {0};
        rc = assign_grouped(ctx, 1, off, pid, begin, end, committed, LA_RESET_EARLIEST, off, rank, 1, member_off, nullptr,
                            grouped_partition, nullptr, nullptr);
        if (rc == LA_OK && !(member_off[1] == 1 && grouped_partition[0] == 0))
            rc = fail(ctx, LA_EHIP, "la_wake: the one-partition rebalance came back wrong (member_off %lld %lld, partition %d)",
                      (long long)member_off[0], (long long)member_off[1], (int)grouped_partition[0]);
    } catch (...) {
        rc = fail(ctx, LA_ENOMEM, "exception in la_wake");
    }
    ctx->hints = saved_hints;
    ctx->hints_set = saved_set;
    ctx->last_pipeline = saved_pipeline;
    ctx->last_launches = saved_launches;
    ctx->last_valid = false;                                  // (the staging buffers of the last real call were reused)
    if (rc != LA_OK) return rc;
    for (size_t s = 0; s < ctx->shards.size(); ++s) {
        Shard& sh = ctx->shards[s];
        LA_HIP(ctx, hipSetDevice(sh.device));
        for (size_t i = (s == 0 ? 1 : 0); i < sh.lanes.size(); ++i)          // (shard 0's first lane just ran the call)
            if (sh.lanes[i].stream) LA_HIP(ctx, la::wake_launch(sh.lanes[i].stream));
    }
    return LA_OK;
}

LA_API int la_assign_batch_grouped(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                                   const int64_t* begin_off, const int64_t* end_off, const int64_t* committed_off,
                                   int32_t reset_mode, const int64_t* cons_off, const int32_t* cons_rank, int32_t n_members,
                                   int64_t* member_off, int32_t* grouped_topic, int32_t* grouped_partition,
                                   int64_t* out_total_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    HintSpender spend_hints(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        return assign_grouped(ctx, n_topics, part_off, partition_id, begin_off, end_off, committed_off, reset_mode, cons_off,
                              cons_rank, n_members, member_off, grouped_topic, grouped_partition, out_total_lag, nullptr);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_assign_batch_grouped");
    }
}

LA_API int la_assign_batch_sparse(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                                  const int64_t* end_off, const int64_t* committed_off, int32_t reset_mode, int64_t n_none,
                                  const int64_t* none_index, const int64_t* none_begin, const int64_t* cons_off,
                                  const int32_t* cons_rank, int32_t* out_partition, int32_t* out_member_rank,
                                  int64_t* out_total_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    HintSpender spend_hints(ctx);
    try {
        HostCall sp;
        sp.n_none = n_none; sp.none_index = none_index; sp.none_begin = none_begin;
        return assign_host(ctx, n_topics, part_off, partition_id, nullptr, end_off, committed_off, nullptr, reset_mode, cons_off,
                           cons_rank, out_partition, out_member_rank, out_total_lag, nullptr, &sp);
    } catch (...) {
        return ctx ? fail(ctx, LA_ENOMEM, "exception in la_assign_batch_sparse") : LA_EINVAL;
    }
}

LA_API int la_assign_batch_grouped_sparse(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                                          const int64_t* end_off, const int64_t* committed_off, int32_t reset_mode,
                                          int64_t n_none, const int64_t* none_index, const int64_t* none_begin,
                                          const int64_t* cons_off, const int32_t* cons_rank, int32_t n_members,
                                          int64_t* member_off, int32_t* grouped_topic, int32_t* grouped_partition,
                                          int64_t* out_total_lag) {
    DeviceGuard restore_device;
    LaunchSpan span(ctx);
    HintSpender spend_hints(ctx);
    if (!ctx) return LA_EINVAL;
    try {
        HostCall sp;
        sp.n_none = n_none; sp.none_index = none_index; sp.none_begin = none_begin;
        return assign_grouped(ctx, n_topics, part_off, partition_id, nullptr, end_off, committed_off, reset_mode, cons_off, cons_rank,
                              n_members, member_off, grouped_topic, grouped_partition, out_total_lag, &sp);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_assign_batch_grouped_sparse");
    }
}

LA_API int la_last_phase_times(la_ctx* ctx, la_phase_times* out) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (!out) return fail(ctx, LA_EINVAL, "out is NULL");
        Shard& sh = ctx->shards[0];
        LA_HIP(ctx, hipSetDevice(sh.device));
        float ms[3] = {};
        int passes[4] = {};
        int64_t n = 0;
        const hipError_t e = la::large_profile_read(sh.lanes[0].large, ms, passes, &n);
        if (e == hipErrorNotReady)
            return fail(ctx, LA_EINVAL, "no large-path topic was profiled (LA_FLAG_PROFILE on la_assign_batch_device)");
        if (e != hipSuccess) return fail(ctx, LA_EHIP, "la_last_phase_times: %s", hipGetErrorString(e));
        out->n_partitions = n;
        out->id_passes = passes[0];
        out->key_passes = passes[1];
        out->keys_ms = ms[0];
        out->sort_ms = ms[1];
        out->greedy_ms = ms[2];
        out->keys_first = passes[2];
        out->redone = passes[3];
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_last_phase_times");
    }
}

LA_API int la_last_phase_times_sized(la_ctx* ctx, void* out, size_t out_size) {
    if (!ctx) return LA_EINVAL;
    if (!out) return fail(ctx, LA_EINVAL, "out is NULL");
    la_phase_times full{};
    if (int rc = la_last_phase_times(ctx, &full)) return rc;
    memcpy(out, &full, out_size < sizeof full ? out_size : sizeof full);
    return LA_OK;
}

LA_API int la_sync_on(la_ctx* ctx, int shard, void* stream) {
    DeviceGuard restore_device;
    if (!ctx) return LA_EINVAL;
    try {
        if (shard < 0 || shard >= (int)ctx->shards.size()) return fail(ctx, LA_EINVAL, "shard %d of %d", shard, (int)ctx->shards.size());
        Shard& sh = ctx->shards[(size_t)shard];
        LA_HIP(ctx, hipSetDevice(sh.device));
        return sync_status(ctx, sh.lanes[0], (hipStream_t)stream);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_sync");
    }
}

LA_API int la_sync(la_ctx* ctx, void* stream) { return la_sync_on(ctx, 0, stream); }
