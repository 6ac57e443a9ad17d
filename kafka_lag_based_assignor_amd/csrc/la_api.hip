// la_api.hip -- the C ABI of include/lagassign.h over the HIP kernels.
//
// There is no CPU fallback here by design: if the device or a kernel is unavailable the
// call fails with a negative code and the caller decides what to do.
#include "../../include/lagassign.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "la_kernels.h"

#define LA_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct la_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t* d_status = nullptr;
    std::string err;
    // grow-only device scratch for the host-buffer entry points
    DevBuf part_off, pid, begin, end, committed, cons_off, cons_rank, out_pid, out_rank, out_total;
    // scratch of the large-topic path
    la::LargeScratch large;
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
    std::vector<uint8_t> topic_class;   // host scratch of the dispatcher: path / class of every topic
    std::vector<int64_t> host_offsets;  // offsets fetched from the device when the caller gave no host copy
    // results of the last host-buffer assign call, still in part_off / out_pid / out_rank (la_group_last_by_member)
    bool last_valid = false;
    int32_t last_topics = 0;
    int64_t last_n = 0;
};

namespace {

int fail(la_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
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
int plan_batch(la_ctx* ctx, const la_device_batch* b, int tile_mode, bool use_block, BatchPlan* plan) {
    const int64_t T = b->n_topics;
    ctx->topic_class.resize((size_t)T);
    uint8_t* code = ctx->topic_class.data();
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
int stage_topic_lists(la_ctx* ctx, const BatchPlan& plan, int64_t T, hipStream_t stream, const int32_t** d_lists) {
    *d_lists = nullptr;
    if (plan.n_lists == 0 || (!plan.classed && plan.n_block_all <= 8)) return LA_OK;   // few block topics go inline
    la_ctx::Stage& sg = ctx->stage[ctx->stage_next++ & 3u];
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
    if (int rc = reserve(ctx, ctx->block_list, (size_t)plan.n_lists * sizeof(int32_t))) return rc;
    LA_HIP(ctx, hipMemcpyAsync(ctx->block_list.p, sg.p, (size_t)plan.n_lists * sizeof(int32_t), hipMemcpyHostToDevice,
                               stream));
    LA_HIP(ctx, hipEventRecord(sg.done, stream));
    *d_lists = (const int32_t*)ctx->block_list.p;
    return LA_OK;
}

int launch_block_topics(la_ctx* ctx, const la_device_batch* b, const BatchPlan& plan, const int32_t* d_lists,
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
    g.status = ctx->d_status;
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

int launch_large_topics(la_ctx* ctx, const la_device_batch* b, const BatchPlan& plan, bool argmin, hipStream_t stream) {
    for (int64_t t = 0; t < b->n_topics; ++t) {
        if (plan.code[t] != kLargeCode) continue;
        const int64_t p = b->h_part_off[t + 1] - b->h_part_off[t], c = b->h_cons_off[t + 1] - b->h_cons_off[t];
        if (c > la::kLargeMaxConsumers)
            return fail(ctx, LA_ESHAPE, "topic %lld has %lld consumers; at most %lld are supported", (long long)t,
                        (long long)c, (long long)la::kLargeMaxConsumers);
        if (p > 0x7FFFFFFF)
            return fail(ctx, LA_ESHAPE, "topic %lld has %lld partitions; at most 2^31-1 are supported", (long long)t,
                        (long long)p);
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
        g.status = ctx->d_status;
        hipError_t e = la::large_topic_launch(ctx->large, g, argmin, stream);
        if (e != hipSuccess)
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "large topic %lld: %s", (long long)t,
                        hipGetErrorString(e));
    }
    return LA_OK;
}

// The dispatcher shared by the host and device entry points.
int enqueue_batch(la_ctx* ctx, const la_device_batch* b, hipStream_t stream) {
    if (b->n_topics < 0 || b->n_partitions < 0 || b->n_consumers < 0)
        return fail(ctx, LA_EINVAL, "negative size");
    if (b->n_topics == 0) return LA_OK;
    if (!b->d_part_off || !b->d_cons_off || !b->d_out_partition || !b->d_out_member_rank)
        return fail(ctx, LA_EINVAL, "null offsets or outputs");
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
    a.status = ctx->d_status;
    a.reset_latest = (b->reset_mode == LA_RESET_LATEST) ? 1 : 0;
    a.n_total = b->n_partitions;
    a.flags = b->flags & (LA_FLAG_INDEX64 | LA_FLAG_DEFER_WIDE);
    a.topic_list = nullptr;
    a.k_total = b->n_consumers;
    if (int rc = reserve(ctx, ctx->defer, la::wave_tile_defer_bytes(b->n_topics))) return rc;
    a.defer_list = (int32_t*)ctx->defer.p;
    const bool argmin = (b->algo == LA_ALGO_ARGMIN);
    const int tile_mode = argmin ? 2 : (b->algo == LA_ALGO_ROUNDS_WIDE ? 1 : 0);

    // The counter pair alternates per LA_ALGO_AUTO launch: such a launch counts into one and its wide
    // kernel zeroes the other (idle by stream order), so no memset node sits between launches.
    // Under stream capture the arguments are frozen into the graph, so the alternation cannot work on replay: a
    // captured launch counts into a third word that a memset node clears first (its wide kernel "resets" a dummy).
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (tile_mode == 0) LA_HIP(ctx, hipStreamIsCapturing(stream, &capture));
    hipError_t counter_err = hipSuccess;
    auto next_counters = [&](la::TileArgs& t) {
        int32_t* pair = (int32_t*)(ctx->d_status + 16);
        if (capture != hipStreamCaptureStatusNone) {
            t.defer_count = pair + 2;
            t.defer_count_next = pair + 3;
            const hipError_t e = hipMemsetAsync(pair + 2, 0, sizeof(int32_t), stream);
            if (e != hipSuccess) counter_err = e;
            return;
        }
        t.defer_count = pair + (ctx->launches & 1u);
        t.defer_count_next = pair + ((ctx->launches + 1u) & 1u);
        if (tile_mode == 0) ++ctx->launches;
    };
    if (b->n_partitions == 0) {
        // nothing to assign; consumers of partition-less topics still report a total of 0
        if (b->d_out_total_lag && b->n_consumers > 0)
            LA_HIP(ctx, hipMemsetAsync(b->d_out_total_lag, 0, (size_t)b->n_consumers * sizeof(int64_t), stream));
        return LA_OK;
    }
    const bool have_host = b->h_part_off && b->h_cons_off;
    const bool fits_hint = la::wave_tile_fits(b->max_partitions_per_topic, b->max_consumers_per_topic);
    if (fits_hint && !(have_host && (b->flags & LA_FLAG_RAGGED) && tile_mode == 0)) {
        // the plain case: every topic fits a wave tile, one shape for all, nothing read on the host
        next_counters(a);
        LA_HIP(ctx, counter_err);
        LA_HIP(ctx, la::wave_tile_launch(a, b->max_partitions_per_topic, b->max_consumers_per_topic, tile_mode, stream));
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
        ctx->host_offsets.resize(2 * words);
        LA_HIP(ctx, hipMemcpyAsync(ctx->host_offsets.data(), b->d_part_off, words * 8, hipMemcpyDeviceToHost, stream));
        LA_HIP(ctx, hipMemcpyAsync(ctx->host_offsets.data() + words, b->d_cons_off, words * 8, hipMemcpyDeviceToHost, stream));
        LA_HIP(ctx, hipStreamSynchronize(stream));
        with_host = *b;
        with_host.h_part_off = ctx->host_offsets.data();
        with_host.h_cons_off = ctx->host_offsets.data() + words;
        b = &with_host;
    }

    // mixed or ragged shapes (see the dispatcher notes above)
    const bool use_block = !argmin && b->algo != LA_ALGO_ROUNDS_WIDE;   // the test-hook algos keep to tile + large
    BatchPlan plan;
    if (int rc = plan_batch(ctx, b, tile_mode, use_block, &plan)) return rc;
    const int32_t* d_lists = nullptr;
    if (int rc = stage_topic_lists(ctx, plan, b->n_topics, stream, &d_lists)) return rc;
    if (plan.classed) {
        for (int k = 0; k < kTileClasses; ++k) {
            if (plan.tile[k].n == 0) continue;
            la::TileArgs run = a;
            run.n_topics = plan.tile[k].n;
            run.topic_list = d_lists + plan.tile_at[k];
            next_counters(run);
            LA_HIP(ctx, counter_err);
            LA_HIP(ctx, la::wave_tile_launch(run, plan.tile[k].mp, plan.tile[k].mc, tile_mode, stream));
        }
    } else if (plan.n_tile > 0) {
        la::TileArgs run = a;
        run.flags |= la::kTileSkipOversize;
        next_counters(run);
        LA_HIP(ctx, counter_err);
        LA_HIP(ctx, la::wave_tile_launch(run, plan.tile_mp, plan.tile_mc, tile_mode, stream));
    }
    if (plan.n_block_all > 0)
        if (int rc = launch_block_topics(ctx, b, plan, d_lists, stream)) return rc;
    if (plan.n_large > 0)
        if (int rc = launch_large_topics(ctx, b, plan, argmin, stream)) return rc;
    return LA_OK;
}

int sync_status(la_ctx* ctx, hipStream_t stream) {
    LA_HIP(ctx, hipStreamSynchronize(stream));
    uint32_t st = 0;
    LA_HIP(ctx, hipMemcpy(&st, ctx->d_status, sizeof st, hipMemcpyDeviceToHost));
    if (st) {
        LA_HIP(ctx, hipMemset(ctx->d_status, 0, sizeof st));
        if (st & la::kStatusUnsorted)
            return fail(ctx, LA_EINVAL, "a topic's cons_rank segment is not strictly ascending");
        return fail(ctx, LA_ESHAPE, "a topic exceeds the batch's shape hint");
    }
    return LA_OK;
}

int assign_host(la_ctx* ctx, int32_t T, const int64_t* part_off, const int32_t* pid, const int64_t* begin,
                const int64_t* end, const int64_t* committed, const int64_t* lag, int32_t reset_mode,
                const int64_t* cons_off, const int32_t* cons_rank, int32_t* out_pid, int32_t* out_rank,
                int64_t* out_total) {
    if (!ctx) return LA_EINVAL;
    ctx->last_valid = false;
    if (T < 0) return fail(ctx, LA_EINVAL, "n_topics < 0");
    if (T == 0) return LA_OK;
    if (!part_off || !cons_off) return fail(ctx, LA_EINVAL, "null offsets");
    Shape s;
    // offsets are checked here; the ascending-rank contract of cons_rank is checked on the device (one pass over
    // K entries there instead of ~1 ns per entry of host time), reported by sync_status as LA_EINVAL
    if (int rc = scan_shape(ctx, T, part_off, cons_off, nullptr, &s)) return rc;
    if (s.n > 0 && (!pid || (!lag && (!end || !committed)))) return fail(ctx, LA_EINVAL, "null per-partition buffer");
    if ((out_pid == nullptr) != (out_rank == nullptr))
        return fail(ctx, LA_EINVAL, "out_partition and out_member_rank must both be given or both be NULL");
    if (s.k > 0 && !cons_rank) return fail(ctx, LA_EINVAL, "null cons_rank");
    if (!lag && reset_mode != LA_RESET_LATEST && !begin && s.n > 0)
        return fail(ctx, LA_EINVAL, "begin_off is required unless reset_mode is LA_RESET_LATEST");
    LA_HIP(ctx, hipSetDevice(ctx->device));

    const size_t nb8 = (size_t)s.n * 8, nb4 = (size_t)s.n * 4, kb8 = (size_t)s.k * 8, kb4 = (size_t)s.k * 4;
    const size_t tb = (size_t)(T + 1) * 8;
    const bool use_begin = !lag && begin && reset_mode != LA_RESET_LATEST;
    int rc;
    if ((rc = reserve(ctx, ctx->part_off, tb)) || (rc = reserve(ctx, ctx->cons_off, tb)) ||
        (rc = reserve(ctx, ctx->pid, nb4 + 16)) || (rc = reserve(ctx, ctx->end, nb8 + 16)) ||
        (rc = reserve(ctx, ctx->committed, lag ? 16 : nb8 + 16)) ||
        (rc = reserve(ctx, ctx->begin, use_begin ? nb8 + 16 : 16)) ||
        (rc = reserve(ctx, ctx->cons_rank, kb4 + 16)) || (rc = reserve(ctx, ctx->out_pid, nb4 + 16)) ||
        (rc = reserve(ctx, ctx->out_rank, nb4 + 16)) || (rc = reserve(ctx, ctx->out_total, kb8 + 16)))
        return rc;

    hipStream_t st = ctx->stream;
    LA_HIP(ctx, hipMemcpyAsync(ctx->part_off.p, part_off, tb, hipMemcpyHostToDevice, st));
    LA_HIP(ctx, hipMemcpyAsync(ctx->cons_off.p, cons_off, tb, hipMemcpyHostToDevice, st));
    if (s.n) LA_HIP(ctx, hipMemcpyAsync(ctx->pid.p, pid, nb4, hipMemcpyHostToDevice, st));
    if (s.k) LA_HIP(ctx, hipMemcpyAsync(ctx->cons_rank.p, cons_rank, kb4, hipMemcpyHostToDevice, st));
    if (s.n) {
        if (lag) {
            LA_HIP(ctx, hipMemcpyAsync(ctx->end.p, lag, nb8, hipMemcpyHostToDevice, st));
        } else {
            LA_HIP(ctx, hipMemcpyAsync(ctx->end.p, end, nb8, hipMemcpyHostToDevice, st));
            LA_HIP(ctx, hipMemcpyAsync(ctx->committed.p, committed, nb8, hipMemcpyHostToDevice, st));
            if (use_begin) LA_HIP(ctx, hipMemcpyAsync(ctx->begin.p, begin, nb8, hipMemcpyHostToDevice, st));
        }
    }

    if (s.k) LA_HIP(ctx, la::check_consumers_launch(T, (const int64_t*)ctx->cons_off.p, (const int32_t*)ctx->cons_rank.p,
                                                    ctx->d_status, st));

    la_device_batch b{};
    b.n_topics = T;
    b.reset_mode = reset_mode == LA_RESET_LATEST ? LA_RESET_LATEST : LA_RESET_EARLIEST;
    b.algo = LA_ALGO_AUTO;
    b.n_partitions = s.n;
    b.n_consumers = s.k;
    b.max_partitions_per_topic = s.max_p;
    b.max_consumers_per_topic = s.max_c;
    b.d_part_off = (const int64_t*)ctx->part_off.p;
    b.d_partition_id = (const int32_t*)ctx->pid.p;
    b.d_begin_off = use_begin ? (const int64_t*)ctx->begin.p : nullptr;
    b.d_end_off = (const int64_t*)ctx->end.p;
    b.d_committed_off = (const int64_t*)ctx->committed.p;
    b.d_lag = lag ? (const int64_t*)ctx->end.p : nullptr;
    b.d_cons_off = (const int64_t*)ctx->cons_off.p;
    b.d_cons_rank = (const int32_t*)ctx->cons_rank.p;
    b.d_out_partition = (int32_t*)ctx->out_pid.p;
    b.d_out_member_rank = (int32_t*)ctx->out_rank.p;
    b.d_out_total_lag = out_total ? (int64_t*)ctx->out_total.p : nullptr;
    b.h_part_off = part_off;
    b.h_cons_off = cons_off;
    b.flags = LA_FLAG_RAGGED;            // the offsets are on the host anyway: let the dispatcher look at the shapes
    if ((rc = enqueue_batch(ctx, &b, st))) return rc;

    if (s.n && out_pid) {
        LA_HIP(ctx, hipMemcpyAsync(out_pid, ctx->out_pid.p, nb4, hipMemcpyDeviceToHost, st));
        LA_HIP(ctx, hipMemcpyAsync(out_rank, ctx->out_rank.p, nb4, hipMemcpyDeviceToHost, st));
    }
    if (out_total && s.k) LA_HIP(ctx, hipMemcpyAsync(out_total, ctx->out_total.p, kb8, hipMemcpyDeviceToHost, st));
    if ((rc = sync_status(ctx, st))) return rc;
    ctx->last_valid = true;
    ctx->last_topics = T;
    ctx->last_n = s.n;
    return LA_OK;
}

}  // namespace

// ---- C ABI --------------------------------------------------------------------------------------
LA_API int la_version(void) { return 100; }   // 0.1.0

LA_API const char* la_last_error(const la_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

LA_API int la_create(la_ctx** out, int device_id, unsigned flags) {
    (void)flags;
    if (!out) return fail(nullptr, LA_EINVAL, "out is NULL");
    *out = nullptr;
    try {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0)
            return fail(nullptr, LA_ENODEV, "no HIP device (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        if (device_id < 0 || device_id >= count) return fail(nullptr, LA_ENODEV, "device %d of %d", device_id, count);
        hipDeviceProp_t prop;
        if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess)
            return fail(nullptr, LA_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, LA_ENODEV, "device %d is %s; this library is built for gfx950 only", device_id,
                        prop.gcnArchName);
        la_ctx* ctx = new (std::nothrow) la_ctx();
        if (!ctx) return fail(nullptr, LA_ENOMEM, "out of host memory");
        ctx->device = device_id;
        if ((e = hipSetDevice(device_id)) != hipSuccess ||
            (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
            (e = hipMalloc((void**)&ctx->d_status, 256)) != hipSuccess ||
            (e = hipMemset(ctx->d_status, 0, 256)) != hipSuccess) {
            int rc = fail(nullptr, LA_EHIP, "context setup: %s", hipGetErrorString(e));
            la_destroy(ctx);
            return rc;
        }
        *out = ctx;
        return LA_OK;
    } catch (...) {
        return fail(nullptr, LA_ENOMEM, "exception in la_create");
    }
}

LA_API void la_destroy(la_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf* b : {&ctx->part_off, &ctx->pid, &ctx->begin, &ctx->end, &ctx->committed, &ctx->cons_off,
                      &ctx->cons_rank, &ctx->out_pid, &ctx->out_rank, &ctx->out_total, &ctx->defer,
                      &ctx->block_list})
        release(*b);
    for (la_ctx::Stage& sg : ctx->stage) {
        if (sg.done) { (void)hipEventSynchronize(sg.done); (void)hipEventDestroy(sg.done); }
        if (sg.p) (void)hipHostFree(sg.p);
    }
    la::large_scratch_release(ctx->large);
    if (ctx->d_status) (void)hipFree(ctx->d_status);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

LA_API int la_compute_lag(la_ctx* ctx, int64_t n, const int64_t* begin_off, const int64_t* end_off,
                          const int64_t* committed_off, int32_t reset_mode, int64_t* out_lag) {
    if (!ctx) return LA_EINVAL;
    try {
        if (n < 0) return fail(ctx, LA_EINVAL, "n < 0");
        if (n == 0) return LA_OK;
        if (!end_off || !committed_off || !out_lag) return fail(ctx, LA_EINVAL, "null buffer");
        const bool latest = reset_mode == LA_RESET_LATEST;
        if (!latest && !begin_off) return fail(ctx, LA_EINVAL, "begin_off is required unless reset_mode is LA_RESET_LATEST");
        LA_HIP(ctx, hipSetDevice(ctx->device));
        const size_t nb = (size_t)n * 8;
        int rc;
        if ((rc = reserve(ctx, ctx->end, nb)) || (rc = reserve(ctx, ctx->committed, nb)) ||
            (rc = reserve(ctx, ctx->begin, latest ? 16 : nb)) || (rc = reserve(ctx, ctx->out_total, nb)))
            return rc;
        hipStream_t st = ctx->stream;
        LA_HIP(ctx, hipMemcpyAsync(ctx->end.p, end_off, nb, hipMemcpyHostToDevice, st));
        LA_HIP(ctx, hipMemcpyAsync(ctx->committed.p, committed_off, nb, hipMemcpyHostToDevice, st));
        if (!latest) LA_HIP(ctx, hipMemcpyAsync(ctx->begin.p, begin_off, nb, hipMemcpyHostToDevice, st));
        LA_HIP(ctx, la::lag_launch(n, latest ? nullptr : (const int64_t*)ctx->begin.p, (const int64_t*)ctx->end.p,
                                   (const int64_t*)ctx->committed.p, latest, (int64_t*)ctx->out_total.p, st));
        LA_HIP(ctx, hipMemcpyAsync(out_lag, ctx->out_total.p, nb, hipMemcpyDeviceToHost, st));
        LA_HIP(ctx, hipStreamSynchronize(st));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_compute_lag");
    }
}

LA_API int la_assign_batch(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* partition_id,
                           const int64_t* begin_off, const int64_t* end_off, const int64_t* committed_off,
                           int32_t reset_mode, const int64_t* cons_off, const int32_t* cons_rank,
                           int32_t* out_partition, int32_t* out_member_rank, int64_t* out_total_lag) {
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

LA_API int la_assign_batch_device(la_ctx* ctx, const la_device_batch* batch, void* stream) {
    if (!ctx) return LA_EINVAL;
    try {
        if (!batch) return fail(ctx, LA_EINVAL, "batch is NULL");
        LA_HIP(ctx, hipSetDevice(ctx->device));
        return enqueue_batch(ctx, batch, (hipStream_t)stream);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_assign_batch_device");
    }
}

LA_API void* la_stream(la_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

LA_API int la_group_by_member_device(la_ctx* ctx, int32_t n_topics, int64_t n_partitions, const int64_t* d_part_off,
                                     const int32_t* d_out_partition, const int32_t* d_out_member_rank,
                                     int32_t n_members, int64_t* d_member_off, int32_t* d_grouped_topic,
                                     int32_t* d_grouped_partition, void* stream) {
    if (!ctx) return LA_EINVAL;
    try {
        if (n_topics < 0 || n_partitions < 0 || n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!d_member_off) return fail(ctx, LA_EINVAL, "member_off is NULL");
        if (n_partitions > 0 && (!d_out_partition || !d_out_member_rank || !d_grouped_partition ||
                                 (d_grouped_topic && !d_part_off)))
            return fail(ctx, LA_EINVAL, "null buffer");
        if (n_partitions > 0x7FFFFFFF) return fail(ctx, LA_ESHAPE, "at most 2^31-1 entries are supported");
        LA_HIP(ctx, hipSetDevice(ctx->device));
        hipError_t e = la::group_by_member_launch(ctx->large, n_partitions, n_members, n_topics, d_part_off,
                                                  d_out_partition, d_out_member_rank, d_member_off,
                                                  d_grouped_topic, d_grouped_partition, nullptr, (hipStream_t)stream);
        if (e != hipSuccess)
            return fail(ctx, e == hipErrorOutOfMemory ? LA_ENOMEM : LA_EHIP, "group_by_member: %s", hipGetErrorString(e));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_by_member_device");
    }
}

LA_API int la_group_by_member(la_ctx* ctx, int32_t n_topics, const int64_t* part_off, const int32_t* out_partition,
                              const int32_t* out_member_rank, int32_t n_members, int64_t* member_off,
                              int32_t* grouped_topic, int32_t* grouped_partition) {
    if (!ctx) return LA_EINVAL;
    try {
        ctx->last_valid = false;                       // this call reuses the scratch the last results live in
        if (n_topics < 0 || n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        if (!member_off || (n_topics > 0 && !part_off)) return fail(ctx, LA_EINVAL, "null buffer");
        const int64_t n = n_topics > 0 ? part_off[n_topics] : 0;
        if (n < 0) return fail(ctx, LA_EINVAL, "part_off decreases");
        if (n > 0 && (!out_partition || !out_member_rank || !grouped_partition)) return fail(ctx, LA_EINVAL, "null buffer");
        LA_HIP(ctx, hipSetDevice(ctx->device));
        const size_t nb4 = (size_t)n * 4, tb = (size_t)(n_topics + 1) * 8, mb = ((size_t)n_members + 1) * 8;
        int rc;
        // scratch reuse: out_pid <- out_partition, out_rank <- member ranks, pid <- grouped_partition,
        // cons_rank <- grouped_topic, out_total <- member_off
        if ((rc = reserve(ctx, ctx->part_off, tb + 16)) || (rc = reserve(ctx, ctx->out_pid, nb4 + 16)) ||
            (rc = reserve(ctx, ctx->out_rank, nb4 + 16)) || (rc = reserve(ctx, ctx->pid, nb4 + 16)) ||
            (rc = reserve(ctx, ctx->cons_rank, nb4 + 16)) || (rc = reserve(ctx, ctx->out_total, mb + 16)))
            return rc;
        hipStream_t st = ctx->stream;
        if (n_topics > 0) LA_HIP(ctx, hipMemcpyAsync(ctx->part_off.p, part_off, tb, hipMemcpyHostToDevice, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(ctx->out_pid.p, out_partition, nb4, hipMemcpyHostToDevice, st));
            LA_HIP(ctx, hipMemcpyAsync(ctx->out_rank.p, out_member_rank, nb4, hipMemcpyHostToDevice, st));
        }
        rc = la_group_by_member_device(ctx, n_topics, n, (const int64_t*)ctx->part_off.p, (const int32_t*)ctx->out_pid.p,
                                       (const int32_t*)ctx->out_rank.p, n_members, (int64_t*)ctx->out_total.p,
                                       grouped_topic ? (int32_t*)ctx->cons_rank.p : nullptr, (int32_t*)ctx->pid.p, st);
        if (rc) return rc;
        LA_HIP(ctx, hipMemcpyAsync(member_off, ctx->out_total.p, mb, hipMemcpyDeviceToHost, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(grouped_partition, ctx->pid.p, nb4, hipMemcpyDeviceToHost, st));
            if (grouped_topic) LA_HIP(ctx, hipMemcpyAsync(grouped_topic, ctx->cons_rank.p, nb4, hipMemcpyDeviceToHost, st));
        }
        LA_HIP(ctx, hipStreamSynchronize(st));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_by_member");
    }
}

LA_API int la_group_last_by_member(la_ctx* ctx, int32_t n_members, int64_t* member_off, int32_t* grouped_topic,
                                   int32_t* grouped_partition) {
    if (!ctx) return LA_EINVAL;
    try {
        if (!ctx->last_valid)
            return fail(ctx, LA_EINVAL, "no result of la_assign_batch / la_assign_batch_lags is held on the device");
        if (n_members < 0) return fail(ctx, LA_EINVAL, "negative size");
        const int64_t n = ctx->last_n;
        if (!member_off || (n > 0 && !grouped_partition)) return fail(ctx, LA_EINVAL, "null buffer");
        LA_HIP(ctx, hipSetDevice(ctx->device));
        const size_t nb4 = (size_t)n * 4, mb = ((size_t)n_members + 1) * 8;
        int rc;
        // part_off, out_pid and out_rank hold the batch and its results; pid <- grouped_partition,
        // cons_rank <- grouped_topic, out_total <- member_off (their old contents are no longer needed)
        if ((rc = reserve(ctx, ctx->pid, nb4 + 16)) || (rc = reserve(ctx, ctx->cons_rank, nb4 + 16)) ||
            (rc = reserve(ctx, ctx->out_total, mb + 16)))
            return rc;
        hipStream_t st = ctx->stream;
        rc = la_group_by_member_device(ctx, ctx->last_topics, n, (const int64_t*)ctx->part_off.p,
                                       (const int32_t*)ctx->out_pid.p, (const int32_t*)ctx->out_rank.p, n_members,
                                       (int64_t*)ctx->out_total.p, grouped_topic ? (int32_t*)ctx->cons_rank.p : nullptr,
                                       (int32_t*)ctx->pid.p, st);
        if (rc) return rc;
        LA_HIP(ctx, hipMemcpyAsync(member_off, ctx->out_total.p, mb, hipMemcpyDeviceToHost, st));
        if (n) {
            LA_HIP(ctx, hipMemcpyAsync(grouped_partition, ctx->pid.p, nb4, hipMemcpyDeviceToHost, st));
            if (grouped_topic) LA_HIP(ctx, hipMemcpyAsync(grouped_topic, ctx->cons_rank.p, nb4, hipMemcpyDeviceToHost, st));
        }
        LA_HIP(ctx, hipStreamSynchronize(st));
        return LA_OK;
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_group_last_by_member");
    }
}

LA_API int la_sync(la_ctx* ctx, void* stream) {
    if (!ctx) return LA_EINVAL;
    try {
        LA_HIP(ctx, hipSetDevice(ctx->device));
        return sync_status(ctx, (hipStream_t)stream);
    } catch (...) {
        return fail(ctx, LA_ENOMEM, "exception in la_sync");
    }
}
