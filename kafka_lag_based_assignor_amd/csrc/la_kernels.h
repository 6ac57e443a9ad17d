// la_kernels.h -- host-visible declarations of the kernel launchers (internal to the library).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace la {

// Kernel launches of the whole library, counted where they are issued: la_last_launches (lagassign.h) reports the difference
// over one call -- how a test asserts "one launch" for a batch whose bounds prove that every tile packs.  Process-wide (the
// lanes of a host-buffer call launch from their own threads), relaxed: a diagnostic, not a synchronisation point.
inline std::atomic<uint64_t> g_kernel_launches{0};
#define LA_LAUNCH(...)                                                        \
    do {                                                                      \
        ::la::g_kernel_launches.fetch_add(1, std::memory_order_relaxed);      \
        hipLaunchKernelGGL(__VA_ARGS__);                                      \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device, and a process may hold contexts
// on several: one bit per device id remembers where a kernel has been opted in (ids >= 32: set on every launch).
// Calls may come from several host threads (one per shard lane), hence the atomic.
struct PerDeviceOnce {
    std::atomic<uint32_t> mask{0};
    template <typename F>
    hipError_t run(F&& set) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint32_t bit = (dev >= 0 && dev < 32) ? (1u << dev) : 0u;
        if (bit && (mask.load(std::memory_order_acquire) & bit)) return hipSuccess;
        if ((e = set()) != hipSuccess) return e;
        if (bit) mask.fetch_or(bit, std::memory_order_release);
        return hipSuccess;
    }
};

// device status word bits (la_ctx::d_status)
constexpr uint32_t kStatusShape = 1u;      // a topic exceeded the shape hint
constexpr uint32_t kStatusUnsorted = 2u;   // a topic's cons_rank segment is not strictly ascending
constexpr uint32_t kStatusInternal = 4u;   // a look-back walk of the radix sort gave up waiting (never expected: a bounded spin
                                           // exists so that a scheduling surprise ends as an error, not as a hung device)

constexpr uint32_t kStatusOrder = 8u;      // the radix sort's result is not in order (emit_ids_kernel / member_emit_kernel compare neighbours
                                           // on their way through): the lane-order property the atomic ranking relies on did not
                                           // hold under load.  Never expected; a checked error instead of a silently wrong assignment

constexpr uint32_t kStatusWire = 16u;      // la_pack_results_on: a partition id or member rank does not fit the wire format it was given

constexpr uint32_t kStatusSparse = 32u;    // la_assign_batch_sparse: none_index is not ascending, or points outside the batch

constexpr uint32_t kStatusBounds = 64u;    // LA_FLAG_BOUNDS: a lag or a partition id lies outside the bounds the caller guaranteed

constexpr int32_t kTileNoDefer = 8;       // TileArgs::flags: the bounds prove that every tile packs -- no deferred list, no wide
                                          // launch; a tile that does not pack after all raises kStatusBounds
constexpr int32_t kTileWireOut = 16;      // TileArgs::flags: out_wire instead of out_pid / out_rank (needs kTileNoDefer)
constexpr int32_t kTileSkipOversize = 4;  // TileArgs::flags: topics beyond the tile belong to another path, no error
constexpr int32_t kTileAlwaysStage2 = 32; // TileArgs::flags: (lab hook, LA_TILE_ALWAYS_STAGE2) never skip the second-stage `begin` loads

constexpr int64_t kTileMaxPartitions = 1024;   // 64 lanes x 16 records
constexpr int64_t kTileMaxConsumers = 64;      // one consumer bin per lane
constexpr int64_t kLargeMaxConsumers = 8192;   // large path: 8 bins per thread x 1024 threads

// The tail of a small rebalance's ONE launch (tile kernel, single-launch form: the whole batch resident at once).  Every
// workgroup counts itself done on `counter`; the last one builds every member's list from the results the others wrote
// (group_small_body, la_group_small.h; skipped when member_off is null: an ungrouped call) and stores `done | status` where the
// calling thread is spinning -- assignment, lists and completion in one launch instead of three (Main.java:147-156).
struct TileTail {
    int32_t enabled;            // 0: no tail (every launch but a zero-copy small call's)
    int32_t n_members;
    int32_t n;                  // entries (partitions of the batch), <= kSmallGroupN when the lists are wanted
    int32_t pad_;
    int64_t n_topics;
    const int64_t* part_off;    // of the WHOLE batch, from topic 0 (what grouped_topic indexes)
    const int32_t* out_pid;     // the launch's own results
    const int32_t* out_rank;
    int64_t* member_off;        // null: no lists
    int32_t* grouped_topic;     // may be null
    int32_t* grouped_partition;
    uint32_t* counter;          // device word, zero between calls (the last workgroup resets it)
    uint32_t* fin_flag;         // host word (coherent, mapped)
};

// Arguments of the fused wave-tile kernel (all device pointers).
struct TileArgs {
    int64_t n_topics;
    int64_t n_total;            // partitions in the whole batch (loads are clamped to it)
    int64_t k_total;            // consumer entries in the whole batch
    const int64_t* part_off;
    const int32_t* pid;
    const int64_t* begin;       // may be null (treated as 0; only read when !reset_latest)
    const int64_t* end;
    const int64_t* committed;
    const int64_t* lag;         // non-null: precomputed lags, offsets ignored
    const int64_t* cons_off;
    const int32_t* cons_rank;
    int32_t* out_pid;
    int32_t* out_rank;
    int64_t* out_total;         // may be null
    uint32_t* status;
    int32_t reset_latest;
    int32_t flags;              // LA_FLAG_* of the batch (test hooks) | kTileSkipOversize
    // tiles the packed kernel leaves to the wide kernel (LA_ALGO_AUTO): a counter pair that alternates
    // per launch (the wide kernel zeroes the other one), and the list of tile ids
    // non-null: slot s of the launch is topic topic_list[s] (shape classes of a ragged batch); null: topic s
    const int32_t* topic_list;
    int32_t* defer_count;
    int32_t* defer_count_next;
    int32_t* defer_list;
    TileTail tail;              // honoured by the single-launch form only (wave_tile_launch says whether it was)
    // LA_FLAG_WIRE_OUT: results as wire elements ((rank + 1) << wire_id_bits) | id of wire_bytes (2 or 4) each instead of the
    // two int32 arrays; only with kTileNoDefer (one launch of the packed-record kernel, 32-bit indexing)
    void* out_wire;
    int32_t wire_bytes, wire_id_bits;
};

// bytes of defer_list a launch over n_topics topics may need (one tile holds >= 1 topic)
inline size_t wave_tile_defer_bytes(int64_t n_topics) { return (size_t)(n_topics > 0 ? n_topics : 1) * sizeof(int32_t); }

inline bool wave_tile_fits(int64_t max_p, int64_t max_c) {
    return max_p <= kTileMaxPartitions && max_c <= kTileMaxConsumers;
}
void wave_tile_pick(int64_t max_p, int64_t max_c, int* L, int* E);
// true when ids in [0, max_id] and lags in [0, max_lag] pack into 64-bit records in every tile of the shape the launch would pick
bool wave_tile_always_packs(int64_t max_p, int64_t max_c, int64_t max_lag, int64_t max_id);
// mode: 0 = rounds, record format picked per wavefront; 1 = rounds, wide records forced; 2 = literal argmin
// *tail_done (may be null) <- true when a.tail.enabled and the launch was the single-launch form, whose last workgroup runs
// the tail; otherwise the caller finishes the call with its own launches.
hipError_t wave_tile_launch(TileArgs a, int64_t max_p, int64_t max_c, int mode, hipStream_t stream, bool* tail_done = nullptr);

// Elementwise lag: out_lag[i] = computePartitionLag(...)   (Main.java:376-404)
hipError_t lag_launch(int64_t n, const int64_t* begin, const int64_t* end, const int64_t* committed,
                      bool reset_latest, int64_t* out_lag, hipStream_t stream);

// begin[idx[j] - base] = val[j] for the m entries of a sparse begin list cut for positions [lo, hi); sets kStatusSparse.
hipError_t sparse_begin_launch(int64_t m, const int64_t* idx, const int64_t* val, int64_t base, int64_t lo, int64_t hi,
                               int64_t* begin, uint32_t* status, hipStream_t stream);

// The last launch of a zero-copy small call: *h_flag = 0x80000000 | *d_status (h_flag in coherent host memory).
hipError_t finish_status_launch(const uint32_t* d_status, uint32_t* h_flag, hipStream_t stream);
hipError_t wake_launch(hipStream_t stream);                  // one empty kernel (la_wake, for the streams its dummy call does not use)

// Checks that every topic's cons_rank segment is strictly ascending; sets kStatusUnsorted.
hipError_t check_consumers_launch(int64_t n_topics, const int64_t* cons_off, const int32_t* cons_rank,
                                  uint32_t* status, hipStream_t stream);

// ---- block path (la_block.hip): one workgroup per topic, everything in LDS --------------------
constexpr int64_t kBlockMaxPartitions = 8192;    // with up to kBlockMaxConsumers consumers
constexpr int64_t kBlockMaxConsumers = 2048;
constexpr int64_t kBlockWidePartitions = 16384;  // with up to kBlockWideConsumers consumers (16 records per thread)
constexpr int64_t kBlockWideConsumers = 1024;
constexpr int kBlockClasses = 5;               // LDS / workgroup size classes, see block_class()

struct BlockArgs {
    const int64_t* part_off;    // the batch's offsets (device)
    const int64_t* cons_off;
    const int32_t* list;        // topic indices this launch handles (device); null: inline_list
    int32_t n_list;
    int32_t inline_list[8];     // a handful of topics travel in the kernel arguments: no list copy
    const int32_t* pid;
    const int64_t* begin;
    const int64_t* end;
    const int64_t* committed;
    const int64_t* lag;
    const int32_t* cons_rank;
    int32_t* out_pid;
    int32_t* out_rank;
    int64_t* out_total;
    uint32_t* status;
    int32_t reset_latest;
    int32_t np_cap, nc_cap;     // LDS capacity in records / bins, set by the launcher
    int32_t dense_ids;          // set by the launcher: ids that are a permutation of 0 .. P-1 are placed by ONE scatter, not by digit passes
    int32_t key32_greedy;       // set by the launcher: 65 .. 256 consumers with packed totals take greedy_one_wave_key32
    int32_t radix_sort;         // set by the launcher: records that pack into one word are sorted by the workgroup's LSD radix
                                // sort (block_sort_radix) -- needs ranks from returning LDS atomics (large_atomic_rank_supported)
};

inline bool block_fits(int64_t p, int64_t c) {
    return (p <= kBlockMaxPartitions && c <= kBlockMaxConsumers) || (p <= kBlockWidePartitions && c <= kBlockWideConsumers);
}
inline int block_class(int64_t p, int64_t c) {
    if (p <= 512 && c <= 256) return 0;
    if (p <= 2048 && c <= 256) return 1;
    if (p <= 4096 && c <= 1024) return 2;
    if (p <= kBlockMaxPartitions) return 3;
    return 4;
}
hipError_t block_launch(BlockArgs a, int cls, hipStream_t stream);

// ---- large-topic path (device-wide radix sort + one-workgroup greedy) ---------------------
// LA_FLAG_PROFILE: HIP events around the phases of the first large topic of a call (la_last_phase_times)
struct LargeProfile {
    hipEvent_t ev[4] = {};      // before the keys kernel, after the plan, after the last pass, after the greedy
    bool armed = false;         // the current call asked for it
    bool recorded = false;      // ev[] hold a topic's phases
    int64_t n = 0;              // its partitions
};

struct LargeScratch {
    void* buf = nullptr;
    size_t cap = 0;
    LargeProfile prof;
    // several large topics side by side (large_topics_launch): their argument blocks are built in a pinned slot (two, used in
    // turn: a slot is rewritten only after the copy that read it has completed) and copied to d_items on the call's stream
    struct Stage {
        void* h = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;
    } stage[2];
    unsigned stage_next = 0;
    void* d_items = nullptr;
    size_t d_items_cap = 0;
};

struct LargeArgs {
    int64_t p0, n_part;         // partition segment of the topic
    int64_t c0, n_cons;         // consumer segment of the topic
    const int32_t* pid;
    const int64_t* begin;
    const int64_t* end;
    const int64_t* committed;
    const int64_t* lag;
    const int32_t* cons_rank;
    int32_t* out_pid;
    int32_t* out_rank;
    int64_t* out_total;
    uint32_t* status;
    int32_t reset_latest;
    int32_t no_sample_sort;     // 1 = LA_FLAG_NO_SAMPLE_SORT: every greedy round sorts its bins with the full network;
                                // 2 = LA_FLAG_SAMPLE_TIGHT: bucket limit 6, so sample-sorted and fallback rounds interleave
    int32_t no_run_merge;       // LA_FLAG_NO_RUN_MERGE: greedy rounds never merge ascending runs (they sort as if there were none)
    int32_t sort_multi_kernel;  // LA_FLAG_SORT_MULTIKERNEL: four kernels per radix pass (count, scans, scatter) instead of one
    int32_t no_moved_sort;      // LA_FLAG_NO_MOVED_SORT: greedy rounds never sort only the bins that move
    int64_t max_lag_hint;       // LA_FLAG_BOUNDS: 0 <= lag <= max_lag_hint and 0 <= id <= max_id_hint for every partition (the caller's
    int64_t max_id_hint;        // guarantee), or -1: radix passes over digits the bounds rule out are not even launched (round 6)
    int32_t rounds_follow;      // 1: greedy_rounds_kernel and map_ranks_kernel follow emit_ids_kernel -- the three agree on the narrow form of
                                // the rounds' lags and results (rounds_io, la_large.hip)
};

// Once per device at context creation (synchronous): checks the hardware property the radix sort's atomic ranking relies on.
hipError_t large_init_device();
int large_atomic_rank_supported();      // of the current device: 1 = ranks come from returning LDS atomics, 0 = match form
hipError_t large_topic_launch(LargeScratch& scratch, const LargeArgs& a, bool argmin, hipStream_t stream);
// `count` large topics (each with n_part > 0 and at most kLargeMaxConsumers consumers) SIDE BY SIDE: the per-topic loop of
// assign(Map,Map) (Main.java:177-184) is independent across topics, so every phase is one launch over all of them -- keys,
// plan, each radix pass (grid.y = topic), ids, one greedy workgroup per topic, ranks -- instead of ~15 launches and a
// one-CU chain per topic, one topic after another.  Falls back to that serial form for a single topic and for the
// four-kernel-pass test hook.
hipError_t large_topics_launch(LargeScratch& scratch, const LargeArgs* args, int count, hipStream_t stream);
// A topic with more than kLargeMaxConsumers consumers: bins in HBM, every round one stable device sort of the bins + one
// pass that hands the round's partitions out (ceil(P / C) rounds of ~11 launches).  Any C up to 2^31 - 1.
hipError_t huge_topic_launch(LargeScratch& scratch, const LargeArgs& a, hipStream_t stream);
void large_scratch_release(LargeScratch& scratch);
// Waits for the profiled topic's events; ms[3] = keys + plan, sort passes (+ tie repair), ids + greedy; passes[4] = active id
// passes, active key passes, 1 if the sort went keys first, 1 if it had to be redone in full.
hipError_t large_profile_read(LargeScratch& scratch, float* ms, int* passes, int64_t* n);

// Stable grouping of the n assignment entries by member rank (see la_group_by_member in lagassign.h).
hipError_t group_by_member_launch(LargeScratch& scratch, int64_t n, int32_t n_members, int64_t n_topics,
                                  const int64_t* part_off, const int32_t* out_partition, const int32_t* member_rank,
                                  int64_t* member_off, int32_t* grouped_topic, int32_t* grouped_partition,
                                  int32_t* grouped_entry, uint32_t* status, hipStream_t stream, uint32_t* fin_flag = nullptr,
                                  bool* fin_done = nullptr);
// (fin_flag: a zero-copy call's completion word in host memory; when the one-workgroup form runs it also finishes the call --
//  *fin_done says whether it did, otherwise the caller launches finish_status_launch)

// ---- narrow wire format of the all-gather (la_wire.hip) -----------------------------------------------------------
void wire_format_for(int64_t max_partition_id, int64_t n_members, int* elem_bytes, int* id_bits);     // pure host code
bool wire_format_valid(int elem_bytes, int id_bits);
hipError_t wire_pack_launch(int64_t n, const int32_t* pid, const int32_t* rank, int elem_bytes, int id_bits, void* out,
                            uint32_t* status, hipStream_t stream);
hipError_t wire_unpack_launch(int64_t n, const void* in, int elem_bytes, int id_bits, int32_t* pid, int32_t* rank,
                              hipStream_t stream);

}  // namespace la
