/*
 * lagassign.h -- C ABI of the MI355X-native lag-based partition assignor.
 *
 * This is the drop-in boundary: the Java host (a ConsumerPartitionAssignor with the
 * same configure()/name()/assign() surface as the reference) does everything that is
 * string- or container-shaped and calls these functions through a thin JNI shim
 * (see INTEGRATION.md).  Everything below is plain pointers and sizes.
 *
 * "Main.java:N" =
 *   reference src/main/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java:N
 *
 * What each entry point replaces:
 *   la_compute_lag          static long computePartitionLag(...)            Main.java:376-404
 *   la_assign_batch         readTopicPartitionLags' per-partition lag call  Main.java:344-356
 *                           + the per-topic loop of assign(Map,Map)         Main.java:177-184
 *                           + assignTopic (sort + greedy + update)          Main.java:204-266
 *   la_assign_batch_sparse  the same with beginning offsets only where they are read   Main.java:384-396
 *   la_assign_batch_lags    static assign(Map,Map) on precomputed lags      Main.java:166-188
 *   la_assign_batch_device  the same two, on buffers already resident in HBM
 *   la_group_by_member      building every member's List<TopicPartition>      Main.java:171-174, :264
 *   la_assign_batch_grouped la_assign_batch + la_group_last_by_member in one call: the body of
 *                           assign(Cluster, GroupSubscription) between the offset RPCs and the wrapping  Main.java:147-156
 *   la_create_multi         the per-topic loop, sharded over the GPUs of a node Main.java:177-184
 *   la_assign_batch_device_on  the same loop for a caller whose data already lives on every GPU
 *   la_plan_shards          (which topics of that loop each shard takes)
 *   la_allgather_results    nothing in the reference (it has one thread and one heap): the reassembly of the global
 *                           assignment on every GPU, one RCCL all-gather over xGMI
 *   la_pack_results_on, la_unpack_results_on, la_allgather_packed, la_wire_format_for
 *                           the same gather in 2 (or 4) bytes per assigned partition instead of 8
 *   la_hint_next_call       nothing in the reference's arithmetic: what the marshalling loop of readTopicPartitionLags
 *                           (Main.java:344-356) knows for free -- the largest end offset and partition id it walked past
 *   la_last_phase_times     nothing: measurement hook (radix-sort phase against the HBM roofline)
 *   la_device_features, la_last_pipeline, la_last_launches   nothing: diagnostics (what the library found / did)
 *   la_wake                       nothing in the reference: the head start its RPC phase (Main.java:147) gives the device
 *
 * Data model (SoA; TopicPartitionLag, Main.java:431-455, flattened):
 *   topic t owns partitions [part_off[t], part_off[t+1]) of the per-partition arrays and
 *   consumers  [cons_off[t], cons_off[t+1]) of cons_rank.
 *
 *   cons_rank[k] is the subscribing member's rank under java.lang.String.compareTo over
 *   all memberIds (0 = smallest).  Within one topic's segment the ranks MUST be strictly
 *   ascending (sorted, de-duplicated); the host does this once per rebalance.  The
 *   device never sees strings: "lowest memberId" (Main.java:259) == "lowest rank".
 *
 * Results, per topic segment, in the order the reference appends them (Main.java:264):
 *   out_partition[i]    partition id of the i-th assignment (lag desc, id asc)
 *   out_member_rank[i]  rank of the member that received it (-1: topic has no consumers,
 *                       Main.java:211-213)
 *   out_total_lag[k]    final consumerTotalLags of consumer k of that topic
 *                       (Main.java:265, the value the debug summary prints, :283-291)
 *
 * Arithmetic is Java's: 64-bit two's-complement wrap on subtract/add, signed compares.
 * Results are bit-identical to the reference for every input, including negative lags
 * and overflowing totals.
 *
 * Threading: a la_ctx is single-threaded for its CALLER (one per assignor instance, like
 * the reference's own non-thread-safe state, Main.java:89); inside a host-buffer call the
 * library runs one host thread per lane of every shard (threads of the context, parked between calls; since ABI 0.4.0 -- spawned and
 * joined per call before) and returns when they are done.  No global mutable state.
 * Errors: every function returns LA_OK or a negative code and never throws or aborts;
 * la_last_error() gives the text.  There is NO CPU fallback in this library: without a
 * usable gfx950 device la_create fails and the caller (the Java host) decides.
 */
#ifndef LAGASSIGN_H
#define LAGASSIGN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LA_OK        0
#define LA_EINVAL   (-1)  /* bad argument (NULL, negative size, unsorted cons_rank, ...) */
#define LA_ENOMEM   (-2)  /* host or device allocation failed                           */
#define LA_EHIP     (-3)  /* a HIP runtime call or kernel failed                        */
#define LA_ENODEV   (-4)  /* no usable gfx950 device                                    */
#define LA_ESHAPE   (-5)  /* a topic exceeds the shape hint given for a device batch    */

/* auto.offset.reset as the reference reads it (Main.java:391-396): "latest"
 * (equalsIgnoreCase) -> LA_RESET_LATEST; EVERY other string, "none" included, behaves
 * as earliest. */
#define LA_RESET_LATEST   0
#define LA_RESET_EARLIEST 1

/* committed_off value meaning "no committed offset" (partitionMetadata == null,
 * Main.java:384).  Any negative value is treated the same; OffsetAndMetadata cannot
 * hold one. */
#define LA_NO_COMMITTED (-1)

/* la_device_batch.algo */
#define LA_ALGO_AUTO    0  /* round-structured greedy (default, fastest)                */
#define LA_ALGO_ROUNDS  1  /* force the round-structured greedy                         */
#define LA_ALGO_ARGMIN  2  /* literal per-partition wavefront argmin over the bins      */
#define LA_ALGO_ROUNDS_WIDE 3 /* rounds, but never the packed 64-bit record format (tile path);
                              * results are identical, this exists so tests can run both  */

/* la_device_batch.flags -- test hooks; results are identical with or without them */
#define LA_FLAG_INDEX64      1  /* 64-bit element indexing even when 32-bit offsets would do   */
#define LA_FLAG_DEFER_WIDE   2  /* never the single-launch form for small batches: tiles that  *
                                 * cannot use packed records go through the deferred-tile list */
#define LA_FLAG_RAGGED       4  /* topic shapes vary a lot: with h_part_off / h_cons_off given, let the  *
                                 * library group tile-sized topics by shape (one launch per class over  *
                                 * a topic list) instead of running all of them at the largest shape.   *
                                 * la_assign_batch decides this itself from the offsets.                */
#define LA_FLAG_SHAPE_CLASSES 8 /* with LA_FLAG_RAGGED: always one launch per non-empty shape class,     *
                                 * whatever the cost estimate says (test hook)                          */
#define LA_FLAG_PROFILE      16 /* measurement hook: HIP events around the phases of the first large-path  *
                                 * topic of this call (keys, radix-sort passes, greedy); la_last_phase_times *
                                 * reads them.  Device entry point only.                                  */
#define LA_FLAG_NO_SAMPLE_SORT 32 /* large path: every greedy round sorts its consumer bins with the full   *
                                 * bitonic network instead of the sample sort (test hook)                 */
#define LA_FLAG_SAMPLE_TIGHT 64 /* large path: the sample sort gives a round up (falls back to the full network)  *
                                 * above 6 bins per bucket instead of 96, so both kinds of round interleave  *
                                 * in one topic (test hook)                                                */

#define LA_FLAG_SORT_MULTIKERNEL 128 /* large path: every radix pass as four kernels (tile counts, two scans, scatter)   *
                                 * instead of the single-kernel pass with decoupled look-back (test hook / A-B)         */

#define LA_FLAG_NO_RUN_MERGE 256 /* large path: greedy rounds whose bins form a few ascending runs sort them like any other   *
                                 * round instead of merging the runs (test hook / A-B)                                  */

#define LA_FLAG_NO_MOVED_SORT 2048 /* large path: greedy rounds in which few bins change places sort all of them like any other    *
                                  * round instead of only the bins that move (test hook / A-B)                            */

#define LA_FLAG_BOUNDS      1024 /* max_lag_hint and max_partition_id_hint below are valid: the caller guarantees 0 <= lag <=          *
                                 * max_lag_hint for every lag the batch produces (the largest end offset will do: a lag never exceeds    *
                                 * it) and 0 <= partition id <= max_partition_id_hint.  When the bounds PROVE that every tile's records  *
                                 * pack into 64 bits, the tile path drops its second launch (the wide-record kernel over the list of     *
                                 * deferred tiles, empty in that case): one launch per batch instead of two.  On the large path (since   *
                                 * round 6) the radix passes over digits the bounds rule out -- key digits above the largest lag's bits, *
                                 * id digits above the largest id's -- are not launched.  A partition that violates the bounds is        *
                                 * reported by la_sync as LA_EINVAL -- never a silently different result.                                */
#define LA_FLAG_WIRE_OUT    4096 /* d_out_wire / wire_elem_bytes / wire_id_bits below are valid (since ABI 0.4.0): the results leave the kernels     *
                                 * in the narrow wire format of the multi-GPU all-gather (la_wire_format_for, below) -- one element of 2 or 4 bytes  *
                                 * per assigned partition, ((member rank + 1) << id_bits) | partition id, in assignment order -- INSTEAD of the two   *
                                 * int32 arrays (d_out_partition / d_out_member_rank are not written and may be NULL): what la_pack_results_on        *
                                 * would produce from them, without the 8 B written and read again per partition.  For the batches the gather is     *
                                 * for: every topic tile-sized (shape hint within 1024 x 64, no LA_FLAG_RAGGED), LA_FLAG_BOUNDS proving that every    *
                                 * tile packs, fewer than 2^29 partitions; anything else is LA_EINVAL (run the batch without the flag and pack).     *
                                 * A pair that does not fit the format is reported by la_sync as LA_EINVAL.  d_out_total_lag is written as usual.     */
#define LA_FLAG_SERIAL_LARGE 512 /* large path: the batch's large topics one after another (round 3's form) instead of side by   *
                                 * side in shared launches (test hook / A-B)                                            */

typedef struct la_ctx la_ctx;

/* la_create / la_create_multi flags */
#define LA_CREATE_LANES_MASK   0xFu   /* low 4 bits: lanes (streams + host threads) per shard for the chunked    *
                                       * H2D / kernels / D2H overlap of the host-buffer calls; 0 = automatic      *
                                       * (3 for one or two shards, 2 beyond; environment LA_LANES overrides)      */
#define LA_CREATE_SPLIT_ALWAYS 0x10u  /* test hook: shard and chunk every batch, however small (normally a batch  *
                                       * under 65 536 partitions per extra shard stays on fewer devices and a     *
                                       * shard under 2 chunks of 524 288 partitions is one chunk, no threads)     */

/* Number of HIP devices visible to the process (>= 0), or a negative code. */
int la_device_count(void);

/* Creates a context on HIP device `device_id` (streams, scratch, kernels). */
int la_create(la_ctx **out, int device_id, unsigned flags);

/* Creates a context over several devices of one node: shard i of every host-buffer call runs on
 * device_ids[i].  This is the multi-GPU form of the per-topic loop of assign(Map,Map) (Main.java:177-184):
 * assignTopic touches only its own topic's bins (Main.java:216-225), so a batch is split into n_devices
 * contiguous topic ranges balanced by partition count (la_plan_shards), every range runs the whole hot path on
 * its device concurrently (one host thread per lane), and each shard's results are copied straight to their
 * offset in the caller's output arrays -- that IS the reassembly of the global assignment; no device ever needs
 * another device's topics.  n_devices = 0 (device_ids NULL) = every device of the node.  An id may appear more than
 * once (several logical shards on one GPU; how the tests exercise the split on a one-GPU box).
 * The device-buffer entry points without a shard argument (la_assign_batch_device, la_group_by_member_device, la_sync,
 * la_stream) use the first device; their *_on forms take any shard.  la_compute_lag and la_group_by_member split large
 * inputs over the shards like the assign calls do. */
int la_create_multi(la_ctx **out, int n_devices, const int *device_ids, unsigned flags);
void la_destroy(la_ctx *ctx);

/* Shards of the context, and the HIP device of shard i. */
int la_shard_count(const la_ctx *ctx);
int la_shard_device(const la_ctx *ctx, int shard);

/* What the library found out about shard i's device when the context was created (bit mask; measurement / diagnostics):
 *   LA_FEATURE_ATOMIC_RANK  the radix sort of the large path ranks equal digits with returning LDS atomics -- the device
 *                           passed the lane-order check that form relies on (otherwise: wave-match ranking, same results). */
#define LA_FEATURE_ATOMIC_RANK 1
int la_device_features(const la_ctx *ctx, int shard);

/* The planner the library uses for shards and for the chunks inside a shard: contiguous topic ranges
 * [bounds[r], bounds[r+1]) for r in [0, n_shards), balanced by partition count (bounds[r] = first topic boundary
 * at or after r/n_shards of the partitions).  Every topic lands in exactly one range; ranges may be empty.
 * bounds has n_shards + 1 entries.  Pure host code: needs no context and no device, so a multi-process
 * launcher (one process per GPU) can shard a batch exactly as a multi-device context would. */
int la_plan_shards(int32_t n_topics, const int64_t *part_off, int32_t n_shards, int32_t *bounds);

/* The split the last la_assign_batch / la_assign_batch_lags call used: returns the number of shards S and writes
 * min(S + 1, capacity) bounds (bounds may be NULL). */
int la_last_shard_bounds(const la_ctx *ctx, int32_t *bounds, int32_t capacity);

/* How the last la_assign_batch / la_assign_batch_lags call moved its data (diagnostics, tests):
 *   LA_PIPELINE_ZERO_COPY the call a real rebalance is: kernels read and write host memory in place, no copy, no stream wait
 *   LA_PIPELINE_ONE_COPY  a small batch: one H2D and one D2H of a staging buffer
 *   LA_PIPELINE_LANES     chunks over the shard's lanes, one host thread of the context per lane (pageable caller arrays:
 *                         their copies block the issuing thread)
 *   LA_PIPELINE_STREAMS   every array of the call is pinned but not device-mapped (hipHostRegister without the mapped flag;
 *                         LA_NO_MAPPED_PIPELINE=1): no threads; all H2D copies in order on one stream, kernels on a second, D2H
 *                         copies on a third, chained per chunk by events -- the input link stays busy from the first byte to the last
 *   LA_PIPELINE_MAPPED    every array is pinned and mapped (la_host_alloc): the kernels work on the caller's arrays in place */
#define LA_PIPELINE_ONE_COPY 0
#define LA_PIPELINE_LANES    1
#define LA_PIPELINE_STREAMS  2
#define LA_PIPELINE_MAPPED    4   /* every array of the call is pinned AND device-mapped (la_host_alloc): no copies and no chunks --
                                   * the kernels read the caller's arrays in place over PCIe (each input byte is touched once,
                                   * `begin` only where there is no committed offset) and write the results straight into them */
#define LA_PIPELINE_ZERO_COPY 3   /* calls whose arrays fit one staging buffer (up to 12 MB, ~340 000 partitions; for pinned
                                   * caller arrays less, see LA_PIPELINE_MAPPED; environment LA_ZERO_COPY_BYTES overrides,
                                   * 0 = never): no copy at all -- the inputs are packed into coherent, device-mapped host
                                   * memory, the kernels read them there and write totals / results / member lists into it;
                                   * the call's last launch stores the status where the calling thread is spinning */
int la_last_pipeline(const la_ctx *ctx);

/* Pinned host memory for the arrays handed to the host-buffer calls (a JNI shim wraps it in a direct ByteBuffer
 * with NewDirectByteBuffer): the copies then run as plain DMA, without the runtime staging or pinning pageable
 * pages per call.  Optional -- every entry point takes pageable memory too.  NULL on failure. */
void *la_host_alloc(la_ctx *ctx, size_t bytes);
void la_host_free(la_ctx *ctx, void *p);
/* Text of the last error on this context ("" if none).  ctx may be NULL: returns the
 * text of the last la_create failure on the calling thread. */
const char *la_last_error(const la_ctx *ctx);
/* ABI version: major*10000 + minor*100 + patch.  LA_VERSION is the header's, la_version() the loaded library's: a shim
 * (JNI, ctypes) compares the two before it binds entry points that older libraries lack.
 *   0.2.1 (201)  rounds 2-3: multi-device contexts, grouped calls, *_on device entry points, la_allgather_results
 *   0.3.0 (300)  round 4: la_wire_format_for / la_pack_results_on / la_unpack_results_on / la_allgather_packed (narrow wire
 *                format of the all-gather), la_assign_batch_sparse / la_assign_batch_grouped_sparse (begin offsets only
 *                where there is no committed offset); every entry point restores the caller's current HIP device
 *   0.4.0 (400)  round 5: la_hint_next_call (the caller's bounds on lags and ids reach the host-buffer calls: one tile launch
 *                instead of two), la_last_launches, la_last_phase_times_sized
 *   0.5.0 (500)  round 6: la_wake (the device's queues woken while the host still fetches offsets); LA_FLAG values unchanged */
#define LA_VERSION 500
int la_version(void);

/* computePartitionLag over n partitions (host buffers).  begin_off may be NULL when
 * reset_mode == LA_RESET_LATEST.  Main.java:376-404.
 * Reuses the device scratch the last assign call's results live in: after it (and after la_group_by_member)
 * la_group_last_by_member returns LA_EINVAL until the next la_assign_batch / la_assign_batch_lags. */
int la_compute_lag(la_ctx *ctx, int64_t n,
                   const int64_t *begin_off, const int64_t *end_off,
                   const int64_t *committed_off, int32_t reset_mode,
                   int64_t *out_lag);

/* What the caller's marshalling loop knows about the NEXT host-buffer assign call on this context (la_assign_batch, _sparse,
 * _lags, _grouped, _grouped_sparse) -- the reference's loop (Main.java:344-356) walks every partition's offsets anyway, so the
 * largest end offset and the largest partition id cost it two compares per partition.  One-shot: the hints apply to the next
 * such call and are forgotten when it returns, whatever it returns (a call without la_hint_next_call before it has none).
 *   LA_HINT_BOUNDS  the caller guarantees 0 <= lag <= max_lag for every lag the call computes (the largest end offset will do
 *                   when no begin / end / committed offset of the batch is negative, which Kafka guarantees: a lag never
 *                   exceeds its end offset) and 0 <= partition id <= max_partition_id.  The library hands them to its kernels
 *                   as LA_FLAG_BOUNDS (below): where they prove that every tile's records pack into 64 bits, the tile path
 *                   is ONE launch per chunk instead of two.  A partition that violates the bounds makes the call fail with
 *                   LA_EINVAL -- never a silently different result; a caller that is not sure gives no hint.
 * struct_size = sizeof(la_call_hints) of the caller's header (the struct may grow).  hints == NULL clears pending hints. */
#define LA_HINT_BOUNDS 1
typedef struct la_call_hints {
    int32_t struct_size;
    int32_t flags;               /* LA_HINT_* */
    int64_t max_lag;             /* LA_HINT_BOUNDS */
    int64_t max_partition_id;    /* LA_HINT_BOUNDS */
} la_call_hints;
int la_hint_next_call(la_ctx *ctx, const la_call_hints *hints);

/* Warm the path of the assign call that is about to come (since ABI 0.5.0).  What it is for: Kafka calls assign() on the group
 * leader once per rebalance -- minutes apart -- and the reference's assign() spends its first milliseconds on broker round trips
 * (readTopicPartitionLags, Main.java:147, :317-365) before it has a single offset to hand over.  After a second of idle a small
 * call costs several times what it costs back to back (tools/cold_c.c at the C ABI on MI355X, profiles/r06_p_cold_c.txt: a
 * 100-partition la_assign_batch_grouped 23 us back to back, 36 us after 50 ms of idle, 85 us after 1 s; 2 000 partitions 32 / 46 /
 * 101 us): the device's queue, the link, the mapped pages, the runtime's code in the host's caches are all cold.  la_wake runs
 * ONE one-partition rebalance through the real small-call path and waits for it (it pays the cold call itself: ~95 us), and
 * launches one empty kernel on every other stream of the context; the assign call that follows within ~20 ms then takes 34 us
 * (100 partitions) / 48 us (2 000) / 73 us (10 000: 127 cold); 200 ms later most of the effect is gone, and calls of hundreds of
 * thousands of partitions gain nothing.  A host calls it when it ENTERS assign(), before its RPCs.  Optional, never needed for
 * correctness.  A pending
 * la_hint_next_call and the diagnostics of the last real call (la_last_pipeline, la_last_launches) survive it; results kept on
 * the device for la_group_last_by_member do not (LA_EINVAL until the next assign call, as after la_compute_lag). */
int la_wake(la_ctx *ctx);

/* Kernel launches the last host-buffer call (or the last la_assign_batch_device[_on] call) on this context enqueued -- every
 * kernel of the library counts, copies and memsets do not.  Diagnostics / tests: a batch of tile-sized topics whose hints prove
 * that every tile packs reports 1 per chunk (+ 1 per chunk for the consumer-rank check of the copying pipelines).  The counter
 * behind it is process-wide: exact while no other context of the process is inside a call. */
int64_t la_last_launches(const la_ctx *ctx);

/* One rebalance: lag from offsets, then sort + greedy per topic.  All pointers are
 * caller-owned host memory, valid for the duration of the call.  N = part_off[T],
 * K = cons_off[T].  out_total_lag may be NULL. */
int la_assign_batch(la_ctx *ctx, int32_t n_topics,
                    const int64_t *part_off,       /* [T+1]                              */
                    const int32_t *partition_id,   /* [N]                                */
                    const int64_t *begin_off,      /* [N]  NULL allowed iff LATEST       */
                    const int64_t *end_off,        /* [N]                                */
                    const int64_t *committed_off,  /* [N]  <0 == none                    */
                    int32_t reset_mode,
                    const int64_t *cons_off,       /* [T+1]                              */
                    const int32_t *cons_rank,      /* [K]  ascending within each topic   */
                    int32_t *out_partition,        /* [N]  (both NULL: results stay on   */
                    int32_t *out_member_rank,      /* [N]   the device, see la_group_last_by_member) */
                    int64_t *out_total_lag);       /* [K]  or NULL                       */

/* la_assign_batch with `begin` handed over SPARSELY.  computePartitionLag reads the beginning offset only where a partition
 * has no committed offset and auto.offset.reset is not "latest" (Main.java:384-396) -- typically ~1 % of a group's
 * partitions -- while a dense begin_off array is 8 of the 28 input bytes per partition that cross PCIe, the link that bounds
 * every host-buffer call.  Here the caller lists only those partitions:
 *   none_index[j]  position (index into the per-partition arrays, 0 .. N-1) of a partition without a committed offset,
 *                  ASCENDING (the marshaller meets them in order: GpuLagBasedPartitionAssignor.java, `md == null`)
 *   none_begin[j]  its beginning offset
 * A partition with committed_off < 0 that is NOT listed has begin 0 -- the reference's getOrDefault(tp, 0L) for a missing
 * beginning offset (Main.java:350-351).  Entries whose partition HAS a committed offset are harmless (never read).
 * In LA_RESET_LATEST mode the list is ignored (may be NULL).  A list that is not ascending or leaves [0, N): LA_EINVAL.
 * Results are those of la_assign_batch on the equivalent dense array, bit for bit. */
int la_assign_batch_sparse(la_ctx *ctx, int32_t n_topics,
                           const int64_t *part_off, const int32_t *partition_id,
                           const int64_t *end_off, const int64_t *committed_off, int32_t reset_mode,
                           int64_t n_none, const int64_t *none_index, const int64_t *none_begin,
                           const int64_t *cons_off, const int32_t *cons_rank,
                           int32_t *out_partition, int32_t *out_member_rank, int64_t *out_total_lag);

/* Same, on precomputed lags: the static assign(Map,Map) seam the reference's own tests
 * use (Test.java:127-128).  lag[] may hold any int64, negatives included. */
int la_assign_batch_lags(la_ctx *ctx, int32_t n_topics,
                         const int64_t *part_off, const int32_t *partition_id,
                         const int64_t *lag,
                         const int64_t *cons_off, const int32_t *cons_rank,
                         int32_t *out_partition, int32_t *out_member_rank,
                         int64_t *out_total_lag);

/* A batch whose bulk arrays already live in device memory (HBM). */
typedef struct la_device_batch {
    int32_t n_topics;
    int32_t reset_mode;
    int32_t algo;                    /* LA_ALGO_*                                        */
    int32_t flags;                   /* LA_FLAG_*; 0 in normal use                       */
    int64_t n_partitions;            /* N                                                */
    int64_t n_consumers;             /* K                                                */
    /* Shape hint: upper bounds over the batch.  A topic that exceeds them is reported
     * as LA_ESHAPE by la_sync(); nothing is written for it. */
    int64_t max_partitions_per_topic;
    int64_t max_consumers_per_topic;
    /* device pointers */
    const int64_t *d_part_off;       /* [T+1]                                            */
    const int32_t *d_partition_id;   /* [N]                                              */
    const int64_t *d_begin_off;      /* [N] or NULL (LATEST only)                        */
    const int64_t *d_end_off;        /* [N] (ignored when d_lag != NULL)                 */
    const int64_t *d_committed_off;  /* [N] (ignored when d_lag != NULL)                 */
    const int64_t *d_lag;            /* [N] or NULL: precomputed lags instead of offsets */
    const int64_t *d_cons_off;       /* [T+1]                                            */
    const int32_t *d_cons_rank;      /* [K]                                              */
    int32_t *d_out_partition;        /* [N]  (not looked at when N == 0)                 */
    int32_t *d_out_member_rank;      /* [N]                                              */
    int64_t *d_out_total_lag;        /* [K] or NULL                                      */
    /* host copies of the two offset arrays; required only when the shape hint exceeds
     * what one wavefront tile holds (1024 partitions or 64 consumers per topic): the
     * library then looks at every topic's size on the host (a few ns per topic) and
     * sends it down the wave-tile path, the block path (one workgroup per topic, up to
     * 8192 partitions x 2048 consumers, all such topics side by side) or the large path
     * (device-wide radix sort + one greedy workgroup per topic, the batch's large topics side by side in
     * shared launches; beyond 8192 consumers the bins live in HBM and every greedy round is a device
     * sort: any consumer count, Main.java:240-263 has no bound).  Also read when LA_FLAG_RAGGED is set.
     * NULL otherwise -- or NULL anyway: the library then fetches the offsets itself, which
     * makes that call wait on `stream` (not asynchronous, not capturable).  Calls on one context are expected to be stream-ordered: the
     * per-call topic lists live in context-owned device memory. */
    const int64_t *h_part_off;
    const int64_t *h_cons_off;
    /* with LA_FLAG_BOUNDS (since ABI 0.3.0; ignored without the flag) */
    int64_t max_lag_hint;            /* upper bound of every lag of the batch, >= 0                */
    int64_t max_partition_id_hint;   /* upper bound of every partition id, >= 0                    */
    /* with LA_FLAG_WIRE_OUT (since ABI 0.4.0; ignored without the flag) */
    void   *d_out_wire;              /* [N] elements of wire_elem_bytes each, element-aligned      */
    int32_t wire_elem_bytes;         /* 2 or 4 (la_wire_format.elem_bytes)                         */
    int32_t wire_id_bits;            /* la_wire_format.id_bits                                     */
} la_device_batch;

/* Enqueues the whole batch on `stream` and returns without waiting.  `stream` is a
 * hipStream_t used exactly as given (NULL = HIP's default stream); la_stream() gives
 * the context's own stream.  Device-detected errors surface in la_sync.
 * A batch of tile-sized topics (shape hint within 1024 x 64, no LA_FLAG_RAGGED) is kernel
 * launches only once the context has seen one call of that size (scratch allocated), so
 * it may be captured in a HIP graph and replayed; eager launches cost about the same. */
int la_assign_batch_device(la_ctx *ctx, const la_device_batch *batch, void *stream);

/* Waits for `stream` and returns LA_OK or the first device-detected error. */
int la_sync(la_ctx *ctx, void *stream);

/* The device entry points on ANY shard of a multi-device context: a caller that keeps its data in HBM drives all the
 * node's GPUs from one context (la_assign_batch_device / la_sync / la_stream are shard 0's).  The batch's pointers must
 * belong to la_shard_device(ctx, shard); `stream` is a stream of that device (la_shard_stream gives the shard's own).
 * Shards are independent -- their calls may be enqueued back to back from one host thread and run concurrently, which
 * is the per-topic loop of assign(Map,Map) (Main.java:177-184) over several devices with the data already resident:
 * split the topics with la_plan_shards, give shard i the range [bounds[i], bounds[i+1]), sync every shard. */
void *la_shard_stream(la_ctx *ctx, int shard);
int la_assign_batch_device_on(la_ctx *ctx, int shard, const la_device_batch *batch, void *stream);
int la_sync_on(la_ctx *ctx, int shard, void *stream);

/* Phase times of the first large-path topic (one topic beyond the block path: device-wide radix sort, then the
 * one-workgroup greedy) of the last la_assign_batch_device call that carried LA_FLAG_PROFILE.  Measurement only:
 * it is how bench.py reports the radix-sort phase against the HBM roofline from inside one run.  Waits for that
 * topic's kernels.  LA_EINVAL when no such topic was profiled. */
typedef struct la_phase_times {
    int64_t n_partitions;
    int32_t id_passes;     /* active 8-bit passes over the partition-id digits (0 when ids arrive ascending)  */
    int32_t key_passes;    /* active passes over the lag-key digits; constant digits are skipped on the device */
    float keys_ms;         /* lag + keys + the 12 digit histograms, pass plan                                  */
    float sort_ms;         /* every active radix pass (+ the tie repair of a keys-first sort): until the order is final */
    float greedy_ms;       /* ids in assignment order + the greedy rounds                                      */
    int32_t keys_first;    /* 1: the sort skipped the id passes and put runs of equal lags in id order afterwards
                            * (large topics with shuffled ids and no frequent lag; since ABI 0.3.0)             */
    int32_t redone;        /* 1: a run of equal lags did not fit the repair and the sort was redone in full     */
} la_phase_times;
int la_last_phase_times(la_ctx *ctx, la_phase_times *out);
/* The same for a caller compiled against any version of this header: writes min(out_size, sizeof(la_phase_times)) bytes.
 * la_last_phase_times writes the whole struct of THIS header (40 bytes since ABI 0.3.0, 32 before): a shim built against 0.2.x
 * must check la_version() < 300 before it calls that one, or call this one with its own sizeof. */
int la_last_phase_times_sized(la_ctx *ctx, void *out, size_t out_size);

/* The context's own (non-blocking) hipStream_t, used by the host-buffer entry points. */
void *la_stream(la_ctx *ctx);

/* Assignment -> per-member lists: the wrap step of the reference (every member's list is created at
 * Main.java:171-174 and appended to at :264, topic by topic in the order the topics were given, inside a
 * topic in assignment order; Main.java:152-156 then wraps each list).  Input: the result arrays of an
 * assign call.  Output, with M = n_members (ranks 0..M-1):
 *   member_off[r] .. member_off[r+1]   member r's slice of the grouped arrays (member_off has M+1 entries);
 *                                      positions before member_off[0] hold the entries of topics that had no
 *                                      consumer (rank -1), which the reference leaves unassigned
 *   grouped_topic[j], grouped_partition[j]   topic index (into part_off) and partition id of the j-th entry,
 *                                      in exactly the order the reference's list for that member holds them
 * It is a stable device radix sort of the entry indices by member rank.  grouped_topic may be NULL.
 * Like la_compute_lag this call takes over the scratch of the last assign call: the results held for
 * la_group_last_by_member are gone afterwards. */
int la_group_by_member(la_ctx *ctx, int32_t n_topics, const int64_t *part_off,
                       const int32_t *out_partition, const int32_t *out_member_rank, int32_t n_members,
                       int64_t *member_off, int32_t *grouped_topic, int32_t *grouped_partition);

/* The same for the results the last successful la_assign_batch / la_assign_batch_lags call on this context left
 * on the device: nothing is uploaded again.  A caller that only wants the grouped form gives that call
 * out_partition = out_member_rank = NULL (both), which also skips their download: the assignment then crosses
 * PCIe once, as member_off + grouped_topic + grouped_partition.  LA_EINVAL when there is no such result (no
 * assign call yet, or another host-buffer call on this context since).  grouped_topic may be NULL.
 * (After a call on pinned, device-mapped arrays -- LA_PIPELINE_MAPPED -- that DID take the ungrouped results, "on the device"
 * means the caller's own result arrays, which the kernels wrote in place: leave them as they are until this call.) */
int la_group_last_by_member(la_ctx *ctx, int32_t n_members,
                            int64_t *member_off, int32_t *grouped_topic, int32_t *grouped_partition);

/* Both steps in one call -- what the reference's assign(Cluster, GroupSubscription) does between reading the offsets
 * and wrapping the lists (Main.java:147-156): la_assign_batch with the ungrouped result left on the device, then
 * la_group_last_by_member.  For the call a real rebalance is (its arrays fit the library's staging buffer: 12 MB) the lists
 * are built behind the assignment kernels on the same stream -- up to 1 024 entries inside the assignment kernel itself -- and
 * land in that buffer with the status and the totals: no copy, one wait.  Larger batches run the two steps one after the other.  The results stay
 * on the device as after la_assign_batch (la_group_last_by_member may be called again).  grouped_topic and out_total_lag
 * may be NULL. */
int la_assign_batch_grouped(la_ctx *ctx, int32_t n_topics, const int64_t *part_off, const int32_t *partition_id,
                            const int64_t *begin_off, const int64_t *end_off, const int64_t *committed_off,
                            int32_t reset_mode, const int64_t *cons_off, const int32_t *cons_rank, int32_t n_members,
                            int64_t *member_off, int32_t *grouped_topic, int32_t *grouped_partition,
                            int64_t *out_total_lag);

/* la_assign_batch_grouped with the sparse `begin` of la_assign_batch_sparse: what the Java host calls. */
int la_assign_batch_grouped_sparse(la_ctx *ctx, int32_t n_topics, const int64_t *part_off, const int32_t *partition_id,
                                   const int64_t *end_off, const int64_t *committed_off, int32_t reset_mode,
                                   int64_t n_none, const int64_t *none_index, const int64_t *none_begin,
                                   const int64_t *cons_off, const int32_t *cons_rank, int32_t n_members,
                                   int64_t *member_off, int32_t *grouped_topic, int32_t *grouped_partition,
                                   int64_t *out_total_lag);

/* Same on device buffers (N = n_partitions entries); enqueues on `stream` and returns. */
int la_group_by_member_device(la_ctx *ctx, int32_t n_topics, int64_t n_partitions,
                              const int64_t *d_part_off, const int32_t *d_out_partition,
                              const int32_t *d_out_member_rank, int32_t n_members,
                              int64_t *d_member_off, int32_t *d_grouped_topic, int32_t *d_grouped_partition,
                              void *stream);

/* The north star's single RCCL all-gather, in native code, for a caller that drives every GPU of the node from ONE process
 * (la_create_multi + la_assign_batch_device_on): every shard's result buffer to every shard's device.
 *   d_send[i]  `count` int32 on shard i's device (e.g. its [2, cap] result buffer: partition order | member rank)
 *   d_recv[i]  n_shards * count int32 on shard i's device; shard j's block lands at d_recv[i] + j * count
 * One ncclAllGather per shard inside one ncclGroupStart / ncclGroupEnd, enqueued on la_shard_stream(ctx, i) -- i.e. behind
 * whatever la_assign_batch_device_on enqueued there; la_sync_on waits for it.  ncclAllGather moves equal counts: pad the
 * shards to the largest (la_plan_shards balances them to within one topic).  The communicators are created on first use
 * (ncclCommInitAll over the context's devices) and live as long as the context.
 * librccl is loaded at run time, on first use (dlopen; a copy already in the process is preferred): LA_ENODEV when it is
 * not there.  RCCL wants one DISTINCT device per rank: LA_EINVAL for a context with several shards on one device.
 * (One process per GPU -- bench.py, torch.distributed -- calls RCCL through its own framework instead.) */
int la_allgather_results(la_ctx *ctx, int64_t count, const int32_t *const *d_send, int32_t *const *d_recv);

/* ---- the narrow wire format of that all-gather ------------------------------------------------------------------------
 * What the gather moves per assigned partition is a (partition id, member) pair in assignment order (Main.java:264): 8 B as
 * two int32 arrays, but 8 + 6 bits of information at the 100 000 x 256 x 32 target.  xGMI is the slow station of the
 * multi-GPU step (7 links x ~77 GB/s per direction against ~5 TB/s of HBM), so the gather's bytes are the step's time.
 * One wire element =
 *       ((member rank + 1) << id_bits) | partition id        (rank + 1 == 0: the topic had no consumer, Main.java:211-213)
 * as an unsigned integer of elem_bytes = 2, 4 or 8 bytes; the 8-byte form (id_bits = 32) carries ANY int32 pair.
 *   la_wire_format_for   the narrowest format for ids in [0, max_partition_id] and ranks in [-1, n_members); pass
 *                        max_partition_id < 0 when ids may be negative or are not known (-> the 8-byte form).  Pure host code.
 *   la_pack_results_on   n results of shard `shard` (device arrays, e.g. what la_assign_batch_device_on wrote) -> n wire
 *                        elements; enqueued on `stream`.  A pair that does not fit `fmt` is reported by la_sync_on as LA_EINVAL.
 *   la_unpack_results_on the reverse, on any shard's device (e.g. over the whole gathered buffer, padding included).
 *   la_allgather_packed  la_allgather_results for `count` elements of elem_bytes each.
 * Pointers 16-byte aligned take the 16-byte-access kernels; any alignment of elem_bytes works. */
typedef struct la_wire_format {
    int32_t elem_bytes;    /* 2, 4 or 8 */
    int32_t id_bits;       /* low bits of an element that hold the partition id (32 in the 8-byte form) */
} la_wire_format;
int la_wire_format_for(int64_t max_partition_id, int64_t n_members, la_wire_format *out);
int la_pack_results_on(la_ctx *ctx, int shard, int64_t n, const int32_t *d_out_partition,
                       const int32_t *d_out_member_rank, const la_wire_format *fmt, void *d_packed, void *stream);
int la_unpack_results_on(la_ctx *ctx, int shard, int64_t n, const void *d_packed, const la_wire_format *fmt,
                         int32_t *d_out_partition, int32_t *d_out_member_rank, void *stream);
int la_allgather_packed(la_ctx *ctx, int64_t count, int32_t elem_bytes, const void *const *d_send, void *const *d_recv);

/* la_group_by_member_device on shard `shard` (buffers on that shard's device). */
int la_group_by_member_device_on(la_ctx *ctx, int shard, int32_t n_topics, int64_t n_partitions,
                                 const int64_t *d_part_off, const int32_t *d_out_partition,
                                 const int32_t *d_out_member_rank, int32_t n_members,
                                 int64_t *d_member_off, int32_t *d_grouped_topic, int32_t *d_grouped_partition,
                                 void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LAGASSIGN_H */
