package com.github.grantneale.kafka.gpu;

import java.nio.ByteBuffer;

/**
 * JNI face of include/lagassign.h (liblagassign.so).  NOT COMPILED in the image this repository is built in (it has
 * no JDK): java/run_reference_tests.sh builds and tests it wherever a JDK 8+ and the two jars exist.
 *
 * All buffers are DIRECT ByteBuffers in native byte order; the shim passes their addresses
 * straight to the C ABI (GetDirectBufferAddress), so nothing is copied on the Java side.
 */
final class LagAssignNative {

    static {
        System.loadLibrary("lagassign_jni");   // links liblagassign.so
    }

    static final int RESET_LATEST = 0;
    static final int RESET_EARLIEST = 1;
    static final long NO_COMMITTED = -1L;

    private LagAssignNative() { }

    /** la_device_count: HIP devices visible to the process (negative: the la_* error code). */
    static native int deviceCount();

    /**
     * la_create_multi over {@code deviceIds}, or over EVERY device of the node when it is null or empty: one shard per
     * device, a batch is split into contiguous topic ranges balanced by partition count and every shard's results land
     * at their offset in the caller's buffers.  Returns the context handle or throws
     * IllegalStateException(la_last_error).
     */
    static native long createMulti(int[] deviceIds);

    /** la_shard_count */
    static native int shardCount(long ctx);

    /** la_destroy */
    static native void destroy(long ctx);

    /**
     * la_host_alloc wrapped by NewDirectByteBuffer: {@code bytes} of pinned host memory (copies to and from it are
     * plain DMA).  The buffer must be returned with {@link #hostFree}; it is NOT garbage collected.  null on failure.
     */
    static native ByteBuffer hostAlloc(long ctx, long bytes);

    /** la_host_free */
    static native void hostFree(long ctx, ByteBuffer buffer);

    /** la_compute_lag: begin/end/committed/outLag int64[n]; begin may be null for RESET_LATEST. */
    static native int computeLag(long ctx, long n, ByteBuffer begin, ByteBuffer end, ByteBuffer committed,
                                 int resetMode, ByteBuffer outLag);

    /**
     * la_assign_batch.  partOff/consOff: int64[T+1]; partitionId/consRank: int32;
     * begin/end/committed: int64[N] (begin may be null for RESET_LATEST);
     * outPartition/outMemberRank: int32[N], or both null (results stay on the device for groupLastByMember);
     * outTotalLag: int64[K] or null.
     * Returns the la_* status code (0 = ok); lastError(ctx) has the text.
     */
    static native int assignBatch(long ctx, int nTopics, ByteBuffer partOff, ByteBuffer partitionId,
                                  ByteBuffer begin, ByteBuffer end, ByteBuffer committed, int resetMode,
                                  ByteBuffer consOff, ByteBuffer consRank, ByteBuffer outPartition,
                                  ByteBuffer outMemberRank, ByteBuffer outTotalLag);

    /**
     * la_assign_batch_grouped: assignBatch with the ungrouped result left on the device + groupLastByMember in ONE native
     * call -- for a rebalance of ordinary size one upload, one download, one wait.  Buffers as for those two calls;
     * groupedTopic and outTotalLag may be null.
     */
    static native int assignBatchGrouped(long ctx, int nTopics, ByteBuffer partOff, ByteBuffer partitionId,
                                         ByteBuffer begin, ByteBuffer end, ByteBuffer committed, int resetMode,
                                         ByteBuffer consOff, ByteBuffer consRank, int nMembers, ByteBuffer memberOff,
                                         ByteBuffer groupedTopic, ByteBuffer groupedPartition, ByteBuffer outTotalLag);

    /**
     * la_assign_batch_grouped_sparse: assignBatchGrouped with the beginning offsets handed over only where they are read --
     * for the {@code nNone} partitions without a committed offset (LagBasedPartitionAssignor.java:384-396 of the reference):
     * noneIndex int64[nNone] = their positions in the per-partition arrays, ascending; noneBegin int64[nNone] = their
     * beginning offsets.  8 of the 28 input bytes per partition no longer cross PCIe.  Needs la_version() &gt;= 300.
     */
    static native int assignBatchGroupedSparse(long ctx, int nTopics, ByteBuffer partOff, ByteBuffer partitionId,
                                               ByteBuffer end, ByteBuffer committed, int resetMode, long nNone,
                                               ByteBuffer noneIndex, ByteBuffer noneBegin, ByteBuffer consOff,
                                               ByteBuffer consRank, int nMembers, ByteBuffer memberOff,
                                               ByteBuffer groupedTopic, ByteBuffer groupedPartition, ByteBuffer outTotalLag);

    /**
     * la_hint_next_call(LA_HINT_BOUNDS): what the marshalling loop saw on its way -- every lag of the NEXT assign call on this
     * context is in [0, maxLag] (the largest end offset will do when no offset is negative) and every partition id in
     * [0, maxPartitionId].  One-shot.  The tile path then runs ONE launch per chunk instead of two; a partition outside the
     * bounds fails that call with LA_EINVAL.  Needs la_version() &gt;= 400.
     */
    static native int hintNextCallBounds(long ctx, long maxLag, long maxPartitionId);

    /** la_last_launches: kernel launches of the last call on this context (diagnostics). */
    static native long lastLaunches(long ctx);

    /** la_version of the loaded library: major * 10000 + minor * 100 + patch. */
    static native int version();

    /** la_assign_batch_lags: the static assign(Map,Map) seam, on precomputed lags (any int64). */
    static native int assignBatchLags(long ctx, int nTopics, ByteBuffer partOff, ByteBuffer partitionId,
                                      ByteBuffer lag, ByteBuffer consOff, ByteBuffer consRank,
                                      ByteBuffer outPartition, ByteBuffer outMemberRank, ByteBuffer outTotalLag);

    /**
     * la_group_last_by_member: every member's list in the reference's order, as slices, from the results the last
     * assignBatch / assignBatchLags call of this context left on the device (that call may be given outPartition =
     * outMemberRank = null, so the assignment crosses PCIe once).  memberOff: int64[M+1]; groupedTopic /
     * groupedPartition: int32[N].  Member r owns [memberOff[r], memberOff[r+1]); positions before memberOff[0] belong
     * to topics without consumers.
     */
    static native int groupLastByMember(long ctx, int nMembers, ByteBuffer memberOff, ByteBuffer groupedTopic,
                                        ByteBuffer groupedPartition);

    /** la_last_error */
    static native String lastError(long ctx);
}
