package com.github.grantneale.kafka.gpu;

import java.nio.ByteBuffer;

/**
 * JNI face of include/lagassign.h (liblagassign.so).  SOURCE ONLY: there is no JDK in the
 * build image, so this file and jni/lagassign_jni.c have not been compiled here.
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

    /** la_create; returns the context handle or throws IllegalStateException(la_last_error). */
    static native long create(int deviceId);

    /** la_destroy */
    static native void destroy(long ctx);

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
     * la_group_by_member: every member's list in the reference's order, as slices.
     * memberOff: int64[M+1]; groupedTopic/groupedPartition: int32[N].  Member r owns
     * [memberOff[r], memberOff[r+1]); positions before memberOff[0] belong to topics without consumers.
     */
    static native int groupByMember(long ctx, int nTopics, ByteBuffer partOff, ByteBuffer outPartition,
                                    ByteBuffer outMemberRank, int nMembers, ByteBuffer memberOff,
                                    ByteBuffer groupedTopic, ByteBuffer groupedPartition);

    /**
     * la_group_last_by_member: groupByMember on the results the last assignBatch call of this context left on the
     * device (that call may be given outPartition = outMemberRank = null), so the assignment crosses PCIe once.
     */
    static native int groupLastByMember(long ctx, int nMembers, ByteBuffer memberOff, ByteBuffer groupedTopic,
                                        ByteBuffer groupedPartition);

    /** la_last_error */
    static native String lastError(long ctx);
}
