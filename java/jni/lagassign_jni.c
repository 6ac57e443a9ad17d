/* lagassign_jni.c -- the thin JNI -> C-ABI shim: one GetDirectBufferAddress per argument, one call.
 * NOT COMPILED in the image this repository is built in (no jni.h there); `make -C java/jni` builds it wherever a
 * JDK exists (JAVA_HOME), java/run_reference_tests.sh does so before running the reference's JUnit class.
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "lagassign.h"

#define ADDR(env, buf) ((buf) ? (*(env))->GetDirectBufferAddress((env), (buf)) : NULL)
#define CTX(h) ((la_ctx *)(intptr_t)(h))

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_deviceCount(JNIEnv *env, jclass cls) {
    (void)env; (void)cls;
    return la_device_count();
}

JNIEXPORT jlong JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_createMulti(JNIEnv *env, jclass cls, jintArray device_ids) {
    la_ctx *ctx = NULL;
    int rc;
    jsize n = device_ids ? (*env)->GetArrayLength(env, device_ids) : 0;
    (void)cls;
    if (n > 0) {
        jint *ids = (*env)->GetIntArrayElements(env, device_ids, NULL);
        if (!ids) return 0;                                   /* OutOfMemoryError is pending */
        rc = la_create_multi(&ctx, (int)n, (const int *)ids, 0);
        (*env)->ReleaseIntArrayElements(env, device_ids, ids, JNI_ABORT);
    } else {
        rc = la_create_multi(&ctx, 0, NULL, 0);               /* every device of the node */
    }
    if (rc != LA_OK) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalStateException");
        if (ex) (*env)->ThrowNew(env, ex, la_last_error(NULL));
        return 0;
    }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_shardCount(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    return la_shard_count(CTX(ctx));
}

JNIEXPORT void JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_destroy(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env; (void)cls;
    la_destroy(CTX(ctx));
}

JNIEXPORT jobject JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_hostAlloc(JNIEnv *env, jclass cls, jlong ctx, jlong bytes) {
    void *p;
    jobject buf;
    (void)cls;
    if (bytes < 0 || bytes > 0x7FFFFFFFLL) return NULL;     /* a ByteBuffer's capacity is an int */
    p = la_host_alloc(CTX(ctx), (size_t)bytes);
    if (!p) return NULL;
    buf = (*env)->NewDirectByteBuffer(env, p, bytes);
    if (!buf) la_host_free(CTX(ctx), p);                    /* OutOfMemoryError pending: do not leak the pinned block */
    return buf;
}

JNIEXPORT void JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_hostFree(JNIEnv *env, jclass cls, jlong ctx, jobject buffer) {
    (void)cls;
    la_host_free(CTX(ctx), ADDR(env, buffer));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_computeLag(
    JNIEnv *env, jclass cls, jlong ctx, jlong n, jobject begin, jobject end, jobject committed, jint reset_mode,
    jobject out_lag) {
    (void)cls;
    return la_compute_lag(CTX(ctx), (int64_t)n, (const int64_t *)ADDR(env, begin), (const int64_t *)ADDR(env, end),
                          (const int64_t *)ADDR(env, committed), reset_mode, (int64_t *)ADDR(env, out_lag));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_assignBatch(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject partition_id,
    jobject begin, jobject end, jobject committed, jint reset_mode, jobject cons_off, jobject cons_rank,
    jobject out_partition, jobject out_member_rank, jobject out_total_lag) {
    (void)cls;
    return la_assign_batch(CTX(ctx), n_topics,
                           (const int64_t *)ADDR(env, part_off), (const int32_t *)ADDR(env, partition_id),
                           (const int64_t *)ADDR(env, begin), (const int64_t *)ADDR(env, end),
                           (const int64_t *)ADDR(env, committed), reset_mode,
                           (const int64_t *)ADDR(env, cons_off), (const int32_t *)ADDR(env, cons_rank),
                           (int32_t *)ADDR(env, out_partition), (int32_t *)ADDR(env, out_member_rank),
                           (int64_t *)ADDR(env, out_total_lag));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_assignBatchLags(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject partition_id, jobject lag,
    jobject cons_off, jobject cons_rank, jobject out_partition, jobject out_member_rank, jobject out_total_lag) {
    (void)cls;
    return la_assign_batch_lags(CTX(ctx), n_topics,
                                (const int64_t *)ADDR(env, part_off), (const int32_t *)ADDR(env, partition_id),
                                (const int64_t *)ADDR(env, lag),
                                (const int64_t *)ADDR(env, cons_off), (const int32_t *)ADDR(env, cons_rank),
                                (int32_t *)ADDR(env, out_partition), (int32_t *)ADDR(env, out_member_rank),
                                (int64_t *)ADDR(env, out_total_lag));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_assignBatchGrouped(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject partition_id,
    jobject begin, jobject end, jobject committed, jint reset_mode, jobject cons_off, jobject cons_rank,
    jint n_members, jobject member_off, jobject grouped_topic, jobject grouped_partition, jobject out_total_lag) {
    (void)cls;
    return la_assign_batch_grouped(CTX(ctx), n_topics,
                                   (const int64_t *)ADDR(env, part_off), (const int32_t *)ADDR(env, partition_id),
                                   (const int64_t *)ADDR(env, begin), (const int64_t *)ADDR(env, end),
                                   (const int64_t *)ADDR(env, committed), reset_mode,
                                   (const int64_t *)ADDR(env, cons_off), (const int32_t *)ADDR(env, cons_rank),
                                   n_members, (int64_t *)ADDR(env, member_off), (int32_t *)ADDR(env, grouped_topic),
                                   (int32_t *)ADDR(env, grouped_partition), (int64_t *)ADDR(env, out_total_lag));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_assignBatchGroupedSparse(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject partition_id, jobject end,
    jobject committed, jint reset_mode, jlong n_none, jobject none_index, jobject none_begin, jobject cons_off,
    jobject cons_rank, jint n_members, jobject member_off, jobject grouped_topic, jobject grouped_partition,
    jobject out_total_lag) {
    (void)cls;
    return la_assign_batch_grouped_sparse(CTX(ctx), n_topics,
                                          (const int64_t *)ADDR(env, part_off), (const int32_t *)ADDR(env, partition_id),
                                          (const int64_t *)ADDR(env, end), (const int64_t *)ADDR(env, committed), reset_mode,
                                          (int64_t)n_none, (const int64_t *)ADDR(env, none_index),
                                          (const int64_t *)ADDR(env, none_begin),
                                          (const int64_t *)ADDR(env, cons_off), (const int32_t *)ADDR(env, cons_rank),
                                          n_members, (int64_t *)ADDR(env, member_off), (int32_t *)ADDR(env, grouped_topic),
                                          (int32_t *)ADDR(env, grouped_partition), (int64_t *)ADDR(env, out_total_lag));
}

/* la_hint_next_call with LA_HINT_BOUNDS: what the marshalling loop saw -- the largest end offset (or lag) and partition id. */
JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_hintNextCallBounds(JNIEnv *env, jclass cls, jlong ctx, jlong max_lag,
                                                                       jlong max_partition_id) {
    la_call_hints h;
    (void)env;
    (void)cls;
    h.struct_size = (int32_t)sizeof h;
    h.flags = LA_HINT_BOUNDS;
    h.max_lag = (int64_t)max_lag;
    h.max_partition_id = (int64_t)max_partition_id;
    return la_hint_next_call(CTX(ctx), &h);
}

JNIEXPORT jlong JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_lastLaunches(JNIEnv *env, jclass cls, jlong ctx) {
    (void)env;
    (void)cls;
    return (jlong)la_last_launches(CTX(ctx));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_version(JNIEnv *env, jclass cls) {
    (void)env;
    (void)cls;
    return la_version();
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_groupLastByMember(
    JNIEnv *env, jclass cls, jlong ctx, jint n_members, jobject member_off, jobject grouped_topic,
    jobject grouped_partition) {
    (void)cls;
    return la_group_last_by_member(CTX(ctx), n_members, (int64_t *)ADDR(env, member_off),
                                   (int32_t *)ADDR(env, grouped_topic), (int32_t *)ADDR(env, grouped_partition));
}

JNIEXPORT jstring JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_lastError(JNIEnv *env, jclass cls, jlong ctx) {
    (void)cls;
    return (*env)->NewStringUTF(env, la_last_error((const la_ctx *)(intptr_t)ctx));
}
