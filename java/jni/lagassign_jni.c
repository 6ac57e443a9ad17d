/* lagassign_jni.c -- the thin JNI -> C-ABI shim.  SOURCE ONLY (no jni.h in the build image).
 *
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *      lagassign_jni.c -L../../kafka_lag_based_assignor_amd -llagassign -o liblagassign_jni.so
 */
#include <jni.h>
#include <stdint.h>

#include "lagassign.h"

#define ADDR(env, buf) ((buf) ? (*(env))->GetDirectBufferAddress((env), (buf)) : NULL)

JNIEXPORT jlong JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_create(JNIEnv *env, jclass cls, jint device) {
    la_ctx *ctx = NULL;
    int rc = la_create(&ctx, device, 0);
    if (rc != LA_OK) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalStateException");
        (*env)->ThrowNew(env, ex, la_last_error(NULL));
        return 0;
    }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_destroy(JNIEnv *env, jclass cls, jlong ctx) {
    la_destroy((la_ctx *)(intptr_t)ctx);
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_assignBatch(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject partition_id,
    jobject begin, jobject end, jobject committed, jint reset_mode, jobject cons_off, jobject cons_rank,
    jobject out_partition, jobject out_member_rank, jobject out_total_lag) {
    return la_assign_batch((la_ctx *)(intptr_t)ctx, n_topics,
                           (const int64_t *)ADDR(env, part_off), (const int32_t *)ADDR(env, partition_id),
                           (const int64_t *)ADDR(env, begin), (const int64_t *)ADDR(env, end),
                           (const int64_t *)ADDR(env, committed), reset_mode,
                           (const int64_t *)ADDR(env, cons_off), (const int32_t *)ADDR(env, cons_rank),
                           (int32_t *)ADDR(env, out_partition), (int32_t *)ADDR(env, out_member_rank),
                           (int64_t *)ADDR(env, out_total_lag));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_groupByMember(
    JNIEnv *env, jclass cls, jlong ctx, jint n_topics, jobject part_off, jobject out_partition,
    jobject out_member_rank, jint n_members, jobject member_off, jobject grouped_topic, jobject grouped_partition) {
    return la_group_by_member((la_ctx *)(intptr_t)ctx, n_topics, (const int64_t *)ADDR(env, part_off),
                              (const int32_t *)ADDR(env, out_partition), (const int32_t *)ADDR(env, out_member_rank),
                              n_members, (int64_t *)ADDR(env, member_off), (int32_t *)ADDR(env, grouped_topic),
                              (int32_t *)ADDR(env, grouped_partition));
}

JNIEXPORT jint JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_groupLastByMember(
    JNIEnv *env, jclass cls, jlong ctx, jint n_members, jobject member_off, jobject grouped_topic,
    jobject grouped_partition) {
    return la_group_last_by_member((la_ctx *)(intptr_t)ctx, n_members, (int64_t *)ADDR(env, member_off),
                                   (int32_t *)ADDR(env, grouped_topic), (int32_t *)ADDR(env, grouped_partition));
}

JNIEXPORT jstring JNICALL
Java_com_github_grantneale_kafka_gpu_LagAssignNative_lastError(JNIEnv *env, jclass cls, jlong ctx) {
    return (*env)->NewStringUTF(env, la_last_error((const la_ctx *)(intptr_t)ctx));
}
