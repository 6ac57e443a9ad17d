/* TEST INFRASTRUCTURE: a declarations-only stand-in for the JDK's jni.h, just wide enough for
 * java/jni/lagassign_jni.c, so that the shim can be syntax- and type-checked where no JDK exists
 * (tests/test_host_cpu.py, `make -C java/jni check`).  Names and signatures follow the JNI specification. */
#ifndef LA_TEST_JNI_STUB_H
#define LA_TEST_JNI_STUB_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

typedef int32_t jint;
typedef int64_t jlong;
typedef jint jsize;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jintArray;
typedef unsigned char jboolean;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *env, const char *name);
    jint (*ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jstring (*NewStringUTF)(JNIEnv *env, const char *utf);
    jsize (*GetArrayLength)(JNIEnv *env, jintArray array);
    jint *(*GetIntArrayElements)(JNIEnv *env, jintArray array, jboolean *isCopy);
    void (*ReleaseIntArrayElements)(JNIEnv *env, jintArray array, jint *elems, jint mode);
    jobject (*NewDirectByteBuffer)(JNIEnv *env, void *address, jlong capacity);
    void *(*GetDirectBufferAddress)(JNIEnv *env, jobject buf);
};
#endif
