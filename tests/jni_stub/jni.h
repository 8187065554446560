/*
 * jni.h — STAND-IN, TEST INFRASTRUCTURE ONLY.  This image has no JDK; this file declares the small subset of the Java Native Interface that
 * integration/jni/raftgpu_jni.c uses, with the names, argument orders and types of the JNI specification (Java SE 8, chapter 4 "JNI Functions"),
 * so that the shim can be type-checked against include/raftgpu.h + include/raftwire.h (tests/test_jni_shim_cpu.py: gcc -fsyntax-only -Wall
 * -Werror) and driven through a fake JNIEnv (tests/native/jni_harness.c).  It is NOT a JDK header: the function table holds only the entries
 * the shim calls, in no particular slot order, and must never be used to build a library that a JVM loads.  A maintainer builds the shim
 * against $JAVA_HOME/include/jni.h (integration/README.md).
 */
#ifndef RAFTGPU_TEST_JNI_STANDIN_H
#define RAFTGPU_TEST_JNI_STANDIN_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef int32_t jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jobjectArray;

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (JNICALL *FindClass)(JNIEnv *env, const char *name);
    jint (JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jstring (JNICALL *NewStringUTF)(JNIEnv *env, const char *utf);
    jsize (JNICALL *GetArrayLength)(JNIEnv *env, jarray array);
    jobject (JNICALL *GetObjectArrayElement)(JNIEnv *env, jobjectArray array, jsize index);
    jobject (JNICALL *NewDirectByteBuffer)(JNIEnv *env, void *address, jlong capacity);
    void *(JNICALL *GetDirectBufferAddress)(JNIEnv *env, jobject buf);
    jlong (JNICALL *GetDirectBufferCapacity)(JNIEnv *env, jobject buf);
};
#endif
