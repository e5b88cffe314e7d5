/* TEST HARNESS ONLY: a minimal stand-in for the JDK's <jni.h> so that java/jni/tsx_jni.c can be compiled and driven without a
 * JVM (this image has no JDK).  It declares just the JNI types and the JNIEnv functions the shim uses; tests/jni/jni_harness.c
 * provides them.  The real build uses $JAVA_HOME/include/jni.h (see the header of tsx_jni.c). */
#ifndef TSX_TEST_JNI_H
#define TSX_TEST_JNI_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jbyteArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
    jsize (*GetArrayLength)(JNIEnv*, jbyteArray);
    void (*GetByteArrayRegion)(JNIEnv*, jbyteArray, jsize, jsize, jbyte*);
    void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
    jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject);
    jstring (*NewStringUTF)(JNIEnv*, const char*);
};
#endif
