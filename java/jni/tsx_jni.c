/*
 * JNI shim between io.aiven.kafka.tieredstorage.gpu.TsxNative and the C ABI of libtsxform.so (include/tsxform.h).
 * Build (needs a JDK, which this repository's build image does not have):
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include tsx_jni.c \
 *       -L../../tiered-storage-for-apache-kafka_amd -ltsxform -o libtsxform_jni.so
 * Nothing here computes: it only moves pointers across the boundary and never throws from native code.
 */
#define _GNU_SOURCE
#include <jni.h>
#include <string.h>   /* explicit_bzero (glibc >= 2.25) */

#include "tsxform.h"

/* The shim was compiled against ONE version of tsxform.h; ABI 3 put src_size in the middle of the batch entry points, so a
 * libtsxform.so of another ABI would be called with shifted arguments (a size taken for a pointer).  init() is the first native
 * call TsxNative's static initialiser makes: a mismatched pair fails there, before any batch call exists. */
JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_init(JNIEnv* env, jclass cls) {
    (void)env; (void)cls;
    if (tsx_abi_version() != TSX_ABI_VERSION) return TSX_E_UNSUPPORTED;
    return tsx_init(0, NULL);
}

JNIEXPORT jstring JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_strerror(JNIEnv* env, jclass cls, jint code) {
    (void)cls;
    return (*env)->NewStringUTF(env, tsx_strerror(code));
}

JNIEXPORT jlong JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformedBound(JNIEnv* env, jclass cls, jlong n, jint flags) {
    (void)env; (void)cls;
    return (jlong)tsx_transformed_bound((size_t)n, (uint32_t)flags);
}

/* tsx_set_thread_device: device of this thread's batches (-1 = least loaded); GpuTransformChunkEnumeration passes
 * floorMod(segment hash, deviceCount()) so that one segment's chunks stay on one GPU */
JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_setThreadDevice(JNIEnv* env, jclass cls, jint device) {
    (void)env; (void)cls;
    return tsx_set_thread_device(device);
}

JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_deviceCount(JNIEnv* env, jclass cls) {
    (void)env; (void)cls;
    return tsx_device_count();
}

/* tsx_host_register / tsx_host_unregister on a direct buffer that is reused for batches (DMA without a staging pass) */
JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostRegister(JNIEnv* env, jclass cls, jobject buf) {
    (void)cls;
    void* p = (*env)->GetDirectBufferAddress(env, buf);
    const jlong cap = (*env)->GetDirectBufferCapacity(env, buf);
    return (p && cap > 0) ? tsx_host_register(p, (size_t)cap) : TSX_E_INVAL;
}

JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostUnregister(JNIEnv* env, jclass cls, jobject buf) {
    (void)cls;
    void* p = (*env)->GetDirectBufferAddress(env, buf);
    return p ? tsx_host_unregister(p) : TSX_E_INVAL;
}

static int fill_params(JNIEnv* env, tsx_batch_params* p, jint flags, jbyteArray key, jbyteArray aad, jint profile) {
    memset(p, 0, sizeof *p);
    p->flags = (uint32_t)flags;
    p->zstd_level = 0;                    /* library default = 3: CompressionChunkEnumeration.java:52 never sets a level */
    p->zstd_profile = (uint32_t)profile;
    if (flags & TSX_ENCRYPT) {
        if (!key || (*env)->GetArrayLength(env, key) != 32) return TSX_E_INVAL;
        (*env)->GetByteArrayRegion(env, key, 0, 32, (jbyte*)p->key);
        if (aad) {
            const jsize n = (*env)->GetArrayLength(env, aad);
            if (n > (jsize)sizeof p->aad) return TSX_E_INVAL;
            (*env)->GetByteArrayRegion(env, aad, 0, n, (jbyte*)p->aad);
            p->aad_len = (uint32_t)n;
        }
    }
    return TSX_OK;
}

static jint run(JNIEnv* env, int detransform, int mem_kind, jint flags, jbyteArray key, jbyteArray aad, jint profile,
                jobject descs, jint n, jobject src, jobject dst) {
    tsx_batch_params p;
    int rc = fill_params(env, &p, flags, key, aad, profile);
    if (rc == TSX_OK) {
        tsx_chunk_desc* d = (tsx_chunk_desc*)(*env)->GetDirectBufferAddress(env, descs);
        const void* s = (*env)->GetDirectBufferAddress(env, src);
        void* o = (*env)->GetDirectBufferAddress(env, dst);
        const jlong cap = (*env)->GetDirectBufferCapacity(env, dst);
        const jlong scap = (*env)->GetDirectBufferCapacity(env, src);
        if (n < 0 || !d || !s || !o || (*env)->GetDirectBufferCapacity(env, descs) < (jlong)n * (jlong)sizeof(tsx_chunk_desc)) rc = TSX_E_INVAL;
        /* the capacities of the two direct buffers are the bounds the library validates every descriptor against (ABI 3) */
        if (rc == TSX_OK)
            rc = detransform ? tsx_detransform_batch(NULL, &p, d, (uint32_t)n, s, (size_t)scap, o, (size_t)cap, mem_kind)
                             : tsx_transform_batch(NULL, &p, d, (uint32_t)n, s, (size_t)scap, o, (size_t)cap, mem_kind);
    }
    explicit_bzero(&p, sizeof p);         /* the key does not outlive the call (ownership); not a dead store */
    return rc;
}

JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatch(
    JNIEnv* env, jclass cls, jint flags, jbyteArray key, jbyteArray aad, jint profile, jobject descs, jint n, jobject src, jobject dst) {
    (void)cls;
    return run(env, 0, TSX_MEM_HOST, flags, key, aad, profile, descs, n, src, dst);
}

/* TSX_MEM_HOST_PACKED: dst is the upload's own buffer (a multipart part, a mapped file); chunk i lands at descs[i].dst_off */
JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatchPacked(
    JNIEnv* env, jclass cls, jint flags, jbyteArray key, jbyteArray aad, jint profile, jobject descs, jint n, jobject src, jobject dst) {
    (void)cls;
    return run(env, 0, TSX_MEM_HOST_PACKED, flags, key, aad, profile, descs, n, src, dst);
}

JNIEXPORT jint JNICALL Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_detransformBatch(
    JNIEnv* env, jclass cls, jint flags, jbyteArray key, jbyteArray aad, jobject descs, jint n, jobject src, jobject dst) {
    (void)cls;
    return run(env, 1, TSX_MEM_HOST, flags, key, aad, 1, descs, n, src, dst);
}
