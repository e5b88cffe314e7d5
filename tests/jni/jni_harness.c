/* TEST HARNESS ONLY: drives the JNI shim (java/jni/tsx_jni.c) the way TsxNative's native methods would, with a hand-made JNIEnv
 * (tests/jni/jni.h) - byte[] and direct ByteBuffers are plain structs here.  Linked against the CPU-emulated build of libtsxform
 * by tests/test_jni_shim.py.  Checks: init, transformedBound, a transform/detransform round trip through direct buffers, packed
 * output, the per-chunk "Tag mismatch" of a forged chunk, rejection of a descriptor that points beyond the src buffer. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jni.h"
#include "tsxform.h"

struct _jobject { void* addr; jlong cap; };             /* a direct ByteBuffer, a byte[] (cap = length) or a String (addr = chars) */
static jsize f_len(JNIEnv* e, jbyteArray a) { (void)e; return (jsize)a->cap; }
static void f_region(JNIEnv* e, jbyteArray a, jsize off, jsize n, jbyte* out) { (void)e; memcpy(out, (char*)a->addr + off, (size_t)n); }
static void* f_addr(JNIEnv* e, jobject b) { (void)e; return b ? b->addr : NULL; }
static jlong f_cap(JNIEnv* e, jobject b) { (void)e; return b ? b->cap : -1; }
static jstring f_str(JNIEnv* e, const char* s) { (void)e; jobject o = malloc(sizeof *o); o->addr = strdup(s); o->cap = (jlong)strlen(s); return o; }
static const struct JNINativeInterface_ kFns = {f_len, f_region, f_addr, f_cap, f_str};

jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_init(JNIEnv*, jclass);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_deviceCount(JNIEnv*, jclass);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_setThreadDevice(JNIEnv*, jclass, jint);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostRegister(JNIEnv*, jclass, jobject);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostUnregister(JNIEnv*, jclass, jobject);
jstring Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_strerror(JNIEnv*, jclass, jint);
jlong Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformedBound(JNIEnv*, jclass, jlong, jint);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatch(JNIEnv*, jclass, jint, jbyteArray, jbyteArray, jint, jobject, jint, jobject, jobject);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatchPacked(JNIEnv*, jclass, jint, jbyteArray, jbyteArray, jint, jobject, jint, jobject, jobject);
jint Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_detransformBatch(JNIEnv*, jclass, jint, jbyteArray, jbyteArray, jobject, jint, jobject, jobject);

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
#define N 5
int main(void) {
    JNIEnv envp = &kFns; JNIEnv* env = &envp;
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_init(env, NULL) >= 1);
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_deviceCount(env, NULL) >= 1);
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_setThreadDevice(env, NULL, 0) == 0);
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_setThreadDevice(env, NULL, 99) == TSX_E_INVAL);
    const int flags = TSX_ENCRYPT | TSX_CRC;
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformedBound(env, NULL, 4194304, TSX_ENCRYPT) == 4194332);
    unsigned char key[32], aad[32];
    for (int i = 0; i < 32; i++) { key[i] = (unsigned char)i; aad[i] = (unsigned char)(32 + i); }
    struct _jobject jkey = {key, 32}, jaad = {aad, 32};
    const uint32_t sizes[N] = {0, 1, 1000, 65537, 70001};
    tsx_chunk_desc d[N]; memset(d, 0, sizeof d);
    size_t so = 0, dof = 0;
    for (int i = 0; i < N; i++) {
        d[i].src_off = so; d[i].dst_off = dof; d[i].src_len = sizes[i]; d[i].dst_cap = sizes[i] + 28;
        for (int k = 0; k < 12; k++) d[i].iv[k] = (uint8_t)(i * 16 + k);
        so += ((sizes[i] + 15) & ~15u) + 16; dof += ((sizes[i] + 28 + 15) & ~15u) + 16;
    }
    unsigned char* src = calloc(so, 1); unsigned char* dst = calloc(dof, 1); unsigned char* back = calloc(so, 1); unsigned char* packed = calloc(dof, 1);
    for (size_t i = 0; i < so; i++) src[i] = (unsigned char)(i * 131 + (i >> 9));
    struct _jobject jd = {d, sizeof d}, jsrc = {src, (jlong)so}, jdst = {dst, (jlong)dof}, jback = {back, (jlong)so}, jpk = {packed, (jlong)dof};
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostRegister(env, NULL, &jsrc) == 0);
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatch(env, NULL, flags, &jkey, &jaad, 0, &jd, N, &jsrc, &jdst) == 0);
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_hostUnregister(env, NULL, &jsrc) == 0);
    uint32_t crc[N];
    for (int i = 0; i < N; i++) { CHECK(d[i].status == 0 && d[i].dst_len == sizes[i] + 28); crc[i] = d[i].crc32c; CHECK(memcmp(dst + d[i].dst_off, d[i].iv, 12) == 0); }
    /* packed: same bytes, back to back */
    tsx_chunk_desc p[N]; memcpy(p, d, sizeof p);
    struct _jobject jp = {p, sizeof p};
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatchPacked(env, NULL, flags, &jkey, &jaad, 0, &jp, N, &jsrc, &jpk) == 0);
    size_t at = 0;
    for (int i = 0; i < N; i++) { CHECK(p[i].status == 0 && p[i].dst_off == at && p[i].dst_len == d[i].dst_len); CHECK(memcmp(packed + at, dst + d[i].dst_off, d[i].dst_len) == 0); at += p[i].dst_len; }
    /* inverse, chunk 3 forged */
    dst[d[3].dst_off + 40] ^= 1;
    tsx_chunk_desc e[N]; memset(e, 0, sizeof e);
    for (int i = 0; i < N; i++) { e[i].src_off = d[i].dst_off; e[i].src_len = d[i].dst_len; e[i].dst_off = d[i].src_off; e[i].dst_cap = sizes[i]; }
    struct _jobject je = {e, sizeof e};
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_detransformBatch(env, NULL, flags, &jkey, &jaad, &je, N, &jdst, &jback) == 0);
    for (int i = 0; i < N; i++) {
        if (i == 3) { CHECK(e[i].status == TSX_E_TAG_MISMATCH && e[i].dst_len == 0); continue; }
        CHECK(e[i].status == 0 && e[i].dst_len == sizes[i] && e[i].crc32c == crc[i]);
        CHECK(memcmp(back + e[i].dst_off, src + d[i].src_off, sizes[i]) == 0);
    }
    jstring s = Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_strerror(env, NULL, TSX_E_TAG_MISMATCH);
    CHECK(strcmp((const char*)s->addr, "Tag mismatch") == 0);
    /* a descriptor that reaches beyond the src ByteBuffer never gets to the library */
    tsx_chunk_desc bad[1]; memcpy(bad, d, sizeof bad); bad[0].src_off = so - 16; bad[0].src_len = 64;
    struct _jobject jbad = {bad, sizeof bad};
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatch(env, NULL, flags, &jkey, &jaad, 0, &jbad, 1, &jsrc, &jdst) == TSX_E_INVAL);
    /* wrong key length */
    struct _jobject shortkey = {key, 16};
    CHECK(Java_io_aiven_kafka_tieredstorage_gpu_TsxNative_transformBatch(env, NULL, flags, &shortkey, &jaad, 0, &jd, N, &jsrc, &jdst) == TSX_E_INVAL);
    printf("jni shim ok\n");
    return 0;
}
