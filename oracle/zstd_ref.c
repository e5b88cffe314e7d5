/*
 * ORACLE (test infrastructure only — never linked into the product library).
 *
 * Zstandard exactly as the reference drives it through zstd-jni:
 *   core/src/main/java/io/aiven/kafka/tieredstorage/transform/CompressionChunkEnumeration.java:50-63
 *       new ZstdCompressCtx(); setPledgedSrcSize(len); setContentSize(true); compress(chunk)
 *       => ZSTD_createCCtx, ZSTD_CCtx_setPledgedSrcSize, ZSTD_c_contentSizeFlag=1, ZSTD_compress2
 *          (level untouched => ZSTD_CLEVEL_DEFAULT = 3, no checksum, no dict, fresh context per chunk)
 *   core/.../transform/DecompressionChunkEnumeration.java:39-46
 *       Zstd.decompressedSize(chunk) (<0 => RuntimeException), Zstd.decompress(chunk, size)
 *       => ZSTD_getFrameContentSize, ZSTD_decompress
 *
 * The arithmetic is third-party: com.github.luben:zstd-jni:1.5.6-9 (core/build.gradle:29) bundles
 * libzstd 1.5.6, which is NOT under /root/reference and not present in this image.  This file does
 * not restate it; it dlopen()s a real libzstd so the checker is the genuine library:
 *   1. $TSX_ORACLE_LIBZSTD if set,
 *   2. libzstd 1.5.7 bundled with Pillow (closest to 1.5.6; same block compressor),
 *   3. the system libzstd 1.4.8.
 * Every parity result names orc_zstd_version().  PARITY UNPINNED vs 1.5.6 for compressible
 * multi-block frames: no reference test pins those bytes; the reference's single golden frame
 * (CT/manifest/index/ChunkIndexSerializationTest.java:39-61) and raw-block frames are exact pins.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void* (*fn_createCCtx)(void);
typedef size_t (*fn_freeCCtx)(void*);
typedef size_t (*fn_setParam)(void*, int, int);
typedef size_t (*fn_setPledged)(void*, unsigned long long);
typedef size_t (*fn_compress2)(void*, void*, size_t, const void*, size_t);
typedef size_t (*fn_compressBound)(size_t);
typedef unsigned (*fn_isError)(size_t);
typedef const char* (*fn_errName)(size_t);
typedef unsigned long long (*fn_frameContentSize)(const void*, size_t);
typedef size_t (*fn_decompress)(void*, size_t, const void*, size_t);
typedef const char* (*fn_versionString)(void);

static struct {
    void* h;
    fn_createCCtx createCCtx; fn_freeCCtx freeCCtx; fn_setParam setParam; fn_setPledged setPledged;
    fn_compress2 compress2; fn_compressBound compressBound; fn_isError isError; fn_errName errName;
    fn_frameContentSize frameContentSize; fn_decompress decompress; fn_versionString versionString;
    char path[512];
} Z;

static const char* k_candidates[] = {
    "/usr/local/lib/python3.10/dist-packages/pillow.libs/libzstd-6ea785c0.so.1.5.7",
    "/usr/lib/x86_64-linux-gnu/libzstd.so.1",
    "libzstd.so.1",
    NULL,
};

static int try_open(const char* p) {
    /* DEEPBIND: the library's internal calls bind to ITSELF even when the process (a profiler) has another libzstd loaded globally -
     * under rocprofv3 they went to the profiler's own copy and crashed.  The sanitizer runtime refuses DEEPBIND, and needs none. */
    const int deep = (dlsym(RTLD_DEFAULT, "__asan_init") || dlsym(RTLD_DEFAULT, "__tsan_init") || dlsym(RTLD_DEFAULT, "__msan_init")) ? 0 : RTLD_DEEPBIND;
    void* h = dlopen(p, RTLD_NOW | RTLD_LOCAL | deep);
    if (!h) return -1;
    Z.h = h;
    Z.createCCtx = (fn_createCCtx)dlsym(h, "ZSTD_createCCtx");
    Z.freeCCtx = (fn_freeCCtx)dlsym(h, "ZSTD_freeCCtx");
    Z.setParam = (fn_setParam)dlsym(h, "ZSTD_CCtx_setParameter");
    Z.setPledged = (fn_setPledged)dlsym(h, "ZSTD_CCtx_setPledgedSrcSize");
    Z.compress2 = (fn_compress2)dlsym(h, "ZSTD_compress2");
    Z.compressBound = (fn_compressBound)dlsym(h, "ZSTD_compressBound");
    Z.isError = (fn_isError)dlsym(h, "ZSTD_isError");
    Z.errName = (fn_errName)dlsym(h, "ZSTD_getErrorName");
    Z.frameContentSize = (fn_frameContentSize)dlsym(h, "ZSTD_getFrameContentSize");
    Z.decompress = (fn_decompress)dlsym(h, "ZSTD_decompress");
    Z.versionString = (fn_versionString)dlsym(h, "ZSTD_versionString");
    if (!Z.createCCtx || !Z.compress2 || !Z.setParam || !Z.decompress || !Z.frameContentSize) { dlclose(h); Z.h = NULL; return -1; }
    snprintf(Z.path, sizeof Z.path, "%s", p);
    return 0;
}

/* Open a specific libzstd (NULL/"" = default search).  Returns 0 on success. */
int orc_zstd_open(const char* path) {
    if (Z.h) { dlclose(Z.h); memset(&Z, 0, sizeof Z); }
    if (path && *path) return try_open(path);
    const char* env = getenv("TSX_ORACLE_LIBZSTD");
    if (env && *env && try_open(env) == 0) return 0;
    for (int i = 0; k_candidates[i]; i++) if (try_open(k_candidates[i]) == 0) return 0;
    return -1;
}

static int ensure(void) { return Z.h ? 0 : orc_zstd_open(NULL); }

const char* orc_zstd_version(void) { return ensure() == 0 && Z.versionString ? Z.versionString() : "unavailable"; }
const char* orc_zstd_path(void) { return ensure() == 0 ? Z.path : ""; }
size_t orc_zstd_compress_bound(size_t n) { return ensure() == 0 ? Z.compressBound(n) : 0; }

/* CompressionChunkEnumeration.nextElement(): one frame per chunk, fresh context.  level == 0 leaves the
 * library default (what the reference does); other values are for oracle experiments only.
 * Returns frame size, or (size_t)-1 on error. */
size_t orc_zstd_compress_chunk(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int level) {
    if (ensure() != 0) return (size_t)-1;
    void* c = Z.createCCtx();
    if (!c) return (size_t)-1;
    if (Z.setPledged) Z.setPledged(c, n);                 /* :54 advisory */
    Z.setParam(c, 200 /* ZSTD_c_contentSizeFlag */, 1);   /* :60 */
    if (level) Z.setParam(c, 100 /* ZSTD_c_compressionLevel */, level);
    size_t r = Z.compress2(c, dst, cap, src, n);          /* :61 */
    Z.freeCCtx(c);
    return Z.isError(r) ? (size_t)-1 : r;
}

/* DecompressionChunkEnumeration.nextElement(): returns the decompressed size, -1 when
 * Zstd.decompressedSize would be negative/unknown (reference throws "Invalid decompressed size"),
 * -2 when dst is too small, -3 when the frame is corrupt. */
long long orc_zstd_decompress_chunk(const uint8_t* frame, size_t len, uint8_t* dst, size_t cap) {
    if (ensure() != 0) return -4;
    unsigned long long sz = Z.frameContentSize(frame, len);
    if (sz == (unsigned long long)-1 || sz == (unsigned long long)-2) return -1;
    if (sz > cap) return -2;
    size_t r = Z.decompress(dst, (size_t)sz, frame, len);
    if (Z.isError(r)) return -3;
    return (long long)r;
}

long long orc_zstd_frame_content_size(const uint8_t* frame, size_t len) {
    if (ensure() != 0) return -4;
    unsigned long long sz = Z.frameContentSize(frame, len);
    if (sz == (unsigned long long)-1) return -1;
    if (sz == (unsigned long long)-2) return -2;
    return (long long)sz;
}
