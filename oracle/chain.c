/*
 * ORACLE (test infrastructure only — never linked into the product library).
 *
 * The reference's per-chunk transform chain and its inverse, restated on the CPU:
 *   upload  core/src/main/java/io/aiven/kafka/tieredstorage/RemoteStorageManager.java:434-453
 *           Base(chunkSize) -> [Compression] -> [Encryption]          (compress THEN encrypt)
 *   fetch   core/.../fetch/DefaultChunkManager.java:50-70
 *           Base -> [Decryption] -> [Decompression]                    (decrypt THEN decompress)
 *   chunker core/.../transform/BaseTransformChunkEnumeration.java:79-97 (fixed chunks, last may be short)
 * plus the additive CRC32C stage of SURVEY §8 a15 (over the ORIGINAL chunk bytes).
 *
 * Two AES-GCM back ends: the plain-C restatement in aes_gcm.c (the checker) and OpenSSL EVP
 * aes-256-gcm (AES-NI/PCLMUL — the instruction class the JDK intrinsics use), the latter only so that
 * bench.py's cpu_baseline leg is a fair, fast CPU number.  The two are cross-checked in tests/.
 * orc_chain_run_threads() is that cpu_baseline leg: one chunk at a time per thread, as the reference's
 * pull-driven enumeration does per RLM task thread (README.md:221).
 */
#define _GNU_SOURCE
#include <openssl/evp.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/types.h>

uint32_t orc_crc32c(const uint8_t* p, size_t n);
size_t orc_gcm_encrypt_chunk(const uint8_t key[32], const uint8_t iv[12], const uint8_t* aad, size_t aad_len,
                             const uint8_t* pt, size_t n, uint8_t* out);
long orc_gcm_decrypt_chunk(const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                           const uint8_t* chunk, size_t len, uint8_t* out);
size_t orc_zstd_compress_chunk(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int level);
long long orc_zstd_decompress_chunk(const uint8_t* frame, size_t len, uint8_t* dst, size_t cap);
size_t orc_zstd_compress_bound(size_t n);

#define ORC_COMPRESS 1u
#define ORC_ENCRYPT 2u
#define ORC_CRC 4u
#define ORC_OPENSSL 0x100u /* use OpenSSL for the GCM stage instead of aes_gcm.c */

size_t orc_gcm_encrypt_chunk_openssl(const uint8_t key[32], const uint8_t iv[12], const uint8_t* aad, size_t aad_len,
                                     const uint8_t* pt, size_t n, uint8_t* out) {
    EVP_CIPHER_CTX* c = EVP_CIPHER_CTX_new();
    int len = 0, ok = 1;
    memcpy(out, iv, 12);
    ok &= EVP_EncryptInit_ex(c, EVP_aes_256_gcm(), NULL, NULL, NULL);
    ok &= EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, NULL);
    ok &= EVP_EncryptInit_ex(c, NULL, NULL, key, iv);
    if (aad_len) ok &= EVP_EncryptUpdate(c, NULL, &len, aad, (int)aad_len);
    size_t off = 0;
    while (off < n) {                    /* EVP takes int lengths */
        size_t m = n - off > (1u << 30) ? (1u << 30) : n - off;
        ok &= EVP_EncryptUpdate(c, out + 12 + off, &len, pt + off, (int)m);
        off += m;
    }
    ok &= EVP_EncryptFinal_ex(c, out + 12 + n, &len);
    ok &= EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_GET_TAG, 16, out + 12 + n);
    EVP_CIPHER_CTX_free(c);
    return ok ? n + 28 : (size_t)-1;
}

long orc_gcm_decrypt_chunk_openssl(const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                                   const uint8_t* chunk, size_t len, uint8_t* out) {
    if (len < 28) return -2;
    size_t n = len - 28;
    EVP_CIPHER_CTX* c = EVP_CIPHER_CTX_new();
    int l = 0, ok = 1;
    ok &= EVP_DecryptInit_ex(c, EVP_aes_256_gcm(), NULL, NULL, NULL);
    ok &= EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_IVLEN, 12, NULL);
    ok &= EVP_DecryptInit_ex(c, NULL, NULL, key, chunk);
    if (aad_len) ok &= EVP_DecryptUpdate(c, NULL, &l, aad, (int)aad_len);
    ok &= EVP_DecryptUpdate(c, out, &l, chunk + 12, (int)n);
    ok &= EVP_CIPHER_CTX_ctrl(c, EVP_CTRL_GCM_SET_TAG, 16, (void*)(chunk + 12 + n));
    int fin = EVP_DecryptFinal_ex(c, out + l, &l);
    EVP_CIPHER_CTX_free(c);
    if (!ok) return -3;
    return fin > 0 ? (long)n : -1;
}

/* One chunk through the chain.  dst must hold orc_chain_bound(n, flags).  Returns the transformed size
 * ((size_t)-1 on failure); *crc receives CRC32C(original) when ORC_CRC is set. */
size_t orc_chain_bound(size_t n, unsigned flags) {
    size_t m = (flags & ORC_COMPRESS) ? orc_zstd_compress_bound(n) : n;
    return (flags & ORC_ENCRYPT) ? m + 28 : m;
}

size_t orc_transform_chunk(unsigned flags, const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                           const uint8_t iv[12], const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                           uint8_t* scratch, uint32_t* crc) {
    if (flags & ORC_CRC) *crc = orc_crc32c(src, n);
    const uint8_t* cur = src;
    size_t m = n;
    if (flags & ORC_COMPRESS) {
        uint8_t* z = (flags & ORC_ENCRYPT) ? scratch : dst;
        size_t zc = (flags & ORC_ENCRYPT) ? orc_zstd_compress_bound(n) : cap;
        m = orc_zstd_compress_chunk(src, n, z, zc, 0);
        if (m == (size_t)-1) return m;
        cur = z;
    }
    if (flags & ORC_ENCRYPT) {
        if (cap < m + 28) return (size_t)-1;
        return (flags & ORC_OPENSSL) ? orc_gcm_encrypt_chunk_openssl(key, iv, aad, aad_len, cur, m, dst)
                                     : orc_gcm_encrypt_chunk(key, iv, aad, aad_len, cur, m, dst);
    }
    if (cur != dst) { if (cap < m) return (size_t)-1; memcpy(dst, cur, m); }
    return m;
}

/* Inverse.  Returns restored size; -1 tag mismatch, -2 short chunk, -10-x zstd error x. */
long long orc_detransform_chunk(unsigned flags, const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                                const uint8_t* src, size_t len, uint8_t* dst, size_t cap, uint8_t* scratch,
                                uint32_t* crc) {
    const uint8_t* cur = src;
    size_t m = len;
    if (flags & ORC_ENCRYPT) {
        uint8_t* p = (flags & ORC_COMPRESS) ? scratch : dst;
        long r = (flags & ORC_OPENSSL) ? orc_gcm_decrypt_chunk_openssl(key, aad, aad_len, src, len, p)
                                       : orc_gcm_decrypt_chunk(key, aad, aad_len, src, len, p);
        if (r < 0) return r;
        cur = p; m = (size_t)r;
    }
    if (flags & ORC_COMPRESS) {
        long long r = orc_zstd_decompress_chunk(cur, m, dst, cap);
        if (r < 0) return -10 + r;
        m = (size_t)r;
    } else if (cur != dst) {
        if (cap < m) return -2;
        memcpy(dst, cur, m);
    }
    if (flags & ORC_CRC) *crc = orc_crc32c(dst, m);
    return (long long)m;
}

/* ---- threaded CPU baseline ------------------------------------------------------------------- */
typedef struct {
    unsigned flags; const uint8_t* key; const uint8_t* aad; size_t aad_len;
    const uint8_t* src; size_t chunk; size_t nchunks; const uint8_t* ivs;
    uint8_t* dst; size_t dst_stride; uint32_t* sizes; uint32_t* crcs;
    int tid, nthreads; volatile long* next;
    int fd;                  /* >= 0: every transformed chunk is also written to this file (BASELINE configs[0]: "to filesystem backend") */
} job_t;

/* Sink of the threaded baseline: a file (tmpfs on the bench box) that receives the transformed chunks at slot offsets, as the
 * reference's FileSystemStorage receives the `.log` object (storage/filesystem/.../FileSystemStorage.java: Files.copy of the
 * transformed stream).  NULL / "" = in memory only. */
static char g_sink_path[512];
void orc_chain_set_sink(const char* path) {
    if (!path) { g_sink_path[0] = 0; return; }
    strncpy(g_sink_path, path, sizeof g_sink_path - 1); g_sink_path[sizeof g_sink_path - 1] = 0;
}

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    uint8_t* scratch = (uint8_t*)malloc(orc_zstd_compress_bound(j->chunk) + 64);
    for (;;) {
        long i = __sync_fetch_and_add(j->next, 1);
        if ((size_t)i >= j->nchunks) break;
        uint32_t crc = 0;
        size_t r = orc_transform_chunk(j->flags, j->key, j->aad, j->aad_len, j->ivs + 12 * i,
                                       j->src + (size_t)i * j->chunk, j->chunk,
                                       j->dst + (size_t)i * j->dst_stride, j->dst_stride, scratch, &crc);
        j->sizes[i] = (uint32_t)r;
        j->crcs[i] = crc;
        if (j->fd >= 0 && r != (size_t)-1) {
            size_t done = 0;
            while (done < r) {
                ssize_t w = pwrite(j->fd, j->dst + (size_t)i * j->dst_stride + done, r - done, (off_t)((size_t)i * j->dst_stride + done));
                if (w <= 0) break;
                done += (size_t)w;
            }
        }
    }
    free(scratch);
    return NULL;
}

/* Transforms nchunks equal chunks with nthreads threads; returns wall seconds. */
double orc_chain_run_threads(unsigned flags, const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                             const uint8_t* src, size_t chunk, size_t nchunks, const uint8_t* ivs,
                             uint8_t* dst, size_t dst_stride, uint32_t* sizes, uint32_t* crcs, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * nthreads);
    job_t* jobs = (job_t*)malloc(sizeof(job_t) * nthreads);
    volatile long next = 0;
    struct timespec t0, t1;
    orc_zstd_compress_bound(1);            /* force dlopen outside the timed region */
    int fd = -1;
    if (g_sink_path[0]) fd = open(g_sink_path, O_CREAT | O_TRUNC | O_WRONLY, 0600);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (job_t){flags, key, aad, aad_len, src, chunk, nchunks, ivs, dst, dst_stride, sizes, crcs, t, nthreads, &next, fd};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    if (fd >= 0) close(fd);                /* inside the timed region, like the copy's close() */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (fd >= 0) unlink(g_sink_path);
    free(th); free(jobs);
    return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
