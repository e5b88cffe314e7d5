/*
 * tsxform — C ABI of the MI355X-native chunk-transform library (libtsxform.so).
 *
 * This is the drop-in boundary for the hot path of aiven/tiered-storage-for-apache-kafka:
 * the per-chunk  compress -> AES-256-GCM encrypt (-> CRC32C)  chain of copyLogSegmentData() and its
 * inverse in fetchLogSegment().  A JNI shim (java/ sources) binds exactly these entry
 * points from GpuTransformChunkEnumeration / GpuDetransformChunkEnumeration / GpuChunkManager, which
 * implement the reference's own interfaces:
 *   core/src/main/java/io/aiven/kafka/tieredstorage/transform/TransformChunkEnumeration.java:28-42
 *   core/src/main/java/io/aiven/kafka/tieredstorage/transform/DetransformChunkEnumeration.java:28
 *   core/src/main/java/io/aiven/kafka/tieredstorage/fetch/ChunkManager.java:26-31
 *
 * Conventions: plain pointers and sizes only; caller owns every buffer; the library never keeps a
 * caller pointer after a call returns; nothing throws across the ABI.  Batch-level failures are the
 * (negative) return value; per-chunk failures are tsx_chunk_desc.status.  All entry points are
 * re-entrant: a tsx_ctx is a per-thread handle (own HIP stream + device workspace), and the
 * ctx-less convenience calls take one from an internal pool (reference threading: >=10 RLM upload
 * threads + the ChunkCache ForkJoinPool call concurrently).
 *
 * There is NO CPU implementation behind this ABI: without a usable gfx950 device tsx_init() fails
 * with TSX_E_DEVICE and every compute entry point returns TSX_E_DEVICE.
 */
#ifndef TSXFORM_H
#define TSXFORM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSX_ABI_VERSION 4 /* 2: ctx-less calls spread over all initialised devices; tsx_set_thread_device, tsx_host_register
                             3: the batch entry points take src_size and reject descriptors that reach beyond it
                             4: tsx_init_ex / tsx_config, tsx_service_stats, tsx_service_quiesce (the compressor service) */

/* flags: which stages of the chain run.  Replaces the reference's chain construction
 * RemoteStorageManager.transformation(), core/.../RemoteStorageManager.java:434-453
 * (Base -> [Compression] -> [Encryption]) and DefaultChunkManager.getChunk(),
 * core/.../fetch/DefaultChunkManager.java:56-67 (Base -> [Decryption] -> [Decompression]). */
#define TSX_COMPRESS 0x1u /* Zstd frame per chunk   (CompressionChunkEnumeration.java:50-63)      */
#define TSX_ENCRYPT  0x2u /* IV||C||TAG, AES-256-GCM (EncryptionChunkEnumeration.java:66-84)       */
#define TSX_CRC      0x4u /* CRC32C of the ORIGINAL chunk bytes, out of band       */

/* where src/dst live */
/* host pointers.  The batch is cut into pieces whose H2D copy, kernels and D2H copy overlap: pieces of >= 64 MiB in order on three
 * streams of the ctx; a compressing batch goes as up to 4 members of the device's compressor service (a chunk is ~1 s of one wave whatever
 * the batch size: piece k is published when its share of the input has landed, its output travels while later pieces run).  Any host memory
 * works - pageable buffers are staged by the HIP runtime; buffers pinned once with tsx_host_register() are copied by DMA without a staging pass.
 * A compressing batch whose WHOLE dst buffer the device can address (inside one tsx_host_register'ed buffer, or one hipHostMalloc'ed
 * allocation) has no output copy at all: the device writes every chunk's transformed bytes straight into [dst_off, dst_off + dst_len)
 * during the call (nothing beyond dst_len is written; what the buffer holds is defined when the call returns, as for any other kind).
 * A dst of which only a part is registered takes the copy path. */
#define TSX_MEM_HOST   0
/* device pointers (same HIP runtime / process): no copies.  The kernels run on the ctx's own streams: work the caller still has
 * queued on other streams for these buffers (the kernel that fills src, a memset of dst) must be complete when the call is made,
 * and the call returns when the library's work is complete. */
#define TSX_MEM_DEVICE 1
/* host pointers, transformed chunks written BACK TO BACK into dst in batch order - the bytes of the `.log` object (or of a
 * multipart part buffer) exactly as TransformFinisher.java:134-151 (SequenceInputStream over the chunks) hands them to
 * ObjectUploader.upload / S3MultiPartOutputStream.java:89-122, without the bound-sized slot per chunk and the gather copy
 * behind it.  tsx_transform_batch only.  descs[i].dst_off / dst_cap are ignored on entry; on return dst_off is chunk i's
 * offset in dst and dst_len its size; a chunk that no longer fits dst_size gets TSX_E_DST_TOO_SMALL (and so do all after it).
 * The bytes of dst BEHIND the packed chunks are scratch: when dst_size has room for one bound-sized slot per chunk and the device can
 * address dst, the chunks are written there first and packed down in place (no device output buffer, no copies). */
#define TSX_MEM_HOST_PACKED 2

/* status / error codes (0 = ok, negative = error) */
#define TSX_OK               0
#define TSX_E_INVAL         -1  /* bad argument                                                   */
#define TSX_E_DEVICE        -2  /* no usable gfx950 device / HIP failure                          */
#define TSX_E_NOMEM         -3
#define TSX_E_DST_TOO_SMALL -4  /* dst_cap < transformed size                                     */
#define TSX_E_TAG_MISMATCH  -5  /* GCM tag check failed  (JCE AEADBadTagException,
                                   DecryptionChunkEnumeration.java:59-61)                           */
#define TSX_E_BAD_FRAME     -6  /* corrupt Zstd frame (zstd-jni ZstdException)                    */
#define TSX_E_BAD_SIZE      -7  /* frame without usable content size: reference throws
                                   "Invalid decompressed size: n" (DecompressionChunkEnumeration.java:42-44) */
#define TSX_E_SHORT_CHUNK   -8  /* encrypted chunk shorter than IV+TAG                            */
#define TSX_E_UNSUPPORTED   -9

/* Per-chunk descriptor; mirrors io.aiven.kafka.tieredstorage.Chunk (core/.../Chunk.java:21-36:
 * id, originalPosition, originalSize, transformedPosition, transformedSize) with the in/out split
 * a batch call needs.  48 bytes, no padding. */
typedef struct tsx_chunk_desc {
    uint64_t src_off;  /* in : offset of this chunk's input within src                              */
    uint64_t dst_off;  /* in : offset of this chunk's output slot within dst                        */
    uint32_t src_len;  /* in : input bytes (originalSize on transform, transformedSize on detransform) */
    uint32_t dst_cap;  /* in : capacity of the output slot                                          */
    uint32_t dst_len;  /* out: bytes produced                                                       */
    uint32_t crc32c;   /* out: CRC32C of the original (pre-transform / restored) bytes if TSX_CRC   */
    int32_t  status;   /* out: TSX_OK or a TSX_E_* code for this chunk                              */
    uint8_t  iv[12];   /* in : GCM IV for transform (host SecureRandom, AesEncryptionProvider.java:66-71);
                          detransform reads the IV from the chunk's first 12 bytes and ignores this    */
} tsx_chunk_desc;

/* Per-batch parameters: one (data key, AAD) pair per segment
 * (AesEncryptionProvider.createDataKeyAndAAD, core/.../security/AesEncryptionProvider.java:52-58). */
typedef struct tsx_batch_params {
    uint32_t flags;        /* TSX_COMPRESS | TSX_ENCRYPT | TSX_CRC                                  */
    uint32_t aad_len;      /* reference: 32                                                         */
    uint8_t  key[32];      /* AES-256 data key (SecretKey.getEncoded())                             */
    uint8_t  aad[64];
    int32_t  zstd_level;   /* 0 = library default (3), what the reference uses; only 3 is implemented */
    uint32_t zstd_profile; /* TSX_ZSTD_PROFILE_*                                                    */
} tsx_batch_params;

/* Which libzstd behaviour the compressor reproduces.  TSX_ZSTD_PROFILE_1_5_7: byte for byte the real libzstd 1.5.7 (checked against
 * the library itself on every test input).  TSX_ZSTD_PROFILE_1_5_6: the same code WITHOUT 1.5.7's pre-block splitter - an UNVERIFIED
 * stand-in for the release the reference ships (1.5.6 inside zstd-jni 1.5.6-9, core/build.gradle:29): no libzstd 1.5.6 exists in the
 * build or test environment, and the 1.5.7 release notes may list more level-3 changes than the splitter (a ratio improvement of the
 * double-fast parser has been recalled by a reviewer; neither recollection could be checked).  Frames are valid Zstandard either way;
 * whether they are 1.5.6's bytes is settled where that library exists: tests/golden/make_vectors_from_lib.py + tests/test_golden.py. */
#define TSX_ZSTD_PROFILE_1_5_6 0u
#define TSX_ZSTD_PROFILE_1_5_7 1u

typedef struct tsx_ctx tsx_ctx;

/* Timing of the last batch on a ctx, measured with HIP events on the ctx's own stream. */
typedef struct tsx_timing {
    float total_ms;    /* first enqueue -> last kernel/copy done                                    */
    float h2d_ms, d2h_ms;
    float crc_ms, zstd_ms, gcm_ms, unzstd_ms;
    uint32_t crc_launches, zstd_launches, gcm_launches, unzstd_launches;
} tsx_timing;

/* ---- library lifetime ---------------------------------------------------------------------- */
uint32_t    tsx_abi_version(void);
const char* tsx_version(void);           /* "tsxform x.y (gfx950; zstd parity target 1.5.7/1.5.6 L3)" */
const char* tsx_strerror(int code);
/* device_ids == NULL: use devices 0..device_count-1; device_count <= 0: all visible devices.  An id may be listed more than once
 * (the same GPU as several logical devices, each with its own pools and compressor service: tests of the multi-device dispatch).
 * Returns the number of devices in use (>0) or TSX_E_DEVICE. */
int  tsx_init(int device_count, const int* device_ids);

/* What a deployment may want to say about the device-side machinery; TSX_CFG_DEFAULT(64) in a field = the library's default.
 * The environment of the process, read once inside tsx_init(_ex), overrides both: TSX_FETCH_RESERVED_CUS,
 * TSX_FETCH_SHARED_CU_WAVES, TSX_FETCH_QUIET_MS, TSX_SERVICE_MAX_LAUNCH_MS, TSX_POOL_IDLE_BYTES.  Nothing on a data path reads the environment. */
#define TSX_CFG_DEFAULT   0xFFFFFFFFu
#define TSX_CFG_DEFAULT64 0xFFFFFFFFFFFFFFFFull
typedef struct tsx_config {
    uint32_t struct_size;            /* sizeof(tsx_config) as the caller was compiled                                           */
    uint32_t fetch_reserved_cus;     /* compute units the compressor never occupies, so that a fetch (fetchLogSegment ->
                                        ChunkCache.java:85-108, get.timeout.ms 10 s) finds room at once while uploads fill the
                                        chip.  Default: one per SHADER ENGINE (32 of an MI355X's 256) - the hardware hands a
                                        kernel's workgroups to the engines in turn and a workgroup waits for room in its own;
                                        a smaller number is spread over the engines and leaves some without a free CU (a fetch
                                        can then wait for the end of a compressor launch); 0 = no reservation                  */
    uint32_t service_max_launch_ms;  /* one launch of the compressor service kernel stops taking chunks at this age (the next
                                        launch takes over): bounds how long a device-wide synchronisation made by OTHER code in
                                        the process (hipFree, hipDeviceSynchronize) can wait under continuous uploads; default
                                        60000, 0 = no limit                                                                   */
    uint32_t fetch_shared_cu_waves;  /* compressor waves that stay on each reserved CU all the same (0 / default: none - the CU is left
                                        alone).  A trade, measured under 5 upload callers (profiles/r05_keep_waves_on_reserved_cus.jsonl):
                                        4 -> uploads + 3 %, a 4 MiB fetch 2.8 -> 3.7 ms; 8 -> + 6 %, 4.2 ms, and now and then a fetch that
                                        waits ~0.5 s for the compressor launch to be rotated; 12 and more: such waits become regular.
                                        At most 8 is accepted                                                                  */
    uint64_t pool_idle_bytes;        /* idle pooled workspace kept per device; default 4/9 of its memory                      */
    uint32_t fetch_quiet_ms;         /* the reservation follows the traffic (default 2000; 0 = the reserved CUs are never used by the
                                        compressor).  Once no fetch (no batch of ordinary kernels) has run for this long, a launch of the
                                        compressor's kernel has waves on the reserved CUs too, as GUESTS: they work for as long as the
                                        queue has work for them - a chip that is full AND busy is what they are for; a guest that finds
                                        the queue dry for 10 ms (500 while most of the chip is busy) gives its slot back - and the next fetch makes them hand their chunks
                                        back and leave, which costs that ONE fetch a block time of a chunk (~30 ms); the CUs then stay
                                        reserved until it has been quiet again.  Measured (round 6, MI355X): bench.py value 18.2 -> 20.5,
                                        continuously fed 19.8 -> 22.3 GiB/s.  Why it was opt-in in round 5 and what was wrong then:
                                        profiles/r06_guest_waves_root_cause.md, r06_full_chip_with_idle_waves.txt                      */
    uint32_t reserved2_;
} tsx_config;
int  tsx_init_ex(int device_count, const int* device_ids, const tsx_config* cfg);   /* cfg == NULL: tsx_init */
/* Every tsx_ctx must have been destroyed and no batch may be in flight.  No entry point of this library changes the calling
 * thread's current HIP device: each one that selects a device puts the previous one back before it returns. */
void tsx_shutdown(void);
int  tsx_device_count(void);

/* ---- contexts ------------------------------------------------------------------------------ */
/* max_chunks/max_chunk_size size the device workspace (grown on demand when exceeded). */
int  tsx_ctx_create(int device_index, uint32_t max_chunks, uint32_t max_chunk_size, tsx_ctx** out);
void tsx_ctx_destroy(tsx_ctx* ctx);
int  tsx_ctx_timing(const tsx_ctx* ctx, tsx_timing* out);
int  tsx_ctx_device(const tsx_ctx* ctx);                /* device index (0 .. tsx_device_count()-1) the ctx lives on */

/* Device of the calling thread's ctx-less calls: 0 .. tsx_device_count()-1, or -1 = automatic (least loaded).  The JVM side
 * passes  segment hash % devices  so that the chunks of one segment stay on one GPU (segment s -> GPU s mod N). */
int  tsx_set_thread_device(int device_index);
/* Pool of the ctx-less calls on one device: idle contexts kept (at most 32, and at most 128 GiB of device workspace between them),
 * contexts out right now, batches served so far. */
int  tsx_pool_stats(int device_index, uint32_t* idle, uint32_t* in_use, uint64_t* batches);

/* The compressor service of a device (every compressing batch is a member of ONE device-wide queue that persistent waves pull
 * chunks from; csrc/tsx_internal.h).  Counters since tsx_init; kernel_ms is the device's own clock (the last wave of a launch reports its begin and end through pinned memory). */
typedef struct tsx_service_info {
    uint64_t launches;           /* launches of the service kernel that have been started                                       */
    uint64_t watchdog_launches;  /* ... of which a waiting caller made because the kernel had ended with work still queued       */
    uint64_t members, chunks;    /* batches (pieces) published / chunks of completed members                                     */
    double   kernel_ms;          /* summed duration of the launches that have ENDED                                              */
    uint32_t running;            /* 1: a launch is out right now                                                                 */
    uint32_t waves;              /* one-wave workgroups per launch                                                               */
    uint32_t compute_units, cu_keys_seen, reserved_cus;   /* CUs of the device, distinct CU ids a probe launch met, CUs left alone */
    uint32_t device_chunks, wave_starts, reserved_exits, skipped_tickets;   /* device-side counters (mod 2^32)                   */
    uint32_t live_waves, live_waves_max;   /* waves of the service resident right now / the most ever                              */
    uint32_t shader_engines;               /* shader engines the probe launch met (groups of CUs the hardware fills separately)        */
    uint32_t rotations;                    /* launches a fetch that had waited 200 ms asked to end early (the safety net of the fetch side) */
    uint32_t guest_launches;               /* launches whose waves used the reserved CUs too (no fetch had been seen for fetch_quiet_ms)  */
    uint32_t yielded_waves;                /* guest waves that handed their chunk back and left when a fetch arrived                      */
    uint32_t returned_chunks;              /* ... chunks handed back that way (each was started again by another wave)                   */
    uint32_t readmissions;                 /* launches asked to end so that the next one could use the reserved CUs again (quiet again)   */
    uint32_t relocated_waves;              /* compressor waves that found themselves on a reserved CU they had not started on (the hardware's
                                              scheduler saves and restores waves), handed their chunk back and left.  (Occupies what was the
                                              struct's tail padding: its size, 112 bytes, is what it was.)                                  */
} tsx_service_info;
int  tsx_service_stats(int device_index, tsx_service_info* out);
/* Returns when the device's service kernel has ended (a moment after its last chunk): brackets a measurement. */
int  tsx_service_quiesce(int device_index);

/* Pin / unpin a host buffer that is reused for TSX_MEM_HOST(_PACKED) batches (hipHostRegister): optional, see TSX_MEM_HOST. */
int  tsx_host_register(void* p, size_t bytes);
int  tsx_host_unregister(void* p);

/* ---- the hot path -------------------------------------------------------------------------- */
/* Upper bound of the transformed size of an n-byte chunk: ZSTD_compressBound(n) if compressing,
 * +28 (IV 12 + tag 16, EncryptionChunkEnumeration.java:82-84) if encrypting. */
size_t tsx_transformed_bound(size_t n, uint32_t flags);

/* Forward chain over a batch of n chunks: [Zstd frame] -> [IV||AES-256-GCM(C)||TAG], CRC32C(original).
 * Replaces CompressionChunkEnumeration.nextElement + EncryptionChunkEnumeration.nextElement for the
 * whole batch.  src_size / dst_size are the sizes of the caller's buffers: a descriptor whose [src_off, src_off + src_len) or
 * [dst_off, dst_off + dst_cap) reaches beyond them fails the call with TSX_E_INVAL before anything is touched (ABI 3; the chunk
 * bounds of the reference are the array lengths of its byte[] chunks).  ctx == NULL borrows a pooled context: on the device the calling thread chose with tsx_set_thread_device(),
 * else on the initialised device with the fewest batches in flight (one JVM drives all GPUs of the node from >= 10 RLM
 * threads, README.md:218-222). */
int tsx_transform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n,
                        const void* src, size_t src_size, void* dst, size_t dst_size, int mem_kind);

/* Inverse chain: [verify tag + AES-256-GCM decrypt] -> [Zstd decode], CRC32C(restored).
 * Replaces DecryptionChunkEnumeration.nextElement + DecompressionChunkEnumeration.nextElement. */
int tsx_detransform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n,
                          const void* src, size_t src_size, void* dst, size_t dst_size, int mem_kind);

/* CRC32C only (BASELINE.json configs[1]); equals java.util.zip.CRC32C over each chunk. */
int tsx_crc32c_batch(tsx_ctx* ctx, tsx_chunk_desc* descs, uint32_t n, const void* src, size_t src_size, int mem_kind);

/* ---- device memory helpers for hosts that do not link HIP themselves (the JNI shim) --------- */
int tsx_device_malloc(int device_index, size_t bytes, void** out);
int tsx_device_free(int device_index, void* p);
int tsx_memcpy_h2d(int device_index, void* dst_dev, const void* src_host, size_t bytes);
int tsx_memcpy_d2h(int device_index, void* dst_host, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* TSXFORM_H */
