// C-ABI front end of libtsxform: device/context management and the batch pipelines.
// See include/tsxform.h for the contract and the reference call sites each entry point replaces.
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "tsx_internal.h"
#include "zstd_gpu.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { tsx_set_err(#x, e_); return TSX_E_DEVICE; } } while (0)

static thread_local char g_last_err[256];
static void tsx_set_err(const char* what, hipError_t e) {
    snprintf(g_last_err, sizeof g_last_err, "%s failed: %s", what, hipGetErrorString(e));
    if (getenv("TSX_DEBUG")) fprintf(stderr, "[tsxform] %s\n", g_last_err);
}

struct tsx_device {
    int hip_id = -1;
    tsx_crc_tables* d_crc = nullptr;
    tsx_aes_tables* d_aes = nullptr;
    tsx_zstd_consts* d_zc = nullptr;
    char name[256] = {0};
    char arch[256] = {0};
};

struct tsx_ctx {
    int dev_index = 0;
    tsx_device* dev = nullptr;
    hipStream_t st = nullptr;
    // device workspace (grown on demand)
    tsx_chunk_desc* d_descs = nullptr; size_t descs_cap = 0;
    tsx_gcm_chunk* d_gchunks = nullptr;
    int32_t* d_status = nullptr;
    uint32_t* d_zlen = nullptr;
    uint32_t* d_partials = nullptr; size_t partials_cap = 0;
    tsx_gcm_key* d_key = nullptr;
    uint8_t* d_keyraw = nullptr;                 // 32 key + 64 aad
    uint8_t* d_in = nullptr; size_t in_cap = 0;    // staging for TSX_MEM_HOST
    uint8_t* d_out = nullptr; size_t out_cap = 0;
    uint8_t* d_mid = nullptr; size_t mid_cap = 0;  // compressed frames between the Zstd and GCM stages
    size_t mid_stride = 0;
    void* d_zwork = nullptr; size_t zwork_cap = 0; // Zstd per-chunk workspace
    hipEvent_t ev[8] = {nullptr};
    tsx_timing timing{};
    bool pooled = false;
};

static std::mutex g_mu;
static std::vector<tsx_device> g_devs;
static std::vector<tsx_ctx*> g_pool;
static char g_version[384];

extern "C" uint32_t tsx_abi_version(void) { return TSX_ABI_VERSION; }

extern "C" const char* tsx_version(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    snprintf(g_version, sizeof g_version,
             "tsxform 0.1 (gfx950 HIP; CRC32C, AES-256-GCM, Zstd level-3 frames; zstd parity target libzstd 1.5.7 / 1.5.6 profile; device: %s)",
             g_devs.empty() ? "uninitialised" : g_devs[0].name);
    return g_version;
}

extern "C" const char* tsx_strerror(int code) {
    switch (code) {
        case TSX_OK: return "ok";
        case TSX_E_INVAL: return "invalid argument";
        case TSX_E_DEVICE: return g_last_err[0] ? g_last_err : "no usable gfx950 device / HIP failure";
        case TSX_E_NOMEM: return "out of memory";
        case TSX_E_DST_TOO_SMALL: return "destination slot too small";
        case TSX_E_TAG_MISMATCH: return "Tag mismatch";                                    // JCE AEADBadTagException text
        case TSX_E_BAD_FRAME: return "corrupt Zstd frame";
        case TSX_E_BAD_SIZE: return "Invalid decompressed size";                           // DecompressionChunkEnumeration.java:43
        case TSX_E_SHORT_CHUNK: return "encrypted chunk shorter than IV + tag";
        case TSX_E_UNSUPPORTED: return "unsupported parameter";
        default: return "unknown error";
    }
}

extern "C" int tsx_init(int device_count, const int* device_ids) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_devs.empty()) return (int)g_devs.size();
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        snprintf(g_last_err, sizeof g_last_err, "no HIP device visible");
        return TSX_E_DEVICE;
    }
    int want = device_count <= 0 ? visible : device_count;
    if (want > visible) return TSX_E_INVAL;
    tsx_crc_tables* hc = new (std::nothrow) tsx_crc_tables;
    tsx_aes_tables* ha = new (std::nothrow) tsx_aes_tables;
    tsx_zstd_consts* hz = (tsx_zstd_consts*)malloc(tsx_zstd_consts_bytes());
    if (!hc || !ha || !hz) return TSX_E_NOMEM;
    tsx_crc_build_tables(hc);
    tsx_aes_build_tables(ha);
    tsx_zstd_build_consts(hz);
    std::vector<tsx_device> devs;
    for (int i = 0; i < want; i++) {
        tsx_device d;
        d.hip_id = device_ids ? device_ids[i] : i;
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, d.hip_id));
        snprintf(d.name, sizeof d.name, "%s", prop.name);
        snprintf(d.arch, sizeof d.arch, "%s", prop.gcnArchName);
        if (strncmp(d.arch, "gfx950", 6) != 0 && !getenv("TSX_ALLOW_ANY_ARCH")) {
            snprintf(g_last_err, sizeof g_last_err, "device %d is %s, this library is built for gfx950 only", d.hip_id, d.arch);
            return TSX_E_DEVICE;
        }
        HIPCHK(hipSetDevice(d.hip_id));
        HIPCHK(hipMalloc((void**)&d.d_crc, sizeof(tsx_crc_tables)));
        HIPCHK(hipMalloc((void**)&d.d_aes, sizeof(tsx_aes_tables)));
        HIPCHK(hipMalloc((void**)&d.d_zc, tsx_zstd_consts_bytes()));
        HIPCHK(hipMemcpy(d.d_crc, hc, sizeof(tsx_crc_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_aes, ha, sizeof(tsx_aes_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_zc, hz, tsx_zstd_consts_bytes(), hipMemcpyHostToDevice));
        devs.push_back(d);
    }
    delete hc; delete ha; free(hz);
    g_devs = devs;
    return (int)g_devs.size();
}

extern "C" int tsx_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_devs.size();
}

static void ctx_free_device_mem(tsx_ctx* c) {
    hipSetDevice(c->dev->hip_id);
    void* ptrs[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen, c->d_partials, c->d_key, c->d_keyraw, c->d_in, c->d_out, c->d_mid, c->d_zwork};
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    if (c->st) hipStreamDestroy(c->st);
}

extern "C" void tsx_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (tsx_ctx* c : g_pool) { ctx_free_device_mem(c); delete c; }
    g_pool.clear();
    for (auto& d : g_devs) {
        hipSetDevice(d.hip_id);
        if (d.d_crc) hipFree(d.d_crc);
        if (d.d_aes) hipFree(d.d_aes);
        if (d.d_zc) hipFree(d.d_zc);
    }
    g_devs.clear();
}

template <class T>
static int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return TSX_OK;
    if (*p) { if (hipFree(*p) != hipSuccess) return TSX_E_DEVICE; *p = nullptr; *cap = 0; }
    size_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc((void**)p, want * sizeof(T));
    if (e != hipSuccess) { tsx_set_err("hipMalloc(workspace)", e); return TSX_E_NOMEM; }
    *cap = want;
    return TSX_OK;
}

static int ctx_reserve(tsx_ctx* c, uint32_t n, uint32_t max_len, uint32_t flags, bool host_mem, size_t in_bytes, size_t out_bytes) {
    HIPCHK(hipSetDevice(c->dev->hip_id));
    if (n > c->descs_cap || !c->d_descs) {
        void* olds[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen};
        for (void* p : olds) if (p) hipFree(p);
        c->d_descs = nullptr; c->d_gchunks = nullptr; c->d_status = nullptr; c->d_zlen = nullptr;
        size_t cap = (size_t)n + n / 4 + 16;
        HIPCHK(hipMalloc((void**)&c->d_descs, cap * sizeof(tsx_chunk_desc)));
        HIPCHK(hipMalloc((void**)&c->d_gchunks, cap * sizeof(tsx_gcm_chunk)));
        HIPCHK(hipMalloc((void**)&c->d_status, cap * sizeof(int32_t)));
        HIPCHK(hipMalloc((void**)&c->d_zlen, cap * sizeof(uint32_t)));
        c->descs_cap = cap;
    }
    // partials: GCM needs 4 u32 per 64 KiB sub-block of the (possibly expanded) stage input, CRC 1 per 256 KiB
    size_t bound = tsx_transformed_bound(max_len, flags & TSX_COMPRESS) + 64;
    size_t subs = (bound + TSX_GCM_SUB_BYTES - 1) / TSX_GCM_SUB_BYTES + 1;
    int rc = grow(&c->d_partials, &c->partials_cap, (size_t)n * subs * 4);
    if (rc) return rc;
    if (host_mem) {
        if ((rc = grow(&c->d_in, &c->in_cap, in_bytes + 64))) return rc;
        if ((rc = grow(&c->d_out, &c->out_cap, out_bytes + 64))) return rc;
    }
    if (flags & TSX_COMPRESS) {
        size_t stride = (tsx_transformed_bound(max_len, TSX_COMPRESS) + 63) & ~(size_t)63;
        if ((rc = grow(&c->d_mid, &c->mid_cap, stride * n))) return rc;
        c->mid_stride = stride;
        size_t zw = tsx_zstd_workspace_bytes(n, max_len);
        uint8_t* zp = (uint8_t*)c->d_zwork;
        if ((rc = grow(&zp, &c->zwork_cap, zw))) return rc;
        c->d_zwork = zp;
    }
    return TSX_OK;
}

extern "C" int tsx_ctx_create(int device_index, uint32_t max_chunks, uint32_t max_chunk_size, tsx_ctx** out) {
    if (!out) return TSX_E_INVAL;
    tsx_device* dev;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_devs.empty()) { snprintf(g_last_err, sizeof g_last_err, "tsx_init has not succeeded"); return TSX_E_DEVICE; }
        if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
        dev = &g_devs[device_index];
    }
    tsx_ctx* c = new (std::nothrow) tsx_ctx;
    if (!c) return TSX_E_NOMEM;
    c->dev_index = device_index; c->dev = dev;
    HIPCHK(hipSetDevice(dev->hip_id));
    HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipMalloc((void**)&c->d_key, sizeof(tsx_gcm_key)));
    HIPCHK(hipMalloc((void**)&c->d_keyraw, 128));
    if (max_chunks && max_chunk_size) {
        int rc = ctx_reserve(c, max_chunks, max_chunk_size, 0, false, 0, 0);
        if (rc) { ctx_free_device_mem(c); delete c; return rc; }
    }
    *out = c;
    return TSX_OK;
}

extern "C" void tsx_ctx_destroy(tsx_ctx* c) {
    if (!c) return;
    ctx_free_device_mem(c);
    delete c;
}

extern "C" int tsx_ctx_timing(const tsx_ctx* c, tsx_timing* out) {
    if (!c || !out) return TSX_E_INVAL;
    *out = c->timing;
    return TSX_OK;
}

static tsx_ctx* pool_acquire(int* rc) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_pool.empty()) { tsx_ctx* c = g_pool.back(); g_pool.pop_back(); return c; }
    }
    tsx_ctx* c = nullptr;
    *rc = tsx_ctx_create(0, 0, 0, &c);
    if (*rc) return nullptr;
    c->pooled = true;
    return c;
}
static void pool_release(tsx_ctx* c) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_pool.push_back(c);
}

// ZSTD_compressBound(n) = n + (n >> 8) + (n < 128 KiB ? ((128 KiB - n) >> 11) : 0)
extern "C" size_t tsx_transformed_bound(size_t n, uint32_t flags) {
    size_t m = n;
    if (flags & TSX_COMPRESS) m = n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
    if (flags & TSX_ENCRYPT) m += 28;
    return m;
}

// ---- small glue kernels ---------------------------------------------------------------------------
// Builds the GCM work items of a batch from the descriptors (and, after compression, the frame sizes),
// checks slot capacities and initialises status / dst_len.  mode 0 = transform, 1 = detransform.
__global__ void plan_gcm_kernel(tsx_chunk_desc* __restrict__ descs, uint32_t n, const uint32_t* __restrict__ zlen, uint64_t mid_stride,
                                int have_mid, int mode, int mid_is_out, tsx_gcm_chunk* __restrict__ g, int32_t* __restrict__ status) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tsx_chunk_desc d = descs[i];
    tsx_gcm_chunk c;
    int32_t st = status[i];
    if (mode == 0) {
        uint32_t len = have_mid ? zlen[i] : d.src_len;
        c.in_off = have_mid ? (uint64_t)i * mid_stride : d.src_off;
        c.out_off = d.dst_off;
        c.len = len;
        for (int k = 0; k < 12; k++) c.iv[k] = d.iv[k];
        if (st == TSX_OK && (uint64_t)len + 28 > d.dst_cap) st = TSX_E_DST_TOO_SMALL;
        descs[i].dst_len = st == TSX_OK ? len + 28 : 0;
    } else {
        c.in_off = d.src_off;
        c.out_off = mid_is_out ? (uint64_t)i * mid_stride : d.dst_off;
        c.len = d.src_len >= 28 ? d.src_len - 28 : 0;
        for (int k = 0; k < 12; k++) c.iv[k] = 0;
        if (st == TSX_OK && d.src_len < 28) st = TSX_E_SHORT_CHUNK;
        if (st == TSX_OK && !mid_is_out && c.len > d.dst_cap) st = TSX_E_DST_TOO_SMALL;
        if (!mid_is_out) descs[i].dst_len = st == TSX_OK ? c.len : 0;
    }
    c.skip = st != TSX_OK;
    status[i] = st;
    g[i] = c;
}

// dst slot <- src span, 16 B per lane when both sides are 16-byte aligned (chunk offsets must be).
// from_mid: source is the compressed-frame staging buffer with per-chunk length zlen[i].
__global__ __launch_bounds__(256) void copy_chunks_kernel(tsx_chunk_desc* __restrict__ descs, const uint32_t* __restrict__ zlen,
                                                          uint64_t mid_stride, int from_mid, const uint8_t* __restrict__ src,
                                                          uint8_t* __restrict__ dst, int32_t* __restrict__ status, uint32_t blocks_per_chunk) {
    const uint32_t i = blockIdx.x / blocks_per_chunk, part = blockIdx.x % blocks_per_chunk;
    const tsx_chunk_desc d = descs[i];
    if (status[i] != TSX_OK) return;
    const uint32_t len = from_mid ? zlen[i] : d.src_len;
    if (len > d.dst_cap) {
        if (part == 0 && threadIdx.x == 0) { status[i] = TSX_E_DST_TOO_SMALL; descs[i].dst_len = 0; }
        return;
    }
    const uint8_t* s = src + (from_mid ? (uint64_t)i * mid_stride : d.src_off);
    uint8_t* o = dst + d.dst_off;
    const uint32_t q = len >> 4;
    for (uint32_t p = part * 256 + threadIdx.x; p < q; p += blocks_per_chunk * 256)
        reinterpret_cast<uint4*>(o)[p] = reinterpret_cast<const uint4*>(s)[p];
    if (part == 0) {
        for (uint32_t b = (q << 4) + threadIdx.x; b < len; b += 256) o[b] = s[b];
        if (threadIdx.x == 0) descs[i].dst_len = len;
    }
}

__global__ void init_status_kernel(int32_t* status, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[i] = TSX_OK;
}

__global__ void publish_status_kernel(tsx_chunk_desc* descs, const int32_t* status, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    descs[i].status = status[i];
    if (status[i] != TSX_OK) descs[i].dst_len = 0;
}

// ---- batch drivers ---------------------------------------------------------------------------------
static int validate(const tsx_chunk_desc* descs, uint32_t n, size_t dst_size, bool need_dst, uint32_t* max_len, size_t* in_bytes) {
    *max_len = 0; *in_bytes = 0;
    for (uint32_t i = 0; i < n; i++) {
        const tsx_chunk_desc& d = descs[i];
        if ((d.src_off & 15) || (need_dst && (d.dst_off & 15))) return TSX_E_INVAL;       // 16-byte aligned slots
        if (need_dst && (d.dst_off + d.dst_cap > dst_size)) return TSX_E_INVAL;
        if (d.src_len >= (1u << 30) + 4096) return TSX_E_INVAL;                             // chunk.size <= 2^30 - 1 (RemoteStorageManagerConfig.java:122-130)
        if (d.src_len > *max_len) *max_len = d.src_len;
        if (d.src_off + d.src_len > *in_bytes) *in_bytes = d.src_off + d.src_len;
    }
    return TSX_OK;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }

static int run_batch(tsx_ctx* c, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src, void* dst,
                     size_t dst_size, int mem_kind, int mode /*0 transform, 1 detransform, 2 crc only*/) {
    if (!descs || (n && !src) || (mode != 2 && (!params || (n && !dst)))) return TSX_E_INVAL;
    if (mem_kind != TSX_MEM_HOST && mem_kind != TSX_MEM_DEVICE && mem_kind != TSX_MEM_HOST_PACKED) return TSX_E_INVAL;
    const bool packed = mem_kind == TSX_MEM_HOST_PACKED;
    if (packed && mode != 0) return TSX_E_INVAL;
    const uint32_t flags = mode == 2 ? TSX_CRC : params->flags;
    if (flags & ~(TSX_COMPRESS | TSX_ENCRYPT | TSX_CRC)) return TSX_E_INVAL;
    if (mode != 2) {
        if (params->aad_len > 64) return TSX_E_INVAL;
        if ((flags & TSX_COMPRESS) && !(params->zstd_level == 0 || params->zstd_level == 3)) return TSX_E_UNSUPPORTED;
        if ((flags & TSX_COMPRESS) && params->zstd_profile > TSX_ZSTD_PROFILE_1_5_7) return TSX_E_UNSUPPORTED;
    }
    if (n == 0) return TSX_OK;
    uint32_t max_len; size_t in_bytes;
    size_t out_bytes = dst_size;                                        // size of the output area the kernels see
    if (packed) {
        // the kernels still write one bound-sized slot per chunk - on the device; only the bytes produced cross PCIe, straight
        // to their final place in the caller's buffer
        uint32_t longest = 0;
        for (uint32_t i = 0; i < n; i++) if (descs[i].src_len > longest) longest = descs[i].src_len;
        const size_t slot = (tsx_transformed_bound(longest, flags) + 63) & ~(size_t)63;
        if (slot >= ((size_t)1 << 32)) return TSX_E_INVAL;
        for (uint32_t i = 0; i < n; i++) { descs[i].dst_off = (uint64_t)i * slot; descs[i].dst_cap = (uint32_t)slot; }
        out_bytes = (size_t)n * slot;
    }
    int rc = validate(descs, n, out_bytes, mode != 2, &max_len, &in_bytes);
    if (rc) return rc;
    const bool host = mem_kind != TSX_MEM_DEVICE;
    rc = ctx_reserve(c, n, max_len, flags, host, in_bytes, out_bytes);
    if (rc) return rc;
    hipStream_t st = c->st;
    bool fused = false;
    memset(&c->timing, 0, sizeof c->timing);
    const uint8_t* d_src = (const uint8_t*)src;
    uint8_t* d_dst = (uint8_t*)dst;
    HIPCHK(hipEventRecord(c->ev[0], st));
    if (host) {
        HIPCHK(hipMemcpyAsync(c->d_in, src, in_bytes, hipMemcpyHostToDevice, st));
        d_src = c->d_in; d_dst = c->d_out;
    }
    HIPCHK(hipMemcpyAsync(c->d_descs, descs, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(init_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_status, n);
    HIPCHK(hipEventRecord(c->ev[1], st));
    const bool enc = flags & TSX_ENCRYPT, comp = flags & TSX_COMPRESS;
    if (enc) {
        HIPCHK(hipMemcpyAsync(c->d_keyraw, params->key, 32, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(c->d_keyraw + 32, params->aad, 64, hipMemcpyHostToDevice, st));
        tsx_launch_gcm_setup(st, c->dev->d_aes, c->d_keyraw, c->d_keyraw + 32, params->aad_len, c->d_key);
    }
    if (mode == 2) {
        tsx_launch_crc32c(st, c->dev->d_crc, d_src, c->d_descs, n, max_len, c->d_partials, 0);
        HIPCHK(hipEventRecord(c->ev[2], st));
        c->timing.crc_launches = 2;
    } else if (mode == 0) {
        // With compression the whole chain of a chunk runs in the wave that compresses it: CRC32C of the source chunk first, GCM over
        // the finished frame last - one launch per batch.  The batch CRC / GCM kernels want 20 / 40 KiB of LDS per workgroup and, on
        // a chip filled by the compressor waves of the batches in flight, sat hundreds of ms in the queue for a few ms of work.
        // TSX_STAGES_SEPARATE=1 keeps one launch per stage (A/B measurements, tests of the stand-alone kernels).
        const bool fuse_stages = comp && !getenv("TSX_STAGES_SEPARATE");
        if ((flags & TSX_CRC) && !fuse_stages) { tsx_launch_crc32c(st, c->dev->d_crc, d_src, c->d_descs, n, max_len, c->d_partials, 0); c->timing.crc_launches = 2; }
        HIPCHK(hipEventRecord(c->ev[2], st));
        if (comp) {
            fused = enc && fuse_stages;
            tsx_chain_fuse fuse{nullptr, nullptr, nullptr, nullptr};
            if (fuse_stages && (flags & TSX_CRC)) fuse.crc = c->dev->d_crc;
            if (fused) { fuse.aes = c->dev->d_aes; fuse.key = c->d_key; fuse.out = d_dst; }
            c->timing.zstd_launches = tsx_launch_zstd_compress(st, c->dev->d_zc, d_src, c->d_descs, n, max_len, c->d_mid, c->mid_stride,
                                                              c->d_zlen, c->d_status, c->d_zwork, params->zstd_profile, fuse);
        }
        HIPCHK(hipEventRecord(c->ev[3], st));
        if (fused) {
        } else if (enc) {
            hipLaunchKernelGGL(plan_gcm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_descs, n, (const uint32_t*)c->d_zlen,
                               (uint64_t)c->mid_stride, comp ? 1 : 0, 0, 0, c->d_gchunks, c->d_status);
            uint32_t glen = comp ? (uint32_t)tsx_transformed_bound(max_len, TSX_COMPRESS) : max_len;
            tsx_launch_gcm(st, c->dev->d_aes, c->d_key, c->d_gchunks, n, glen, comp ? c->d_mid : d_src, d_dst, c->d_partials, c->d_status, 0);
            c->timing.gcm_launches = 2;
        } else {
            uint32_t bpc = max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, c->d_descs, (const uint32_t*)c->d_zlen,
                               (uint64_t)c->mid_stride, comp ? 1 : 0, comp ? (const uint8_t*)c->d_mid : d_src, d_dst, c->d_status, bpc);
        }
        HIPCHK(hipEventRecord(c->ev[4], st));
    } else {
        HIPCHK(hipEventRecord(c->ev[2], st));
        const uint8_t* zsrc = d_src;      // where the Zstd frames live when there is no encryption
        if (enc) {
            hipLaunchKernelGGL(plan_gcm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_descs, n, (const uint32_t*)c->d_zlen,
                               (uint64_t)c->mid_stride, 0, 1, comp ? 1 : 0, c->d_gchunks, c->d_status);
            tsx_launch_gcm(st, c->dev->d_aes, c->d_key, c->d_gchunks, n, max_len, d_src, comp ? c->d_mid : d_dst, c->d_partials, c->d_status, 1);
            c->timing.gcm_launches = 2;
            zsrc = c->d_mid;
        }
        HIPCHK(hipEventRecord(c->ev[3], st));
        if (comp) {
            c->timing.unzstd_launches = tsx_launch_zstd_decompress(st, c->dev->d_zc, zsrc, enc ? 1 : 0, (uint64_t)c->mid_stride, c->d_descs, n,
                                                                  d_dst, c->d_status, c->d_zwork);
        } else if (!enc) {
            uint32_t bpc = max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, c->d_descs, (const uint32_t*)c->d_zlen,
                               (uint64_t)0, 0, d_src, d_dst, c->d_status, bpc);
        }
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_descs, (const int32_t*)c->d_status, n);
        HIPCHK(hipEventRecord(c->ev[4], st));
        if (flags & TSX_CRC) {
            // CRC of the restored bytes; upper bound of a restored chunk is its slot capacity
            uint32_t max_out = 0;
            for (uint32_t i = 0; i < n; i++) if (descs[i].dst_cap > max_out) max_out = descs[i].dst_cap;
            tsx_launch_crc32c(st, c->dev->d_crc, d_dst, c->d_descs, n, max_out, c->d_partials, 1);
            c->timing.crc_launches = 2;
        }
    }
    if (mode != 1) hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, c->d_descs, (const int32_t*)c->d_status, n);
    HIPCHK(hipEventRecord(c->ev[5], st));
    HIPCHK(hipMemcpyAsync(descs, c->d_descs, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyDeviceToHost, st));
    if (packed) {
        HIPCHK(hipStreamSynchronize(st));                               // sizes first: they say where each chunk goes
        size_t at = 0; bool full = false;
        for (uint32_t i = 0; i < n; i++) {
            const size_t slot_off = descs[i].dst_off;
            descs[i].dst_off = at;
            if (descs[i].status != TSX_OK) { descs[i].dst_len = 0; continue; }
            if (full || at + descs[i].dst_len > dst_size) { full = true; descs[i].status = TSX_E_DST_TOO_SMALL; descs[i].dst_len = 0; continue; }
            if (descs[i].dst_len) HIPCHK(hipMemcpyAsync((uint8_t*)dst + at, c->d_out + slot_off, descs[i].dst_len, hipMemcpyDeviceToHost, st));
            at += descs[i].dst_len;
        }
    } else if (host && mode != 2) {
        // only the bytes each chunk produced travel back (a compressed chunk fills ~1/3 of its slot): the descriptors
        // first, then one copy per run of chunks whose outputs are adjacent in dst
        HIPCHK(hipStreamSynchronize(st));
        uint32_t i = 0;
        while (i < n) {
            if (descs[i].status != TSX_OK || descs[i].dst_len == 0) { i++; continue; }
            size_t lo = descs[i].dst_off, hi = lo + descs[i].dst_len;
            uint32_t j = i + 1;
            while (j < n && descs[j].status == TSX_OK && descs[j].dst_off >= hi && descs[j].dst_off - hi <= 4096) { hi = descs[j].dst_off + descs[j].dst_len; j++; }
            HIPCHK(hipMemcpyAsync((uint8_t*)dst + lo, c->d_out + lo, hi - lo, hipMemcpyDeviceToHost, st));
            i = j;
        }
    }
    HIPCHK(hipEventRecord(c->ev[6], st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    tsx_timing& t = c->timing;
    t.h2d_ms = ev_ms(c->ev[0], c->ev[1]);
    t.d2h_ms = ev_ms(c->ev[5], c->ev[6]);
    t.total_ms = ev_ms(c->ev[0], c->ev[6]);
    if (mode == 2) t.crc_ms = ev_ms(c->ev[1], c->ev[2]);
    else if (mode == 0) {
        t.crc_ms = (flags & TSX_CRC) ? ev_ms(c->ev[1], c->ev[2]) : 0;
        t.zstd_ms = comp ? ev_ms(c->ev[2], c->ev[3]) : 0;
        t.gcm_ms = ev_ms(c->ev[3], c->ev[4]);
    } else {
        t.gcm_ms = enc ? ev_ms(c->ev[2], c->ev[3]) : 0;
        t.unzstd_ms = ev_ms(c->ev[3], c->ev[4]);
        t.crc_ms = (flags & TSX_CRC) ? ev_ms(c->ev[4], c->ev[5]) : 0;
    }
    return TSX_OK;
}

static int with_ctx(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src, void* dst,
                    size_t dst_size, int mem_kind, int mode) {
    if (ctx) return run_batch(ctx, params, descs, n, src, dst, dst_size, mem_kind, mode);
    int rc = TSX_OK;
    tsx_ctx* c = pool_acquire(&rc);
    if (!c) return rc;
    rc = run_batch(c, params, descs, n, src, dst, dst_size, mem_kind, mode);
    pool_release(c);
    return rc;
}

extern "C" int tsx_transform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src,
                                   void* dst, size_t dst_size, int mem_kind) {
    return with_ctx(ctx, params, descs, n, src, dst, dst_size, mem_kind, 0);
}

extern "C" int tsx_detransform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src,
                                     void* dst, size_t dst_size, int mem_kind) {
    return with_ctx(ctx, params, descs, n, src, dst, dst_size, mem_kind, 1);
}

extern "C" int tsx_crc32c_batch(tsx_ctx* ctx, tsx_chunk_desc* descs, uint32_t n, const void* src, int mem_kind) {
    return with_ctx(ctx, nullptr, descs, n, src, nullptr, 0, mem_kind, 2);
}

// ---- device memory helpers -------------------------------------------------------------------------
static int set_dev(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    return hipSetDevice(g_devs[device_index].hip_id) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_device_malloc(int device_index, size_t bytes, void** out) {
    if (!out) return TSX_E_INVAL;
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMalloc(out, bytes) == hipSuccess ? TSX_OK : TSX_E_NOMEM;
}
extern "C" int tsx_device_free(int device_index, void* p) {
    int rc = set_dev(device_index); if (rc) return rc;
    return hipFree(p) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_memcpy_h2d(int device_index, void* d, const void* s, size_t bytes) {
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_memcpy_d2h(int device_index, void* d, const void* s, size_t bytes) {
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
