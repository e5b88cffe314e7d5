// C-ABI front end of libtsxform: device/context management and the batch pipelines.
// See include/tsxform.h for the contract and the reference call sites each entry point replaces.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
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

// Every entry point that selects a device puts the calling thread's current device back on the way out: the caller may share
// the thread with another HIP user (a torch process, another JNI library) whose notion of "current device" is not ours to change.
struct tsx_device_scope {
    int prev = -1;
    tsx_device_scope() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~tsx_device_scope() { if (prev >= 0) (void)hipSetDevice(prev); }
    tsx_device_scope(const tsx_device_scope&) = delete;
    tsx_device_scope& operator=(const tsx_device_scope&) = delete;
};

struct tsx_ctx;
struct tsx_run;

// ---- launch combiner (ctx-less compressing batches) ----------------------------------------------------------------------------
// The HIP runtime multiplexes a process's streams onto a handful of hardware queues (4 unless GPU_MAX_HW_QUEUES says otherwise),
// and two streams that share one run their kernels one after the other.  A broker drives the forward chain from >= 10 threads, one
// 256-chunk segment each (reference README.md:218-222): with a stream per caller, measured on MI355X, 10-24 callers moved 5.5-5.9
// GiB/s where three 2048-chunk batches move 19 - four kernels at a time, 1024 chunks on a chip that holds 6144 - and the copy
// streams' event markers sat behind other callers' second-long kernels.  So ctx-less compressing batches do not get streams of
// their own: per device there are TSX_LANES compute streams and two copy streams, and whoever arrives while all lanes are busy joins
// the group that the next free lane launches as ONE kernel (zstd_compress_kernel's segment table: workgroup -> caller's buffers).
// Group commit: the first waiting caller leads - it queues every member's descriptor upload, key schedule, the one compressor
// launch, every member's status publication and descriptor download on the lane - the others wait for their own completion event.
#define TSX_LANES_MAX 24
#define TSX_COPY_STREAMS_MAX 8
#define TSX_GROUP_MAX_SEGS 64
#define TSX_GROUP_MAX_CHUNKS 8192
struct tsx_zreq { tsx_ctx* c; tsx_run* r; hipEvent_t in_ready; int rc; bool done; };
struct tsx_lane { hipStream_t st = nullptr; hipEvent_t end = nullptr; bool busy = false; tsx_zseg* h_segs = nullptr; tsx_zseg* d_segs = nullptr; /* pinned segment table: host view, device alias */ };
struct tsx_combiner {
    std::mutex mu; std::condition_variable cv;
    std::vector<tsx_zreq*> pending; bool leader = false;
    tsx_lane lane[TSX_LANES_MAX]; uint32_t nlanes = 0;
    // copy streams of the context-less path, shared by the callers (created before the lanes, so that their event markers get hardware queues
    // of their own): a call takes one input and one output stream in turn.  ONE of each by default: with 48 callers an output copy phase of
    // 330 MB stood 1.05 s in that queue - and more streams made it worse (they collide with the lanes' hardware queues: 2 + 4 streams -5 %,
    // 4 + 8 -30 %, profiles/r04_broker_shape_experiments.txt); what removed the phase is zero-copy output (run_combined).  TSX_COPY_STREAMS="in,out".
    hipStream_t copy_in_s[TSX_COPY_STREAMS_MAX] = {nullptr}, copy_out_s[TSX_COPY_STREAMS_MAX] = {nullptr};
    uint32_t n_in = 1, n_out = 1;
    std::atomic<uint32_t> rr_in{0}, rr_out{0};
    uint64_t groups = 0, members = 0;                      // launches made, batches they carried (tsx_debug_combiner_stats)
    // Admission cap (opt-in, TSX_COMBINER_MAX_CHUNKS; 0 = none): compressor chunks launched and not yet done on this device.  With more
    // chunks launched than the chip has slots (6144) every freed slot is refilled by the hardware at once and anything that is not a
    // compressor wave starves (mixed load); members are 256 chunks and complete one by one, so holding launches back at ~5600
    // keeps a few hundred slots turning over in the open.  One device run (32 callers, cap 5632, profiles/r04_mixed_load.txt): a fetch still
    // takes 1.8-3.4 s - scattered free slots rarely line up three on one CU - so this is a knob, off by default; TSX_FETCH_RESERVED_CUS is the remedy.
    uint32_t inflight = 0, inflight_peak = 0, cap = 0;
};

struct tsx_device {
    int hip_id = -1;
    tsx_crc_tables* d_crc = nullptr;
    tsx_aes_tables* d_aes = nullptr;
    tsx_zstd_consts* d_zc = nullptr;
    uint8_t* h_zeros = nullptr;                    // pinned sizeof(tsx_gcm_key) zero bytes: key material is wiped by COPYING zeros - a memset is a
                                                   // kernel, and a kernel waits for a slot on a chip full of compressor waves (measured: 52 ms on
                                                   // average, up to 489 ms, per wipe: profiles/r03_bench_rocprofv3_kernel_stats_before_zero_copy_wipes.csv)
    char name[256] = {0};
    char arch[256] = {0};
    // pooled contexts of the ctx-less calls: idle ones, how many are out, batches served (all under g_mu)
    std::vector<tsx_ctx*> idle;
    size_t idle_bytes = 0;
    size_t idle_cap = 0;                                     // most idle workspace kept (init_devices: a fraction of THIS device's memory)
    std::unique_ptr<tsx_combiner> comb;                      // created with the first ctx-less compressing batch
    uint32_t in_use = 0;
    uint64_t batches = 0;
};

#define TSX_MAX_SUBS 64                 /* sub-batches of one host-memory batch (staging pipeline) */
#define TSX_SUB_BYTES ((size_t)64 << 20) /* input bytes per sub-batch: >= 1000 workgroups of the GCM / CRC kernels */
#define TSX_POOL_MAX_IDLE 32            /* idle pooled contexts kept per device (a broker: >= 10 RLM threads + read-ahead helpers + the fetch pool) ... */
// ... as long as their workspaces together stay under tsx_device.idle_cap = 4/9 of the device's memory (128 of the MI355X's 288 GB; a smaller
// device or several processes per GPU get their share: TSX_POOL_IDLE_BYTES overrides); the rest are destroyed on release, and an
// allocation that fails drains the idle pool and is tried again (reserve_or_drain) - cached memory is never the reason for TSX_E_NOMEM.
#define TSX_POOL_MAX_IDLE_BWORK 4       /* idle contexts that keep their block-form decoder workspace (37 MiB per 4 MiB chunk: 9.4 GiB for a segment) */
#define TSX_COMP_PIECES 4               /* pieces of a host-memory batch on the compress path: one compute stream each (they must co-reside) */

struct tsx_ctx {
    int dev_index = 0;
    tsx_device* dev = nullptr;
    hipStream_t st = nullptr;                      // kernels (+ descriptor copies)
    hipStream_t st_in = nullptr, st_out = nullptr; // H2D / D2H of the host-memory staging pipeline
    hipStream_t st_out2 = nullptr;                 // second D2H stream of a fetch cut into pieces (odd pieces; created on first use)
    hipStream_t st_pc[TSX_COMP_PIECES - 1] = {nullptr}; // compute streams of pieces 1.. of a host batch cut into co-resident pieces (created on first use)
    hipStream_t st_fwd = nullptr;                  // compressing batches with CUs reserved for everything else (tsx_compressor_stream): their own stream ...
    hipStream_t st_pcf[TSX_COMP_PIECES - 1] = {nullptr}; // ... and their pieces' (created on first use)
    hipEvent_t ev_key = nullptr;                   // key schedule ready (the piece streams wait for it)
    // device workspace (grown on demand)
    tsx_chunk_desc* d_descs = nullptr; size_t descs_cap = 0;
    tsx_chunk_desc* h_descs = nullptr;             // pinned mirror of the descriptors: no pageable copy ever sits in a stream
    tsx_chunk_desc* hd_descs = nullptr;            // ... as the device addresses it: lean compressing batches read and write it in place
    uint8_t* h_keyraw = nullptr;                   // pinned 128 bytes: key + aad on their way in (wiped after the batch)
    tsx_gcm_key* h_key = nullptr;                  // pinned: the key schedule built on the host (compressing batches; wiped after the batch)
    tsx_gcm_key* hd_key = nullptr;                 // ... as the device addresses it (every compressor wave takes its own copy, tsx_chain_fuse.key_on_host)
    uint32_t* d_segdone = nullptr;                 // combined launches: chunks of this context's batch that are done (device counter, self-resetting)
    uint32_t* h_segflag = nullptr;                 // ... and the word the last of them raises (pinned; hd_segflag = the device's address of it)
    uint32_t* hd_segflag = nullptr;
    tsx_gcm_chunk* d_gchunks = nullptr;
    int32_t* d_status = nullptr;
    uint32_t* d_zlen = nullptr;
    uint32_t* d_partials = nullptr; size_t partials_cap = 0; size_t partials_per_chunk = 0;   // (pieces that run side by side take their own slice)
    uint32_t last_max_out = 0;
    std::vector<std::pair<uint32_t, uint32_t>> blk_pieces;  // (first chunk, chunks) of every block-form decoder launch of the last batch
    tsx_gcm_key* d_key = nullptr;
    uint8_t* d_keyraw = nullptr;                 // 32 key + 64 aad
    uint8_t* d_in = nullptr; size_t in_cap = 0;    // staging for TSX_MEM_HOST
    uint8_t* d_out = nullptr; size_t out_cap = 0;
    uint8_t* d_mid = nullptr; size_t mid_cap = 0;  // compressed frames between the Zstd and GCM stages
    size_t mid_stride = 0;
    void* d_zwork = nullptr; size_t zwork_cap = 0; // Zstd per-chunk workspace
    void* d_bwork = nullptr; size_t bwork_cap = 0; // block-parallel frame decoder (small batches): chunk headers + literal / sequence arenas
    hipEvent_t ev[4] = {nullptr};                  // batch begin / end, first H2D, last D2H
    hipEvent_t sub_ev[TSX_MAX_SUBS][6] = {{nullptr}}; // per sub-batch: stage boundaries 0..4 (st), [5] = staged in (st_in)
    tsx_timing timing{};
    bool pooled = false;
    bool last_used_blocks = false;                 // the last batch ran the block-parallel frame decoder (test hook)
};

static std::mutex g_mu;
static std::vector<tsx_device> g_devs;
static uint32_t g_rr = 0;
static thread_local int t_dev_hint = -1;
static const char kUninitVersion[] = "tsxform 0.4 (gfx950 HIP; uninitialised)";
static char g_version_buf[2][512];
static unsigned g_version_gen = 0;
static std::atomic<const char*> g_version{kUninitVersion};

extern "C" uint32_t tsx_abi_version(void) { return TSX_ABI_VERSION; }

// The string is composed in a buffer no reader can see yet and published with one pointer store at the end of a successful
// tsx_init: callers never observe a half-written string and need no lock.
extern "C" const char* tsx_version(void) { return g_version.load(std::memory_order_acquire); }

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

static void device_free_consts(tsx_device& d) {
    if (d.hip_id < 0) return;
    hipSetDevice(d.hip_id);
    if (d.comb) {
        for (uint32_t i = 0; i < d.comb->nlanes; i++) {
            tsx_lane& l = d.comb->lane[i];
            if (l.st) { hipStreamSynchronize(l.st); hipStreamDestroy(l.st); }
            if (l.end) hipEventDestroy(l.end);
            if (l.h_segs) hipHostFree(l.h_segs);
        }
        for (auto& q : d.comb->copy_in_s) if (q) hipStreamDestroy(q);
        for (auto& q : d.comb->copy_out_s) if (q) hipStreamDestroy(q);
        d.comb.reset();
    }
    if (d.d_crc) hipFree(d.d_crc);
    if (d.d_aes) hipFree(d.d_aes);
    if (d.d_zc) hipFree(d.d_zc);
    if (d.h_zeros) hipHostFree(d.h_zeros);
    d.d_crc = nullptr; d.d_aes = nullptr; d.d_zc = nullptr; d.h_zeros = nullptr;
}

static int init_devices(std::vector<tsx_device>& devs, int want, const int* device_ids, const tsx_crc_tables* hc, const tsx_aes_tables* ha,
                        const tsx_zstd_consts* hz) {
    for (int i = 0; i < want; i++) {
        devs.emplace_back();
        tsx_device& d = devs.back();
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
        {   size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) { (void)hipGetLastError(); total_b = prop.totalGlobalMem; }
            d.idle_cap = total_b / 9 * 4;
            if (const char* e = getenv("TSX_POOL_IDLE_BYTES")) { const long long v = atoll(e); if (v >= 0) d.idle_cap = (size_t)v; }
        }
        HIPCHK(hipMalloc((void**)&d.d_crc, sizeof(tsx_crc_tables)));
        HIPCHK(hipMalloc((void**)&d.d_aes, sizeof(tsx_aes_tables)));
        HIPCHK(hipMalloc((void**)&d.d_zc, tsx_zstd_consts_bytes()));
        HIPCHK(hipHostMalloc((void**)&d.h_zeros, sizeof(tsx_gcm_key), hipHostMallocDefault));
        memset(d.h_zeros, 0, sizeof(tsx_gcm_key));
        HIPCHK(hipMemcpy(d.d_crc, hc, sizeof(tsx_crc_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_aes, ha, sizeof(tsx_aes_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_zc, hz, tsx_zstd_consts_bytes(), hipMemcpyHostToDevice));
    }
    return TSX_OK;
}

extern "C" int tsx_init(int device_count, const int* device_ids) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_devs.empty()) return (int)g_devs.size();
    tsx_device_scope keep;
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
    int rc = TSX_E_NOMEM;
    std::vector<tsx_device> devs;
    devs.reserve((size_t)want);
    if (hc && ha && hz) {
        tsx_crc_build_tables(hc);
        tsx_aes_build_tables(ha);
        tsx_zstd_build_consts(hz);
        rc = init_devices(devs, want, device_ids, hc, ha, hz);
    }
    delete hc; delete ha; free(hz);
    if (rc != TSX_OK) { for (auto& d : devs) device_free_consts(d); return rc; }   // nothing of a failed init stays allocated
    g_devs.swap(devs);
    char* vb = g_version_buf[g_version_gen++ & 1];
    snprintf(vb, sizeof g_version_buf[0],
             "tsxform 0.4 (gfx950 HIP; CRC32C, AES-256-GCM, Zstd level-3 frames; zstd parity target libzstd 1.5.7 / 1.5.6 profile; %d device(s): %s)",
             (int)g_devs.size(), g_devs[0].name);
    g_version.store(vb, std::memory_order_release);
    return (int)g_devs.size();
}

extern "C" int tsx_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_devs.size();
}

static void ctx_free_device_mem(tsx_ctx* c) {
    hipSetDevice(c->dev->hip_id);
    // the key schedule and the raw key never outlive the context in readable form
    if (c->d_key) hipMemset(c->d_key, 0, sizeof(tsx_gcm_key));
    if (c->d_keyraw) hipMemset(c->d_keyraw, 0, 128);
    void* ptrs[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen, c->d_partials, c->d_key, c->d_keyraw, c->d_in, c->d_out, c->d_mid, c->d_zwork, c->d_bwork, c->d_segdone};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->h_segflag) hipHostFree(c->h_segflag);
    if (c->h_descs) hipHostFree(c->h_descs);
    if (c->h_keyraw) { memset(c->h_keyraw, 0, 128); hipHostFree(c->h_keyraw); }
    if (c->h_key) { memset(c->h_key, 0, sizeof(tsx_gcm_key)); hipHostFree(c->h_key); }
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    for (auto& row : c->sub_ev) for (auto& e : row) if (e) hipEventDestroy(e);
    if (c->st) hipStreamDestroy(c->st);
    if (c->st_in) hipStreamDestroy(c->st_in);
    if (c->st_out) hipStreamDestroy(c->st_out);
    if (c->st_out2) hipStreamDestroy(c->st_out2);
    for (auto& q : c->st_pc) if (q) hipStreamDestroy(q);
    for (auto& q : c->st_pcf) if (q) hipStreamDestroy(q);
    if (c->st_fwd) hipStreamDestroy(c->st_fwd);
    if (c->ev_key) hipEventDestroy(c->ev_key);
}

extern "C" void tsx_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    tsx_device_scope keep;
    for (auto& d : g_devs) {
        for (tsx_ctx* c : d.idle) { ctx_free_device_mem(c); delete c; }
        d.idle.clear(); d.idle_bytes = 0;
        device_free_consts(d);
    }
    g_devs.clear();
    g_version.store(kUninitVersion, std::memory_order_release);
}

// A chip full of compressor waves starves everything else: a freed slot (one wave, 6.7 KB of LDS) is taken at once by the next queued
// compressor workgroup, while a decoder workgroup (2-8 waves, up to 19.5 KB of LDS) needs several neighbouring slots free at the same
// time - measured with five callers keeping 10 240 chunks queued: a single-chunk fetch took 64 s instead of 1.6 ms
// (profiles/r04_mixed_load.txt).  A broker that uploads and serves fetches from the same device can ask for a reservation:
// TSX_FETCH_RESERVED_CUS=n creates the COMPRESSOR's streams (the combiner's lanes, a context's stream for compressing batches) with a CU
// mask that leaves n compute units - one per XCD for n = 8: the mask's bit i belongs to XCD i mod 8 - to whatever else runs.  Measured
// with n = 8: a fetch under full upload load 3.3 ms median (p95 0.8 s; n = 16: 3.4 ms, p95 64 ms) instead of 50-80 s.  The price is not
// the 3 % of the CUs: kernels on masked queues overlap worse - 14.5-15.2 GiB/s with five batches in flight against 19.4-19.8 (a lone batch:
// 10.0 against 10.45), whatever GPU_MAX_HW_QUEUES says - so the default is 0, no reservation, and a deployment that serves consumers from
// tiered storage while it uploads chooses (or gives fetches a device of their own: tsx_set_thread_device).  (Queue priority is no remedy:
// with the fetch context's stream at the highest priority the same fetch still took 40 s - what is missing is room, not turn.)
static uint32_t reserved_cus() {
    static const uint32_t v = [] { const char* e = getenv("TSX_FETCH_RESERVED_CUS"); const long x = e ? atol(e) : 0; return (uint32_t)(x < 0 ? 0 : x > 128 ? 128 : x); }();
    return v;
}
static hipError_t tsx_compressor_stream(hipStream_t* out, int hip_device) {
    const uint32_t r = reserved_cus();
    hipDeviceProp_t prop;
    if (r == 0 || hipGetDeviceProperties(&prop, hip_device) != hipSuccess || prop.multiProcessorCount <= (int)r + 8) {
        (void)hipGetLastError();
        return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    }
    const uint32_t cus = (uint32_t)prop.multiProcessorCount, words = (cus + 31) / 32;
    uint32_t mask[64] = {0};
    for (uint32_t i = 0; i + r < cus && i < 64 * 32; i++) mask[i / 32] |= 1u << (i % 32);      // every CU but the last r of the numbering
    const hipError_t e = hipExtStreamCreateWithCUMask(out, words, mask);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);        // a runtime without CU masks: no reservation, as before
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

// Small detransform batches (a fetch: one chunk, a prefetch window) decode one workgroup per BLOCK instead of per chunk: the
// chunk-serial decoder needs 25-50 ms for a chunk however idle the chip is.  TSX_DEC_BLOCK_CHUNKS: largest batch that takes this
// form (default 256, 0 = never).
static uint32_t dec_block_chunks() {
    if (const char* e = getenv("TSX_DEC_BLOCK_CHUNKS")) { const long v = atol(e); return v < 0 ? 0u : (uint32_t)v; }
    return 256u;                                            // measured: 27.6 ms at 256 chunks against the chunk form's 33, profiles/r03_dec_latency_block_form.jsonl
}
static bool dec_use_blocks(uint32_t n, uint32_t max_out) { return n <= dec_block_chunks() && tsx_zstd_blockmode_takes(max_out); }

// max_out: largest output slot of the batch (detransform: the CRC of the restored bytes runs over dst_cap-sized slots)
static int ctx_reserve(tsx_ctx* c, uint32_t n, uint32_t max_len, uint32_t max_out, uint32_t flags, bool host_mem, size_t in_bytes, size_t out_bytes) {
    HIPCHK(hipSetDevice(c->dev->hip_id));
    if (n > c->descs_cap || !c->d_descs) {
        void* olds[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen};
        for (void* p : olds) if (p) hipFree(p);
        if (c->h_descs) hipHostFree(c->h_descs);
        c->d_descs = nullptr; c->d_gchunks = nullptr; c->d_status = nullptr; c->d_zlen = nullptr; c->h_descs = nullptr; c->hd_descs = nullptr;
        c->descs_cap = 0;                                    // a failure below leaves a context that reallocates, not one with holes
        size_t cap = (size_t)n + n / 4 + 16;
        HIPCHK(hipMalloc((void**)&c->d_descs, cap * sizeof(tsx_chunk_desc)));
        HIPCHK(hipHostMalloc((void**)&c->h_descs, cap * sizeof(tsx_chunk_desc), hipHostMallocMapped | hipHostMallocPortable));
        HIPCHK(hipHostGetDevicePointer((void**)&c->hd_descs, c->h_descs, 0));
        HIPCHK(hipMalloc((void**)&c->d_gchunks, cap * sizeof(tsx_gcm_chunk)));
        HIPCHK(hipMalloc((void**)&c->d_status, cap * sizeof(int32_t)));
        HIPCHK(hipMalloc((void**)&c->d_zlen, cap * sizeof(uint32_t)));
        c->descs_cap = cap;
    }
    // partials: GCM needs 4 u32 per 64 KiB sub-block of the (possibly expanded) stage input, CRC 1 per 256 KiB of the
    // bytes it runs over - the source chunks on the way in, the dst_cap-sized output slots on the way back
    size_t bound = tsx_transformed_bound(max_len, flags & TSX_COMPRESS) + 64;
    size_t subs = (bound + TSX_GCM_SUB_BYTES - 1) / TSX_GCM_SUB_BYTES + 1;
    size_t crc_subs = ((size_t)(max_out > max_len ? max_out : max_len) + TSX_CRC_SUB_BYTES - 1) / TSX_CRC_SUB_BYTES + 1;
    size_t per_chunk = subs * 4 > crc_subs ? subs * 4 : crc_subs;
    int rc = grow(&c->d_partials, &c->partials_cap, (size_t)n * per_chunk);
    if (rc) return rc;
    c->partials_per_chunk = per_chunk;
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
        rc = grow(&zp, &c->zwork_cap, zw);
        c->d_zwork = zp;                                     // also when grow failed: it has freed the old block
        if (rc) return rc;
        if (max_out && dec_use_blocks(n, max_out)) {         // inverse chain, small batch
            uint8_t* bp = (uint8_t*)c->d_bwork;
            rc = grow(&bp, &c->bwork_cap, tsx_zstd_blockmode_bytes(n, max_out));
            c->d_bwork = bp;
            if (rc) { c->d_bwork = nullptr; c->bwork_cap = 0; (void)hipGetLastError(); }   // no room for the fast path: the chunk form decodes the batch
        }
    }
    return TSX_OK;
}

static int ctx_init_device_objects(tsx_ctx* c) {
    HIPCHK(hipSetDevice(c->dev->hip_id));
    HIPCHK(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->st_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->st_out, hipStreamNonBlocking));
    for (auto& e : c->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipEventCreate(&c->ev_key));
    for (auto& e : c->sub_ev[0]) HIPCHK(hipEventCreate(&e));      // the other rows are created by the first pipelined batch
    HIPCHK(hipMalloc((void**)&c->d_key, sizeof(tsx_gcm_key)));
    HIPCHK(hipMalloc((void**)&c->d_keyraw, 128));
    // the raw-key buffer is only written with TSX_GCM_SETUP_KERNEL: what tsx_debug_key_residue reads must never be an allocator's leftovers.
    // On the context's OWN stream: it is non-blocking, a memset on the null stream could land behind the first batch's key copy
    // (seen on the device: the first batch of a fresh pooled context encrypted with a zeroed schedule).
    HIPCHK(hipMemsetAsync(c->d_key, 0, sizeof(tsx_gcm_key), c->st));
    HIPCHK(hipMemsetAsync(c->d_keyraw, 0, 128, c->st));
    HIPCHK(hipHostMalloc((void**)&c->h_keyraw, 128, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->h_key, sizeof(tsx_gcm_key), hipHostMallocMapped | hipHostMallocPortable));
    HIPCHK(hipHostGetDevicePointer((void**)&c->hd_key, c->h_key, 0));
    HIPCHK(hipMalloc((void**)&c->d_segdone, 64));
    HIPCHK(hipMemsetAsync(c->d_segdone, 0, 64, c->st));
    HIPCHK(hipHostMalloc((void**)&c->h_segflag, 64, hipHostMallocMapped | hipHostMallocPortable));
    HIPCHK(hipHostGetDevicePointer((void**)&c->hd_segflag, c->h_segflag, 0));
    *c->h_segflag = 0;
    HIPCHK(hipStreamSynchronize(c->st));                                // (the counter is zero before a lane of the combiner can touch it)
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
    tsx_device_scope keep;
    c->dev_index = device_index; c->dev = dev;
    int rc = ctx_init_device_objects(c);
    if (rc == TSX_OK && max_chunks && max_chunk_size) rc = ctx_reserve(c, max_chunks, max_chunk_size, max_chunk_size, 0, false, 0, 0);
    if (rc) { ctx_free_device_mem(c); delete c; return rc; }     // a half-built context leaves nothing behind
    *out = c;
    return TSX_OK;
}

extern "C" void tsx_ctx_destroy(tsx_ctx* c) {
    if (!c) return;
    tsx_device_scope keep;
    ctx_free_device_mem(c);
    delete c;
}

extern "C" int tsx_ctx_timing(const tsx_ctx* c, tsx_timing* out) {
    if (!c || !out) return TSX_E_INVAL;
    *out = c->timing;
    return TSX_OK;
}

extern "C" int tsx_ctx_device(const tsx_ctx* c) { return c ? c->dev_index : TSX_E_INVAL; }

// ---- ctx-less calls: which device, which pooled context ---------------------------------------------
// One JVM per broker drives ALL GPUs of the node from >= 10 RLM threads plus the ChunkCache pool (README.md:218-222,
// RemoteStorageManager.java:212): a ctx-less call goes to the device its thread asked for (tsx_set_thread_device: the JVM side
// passes segment hash % devices, SURVEY 8e "segment s -> GPU s mod N"), otherwise to the device with the fewest batches in
// flight (ties broken round-robin).
extern "C" int tsx_set_thread_device(int device_index) {
    if (device_index >= 0) {
        std::lock_guard<std::mutex> lk(g_mu);
        if (device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    }
    t_dev_hint = device_index < 0 ? -1 : device_index;
    return TSX_OK;
}

extern "C" int tsx_pool_stats(int device_index, uint32_t* idle, uint32_t* in_use, uint64_t* batches) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    if (idle) *idle = (uint32_t)g_devs[device_index].idle.size();
    if (in_use) *in_use = g_devs[device_index].in_use;
    if (batches) *batches = g_devs[device_index].batches;
    return TSX_OK;
}

static size_t ctx_workspace_bytes(const tsx_ctx* c) {
    return c->descs_cap * (sizeof(tsx_chunk_desc) + sizeof(tsx_gcm_chunk) + 8) + c->partials_cap * 4 + c->in_cap + c->out_cap + c->mid_cap + c->zwork_cap + c->bwork_cap;
}

static tsx_ctx* pool_acquire(int* rc) {
    int di = -1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        const int nd = (int)g_devs.size();
        if (nd == 0) { snprintf(g_last_err, sizeof g_last_err, "tsx_init has not succeeded"); *rc = TSX_E_DEVICE; return nullptr; }
        if (t_dev_hint >= 0 && t_dev_hint < nd) di = t_dev_hint;
        else {
            const int first = (int)(g_rr++ % (uint32_t)nd);
            di = first;
            for (int k = 1; k < nd; k++) { const int j = (first + k) % nd; if (g_devs[j].in_use < g_devs[di].in_use) di = j; }
        }
        tsx_device& d = g_devs[di];
        d.in_use++; d.batches++;
        if (!d.idle.empty()) { tsx_ctx* c = d.idle.back(); d.idle.pop_back(); d.idle_bytes -= ctx_workspace_bytes(c); return c; }
    }
    tsx_ctx* c = nullptr;
    *rc = tsx_ctx_create(di, 0, 0, &c);
    if (*rc) { std::lock_guard<std::mutex> lk(g_mu); g_devs[di].in_use--; return nullptr; }
    c->pooled = true;
    return c;
}
// Idle pooled contexts of device di give their memory back (an allocation has just failed: what is cached must not be the reason).
static bool pool_drain(tsx_device* dev) {
    std::vector<tsx_ctx*> dead;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        dead.swap(dev->idle); dev->idle_bytes = 0;
    }
    if (dead.empty()) return false;
    tsx_device_scope keep;
    for (tsx_ctx* c : dead) { ctx_free_device_mem(c); delete c; }
    return true;
}
static int reserve_or_drain(tsx_ctx* c, uint32_t n, uint32_t max_len, uint32_t max_out, uint32_t flags, bool host_mem, size_t in_bytes, size_t out_bytes) {
    int rc = ctx_reserve(c, n, max_len, max_out, flags, host_mem, in_bytes, out_bytes);
    if (rc != TSX_E_NOMEM) return rc;
    (void)hipGetLastError();
    if (!pool_drain(c->dev)) return rc;                                 // nothing was cached: the device really is full
    return ctx_reserve(c, n, max_len, max_out, flags, host_mem, in_bytes, out_bytes);
}
static void pool_release(tsx_ctx* c) {
    {
        std::unique_lock<std::mutex> lk(g_mu);
        tsx_device& d = *c->dev;
        d.in_use--;
        if (c->d_bwork) {
            // the fetch side's ForkJoinPool issues from dozens of threads (ChunkCache.java:140): without a bound every idle context would
            // park a block-form workspace; beyond a few the next small fetch on such a context re-allocates it (~1 ms) instead
            uint32_t with = 0;
            for (const tsx_ctx* o : d.idle) if (o->d_bwork) with++;
            if (with >= TSX_POOL_MAX_IDLE_BWORK) {
                lk.unlock();
                { tsx_device_scope keep; hipSetDevice(d.hip_id); hipFree(c->d_bwork); }
                c->d_bwork = nullptr; c->bwork_cap = 0;
                lk.lock();
            }
        }
        const size_t b = ctx_workspace_bytes(c);
        if (d.idle.size() < TSX_POOL_MAX_IDLE && (d.idle.empty() || d.idle_bytes + b <= d.idle_cap)) { d.idle.push_back(c); d.idle_bytes += b; return; }
    }
    tsx_device_scope keep;
    ctx_free_device_mem(c);                                            // a burst of callers does not pin its workspaces forever
    delete c;
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

// Zeroes what a failed chunk of the inverse chain left in its output slot (device-memory calls: a forged chunk's
// unauthenticated plaintext must not stay readable - JCE's doFinal releases nothing on a bad tag,
// DecryptionChunkEnumeration.java:54-62).  Up to min(dst_cap, src_len) bytes can have been written.
__global__ __launch_bounds__(256) void scrub_failed_kernel(const tsx_chunk_desc* __restrict__ descs, const int32_t* __restrict__ status,
                                                           uint8_t* __restrict__ dst, uint32_t blocks_per_chunk) {
    const uint32_t i = blockIdx.x / blocks_per_chunk, part = blockIdx.x % blocks_per_chunk;
    if (status[i] == TSX_OK) return;
    const tsx_chunk_desc d = descs[i];
    const uint32_t len = d.src_len < d.dst_cap ? d.src_len : d.dst_cap;
    uint8_t* o = dst + d.dst_off;
    const uint32_t q = len >> 4;                                         // slots are 16-byte aligned
    uint4 z; z.x = z.y = z.z = z.w = 0;
    for (uint32_t p = part * 256 + threadIdx.x; p < q; p += blocks_per_chunk * 256) reinterpret_cast<uint4*>(o)[p] = z;
    if (part == 0) for (uint32_t b = (q << 4) + threadIdx.x; b < len; b += 256) o[b] = 0;
}

// ---- batch drivers ---------------------------------------------------------------------------------
static int validate(const tsx_chunk_desc* descs, uint32_t n, size_t src_size, size_t dst_size, bool need_dst, uint32_t* max_len, uint32_t* max_out,
                    size_t* in_bytes, bool* monotonic) {
    *max_len = 0; *max_out = 0; *in_bytes = 0; *monotonic = true;
    uint64_t prev_src_end = 0, prev_dst_end = 0;
    for (uint32_t i = 0; i < n; i++) {
        const tsx_chunk_desc& d = descs[i];
        if ((d.src_off & 15) || (need_dst && (d.dst_off & 15))) return TSX_E_INVAL;       // 16-byte aligned slots
        if (need_dst && (d.dst_off > dst_size || d.dst_cap > dst_size - d.dst_off)) return TSX_E_INVAL;   // no wrap-around
        if (d.src_len >= (1u << 30) + 4096) return TSX_E_INVAL;                             // chunk.size <= 2^30 - 1 (RemoteStorageManagerConfig.java:122-130)
        if (d.src_off > src_size || d.src_len > src_size - d.src_off) return TSX_E_INVAL;    // inside the caller's source buffer, no wrap-around
        if (d.src_len > *max_len) *max_len = d.src_len;
        if (need_dst && d.dst_cap > *max_out) *max_out = d.dst_cap;
        if (d.src_off + d.src_len > *in_bytes) *in_bytes = d.src_off + d.src_len;
        if (d.src_off < prev_src_end || (need_dst && d.dst_off < prev_dst_end)) *monotonic = false;
        prev_src_end = d.src_off + d.src_len;
        if (need_dst) prev_dst_end = d.dst_off + d.dst_cap;
    }
    return TSX_OK;
}

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }

struct tsx_sub { uint32_t lo, n; size_t in_lo, in_hi; };     // chunks [lo, lo + n), their input bytes [in_lo, in_hi) of src

struct tsx_run {                                              // what one batch needs everywhere below
    tsx_ctx* c; const tsx_batch_params* params; tsx_chunk_desc* descs; uint32_t n; const void* src; void* dst; size_t src_size, dst_size;
    int mem_kind, mode; uint32_t flags, max_len, max_out; bool host, packed, enc, comp, fuse_stages, combined;
    const uint8_t* d_src; uint8_t* d_dst;
};

static uint32_t zstd_sched_from_env();
// Enqueues the kernels of chunks [lo, lo + n) on compute stream st; e[0..3] are recorded at the stage boundaries.
static int launch_stages(const tsx_run& r, const tsx_sub& sb, hipEvent_t* e, hipStream_t st) {
    tsx_ctx* c = r.c;
    const uint32_t n = sb.n, lo = sb.lo, flags = r.flags;
    tsx_chunk_desc* dd = c->d_descs + lo;                              // (lean batches: the pinned host mirror instead, below)
    int32_t* ds = c->d_status + lo;
    uint32_t* dz = c->d_zlen + lo;
    tsx_gcm_chunk* dg = c->d_gchunks + lo;
    uint8_t* dmid = c->d_mid ? c->d_mid + (size_t)lo * c->mid_stride : nullptr;
    // the Zstd workspace of chunk i is slot i of the batch, whichever piece it travels in: pieces of one batch co-reside
    void* dzw = c->d_zwork ? (uint8_t*)c->d_zwork + (size_t)lo * tsx_zstd_workspace_bytes(1, 0) : nullptr;
    uint32_t* const dpart = c->d_partials + (size_t)lo * c->partials_per_chunk;     // this piece's slice of the CRC / GHASH partial sums
    tsx_timing& t = c->timing;
    memcpy(c->h_descs + lo, r.descs + lo, (size_t)n * sizeof(tsx_chunk_desc));
    // lean: the compressor waves own and publish their chunks' statuses and work on the descriptors and the key schedule where they
    // are - in the context's pinned host memory.  Nothing but the launch goes into the stream: a small copy queued while the copy
    // engines move other callers' gigabytes waits for them (measured: +550 ms per 256-chunk call with 10 callers), a small kernel
    // waits for a slot on a chip full of compressor waves.
    const bool lean = r.mode == 0 && r.comp && r.enc && r.fuse_stages;
    if (lean) {
        dd = c->hd_descs + lo;
        // the waves are the only writers of status / dst_len here and no init or publish kernel runs: what the caller handed in (often a
        // stale TSX_OK) must not survive a launch that never ran
        for (uint32_t i = lo; i < lo + n; i++) { c->h_descs[i].status = TSX_E_DEVICE; c->h_descs[i].dst_len = 0; }
    } else {
        HIPCHK(hipMemcpyAsync(dd, c->h_descs + lo, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(init_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ds, n);
    }
    HIPCHK(hipEventRecord(e[0], st));
    if (r.mode == 2) {
        tsx_launch_crc32c(st, c->dev->d_crc, r.d_src, dd, n, r.max_len, dpart, 0);
        HIPCHK(hipEventRecord(e[1], st)); HIPCHK(hipEventRecord(e[2], st));
        t.crc_launches += 2;
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n);      // a reused descriptor must not keep an old status
    } else if (r.mode == 0) {
        // With compression the whole chain of a chunk runs in the wave that compresses it: CRC32C of the source chunk first, GCM over
        // the finished frame last - one launch per batch.  The batch CRC / GCM kernels want 20 / 40 KiB of LDS per workgroup and, on
        // a chip filled by the compressor waves of the batches in flight, sat hundreds of ms in the queue for a few ms of work.
        // TSX_STAGES_SEPARATE=1 keeps one launch per stage (A/B measurements, tests of the stand-alone kernels).
        if ((flags & TSX_CRC) && !r.fuse_stages) { tsx_launch_crc32c(st, c->dev->d_crc, r.d_src, dd, n, r.max_len, dpart, 0); t.crc_launches += 2; }
        HIPCHK(hipEventRecord(e[1], st));
        bool fused = false;
        if (r.comp) {
            fused = r.enc && r.fuse_stages;
            tsx_chain_fuse fuse{nullptr, nullptr, nullptr, nullptr, 0, 0};
            if (r.fuse_stages && (flags & TSX_CRC)) fuse.crc = c->dev->d_crc;
            if (fused) { fuse.aes = c->dev->d_aes; fuse.key = c->hd_key; fuse.out = r.d_dst; fuse.self_status = 1; fuse.key_on_host = 1; }
            const uint32_t sched = zstd_sched_from_env();
            (void)hipGetLastError();
            t.zstd_launches += tsx_launch_zstd_compress(st, c->dev->d_zc, r.d_src, dd, n, r.max_len, dmid, c->mid_stride, dz, ds, dzw,
                                                       r.params->zstd_profile, sched, fuse);
            // hipLaunchKernelGGL reports through the last-error slot only; a lean batch has nothing behind the launch that would notice
            if (hipGetLastError() != hipSuccess) return TSX_E_DEVICE;
        }
        HIPCHK(hipEventRecord(e[2], st));
        if (fused) {
        } else if (r.enc) {
            hipLaunchKernelGGL(plan_gcm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, n, (const uint32_t*)dz,
                               (uint64_t)c->mid_stride, r.comp ? 1 : 0, 0, 0, dg, ds);
            uint32_t glen = r.comp ? (uint32_t)tsx_transformed_bound(r.max_len, TSX_COMPRESS) : r.max_len;
            tsx_launch_gcm(st, c->dev->d_aes, c->d_key, dg, n, glen, r.comp ? dmid : r.d_src, r.d_dst, dpart, ds, 0);
            t.gcm_launches += 2;
        } else {
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, dd, (const uint32_t*)dz,
                               (uint64_t)c->mid_stride, r.comp ? 1 : 0, r.comp ? (const uint8_t*)dmid : r.d_src, r.d_dst, ds, bpc);
        }
        if (!lean) hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n);
    } else {
        HIPCHK(hipEventRecord(e[1], st));
        const uint8_t* zsrc = r.d_src;    // where the Zstd frames live when there is no encryption
        if (r.enc) {
            hipLaunchKernelGGL(plan_gcm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, n, (const uint32_t*)dz,
                               (uint64_t)c->mid_stride, 0, 1, r.comp ? 1 : 0, dg, ds);
            tsx_launch_gcm(st, c->dev->d_aes, c->d_key, dg, n, r.max_len, r.d_src, r.comp ? dmid : r.d_dst, dpart, ds, 1);
            t.gcm_launches += 2;
            zsrc = dmid;
        }
        HIPCHK(hipEventRecord(e[2], st));
        if (r.comp) {
            const uint32_t* skip = nullptr; uint32_t skip_stride = 0;
            if (c->d_bwork && c->bwork_cap >= tsx_zstd_blockmode_bytes(r.n, r.max_out) && dec_use_blocks(r.n, r.max_out)) {
                c->last_used_blocks = true;
                // one workgroup per block; what that form does not take (or gives up on) is decoded by the chunk-serial kernel behind it.
                // A piece of a batch works in its own part of the workspace (the layout is per chunk: headers, then arenas, of THIS launch).
                void* const bw = (uint8_t*)c->d_bwork + tsx_zstd_blockmode_bytes(lo, r.max_out);
                t.unzstd_launches += tsx_launch_zstd_decompress_blocks(st, zsrc, r.enc ? 1 : 0, (uint64_t)c->mid_stride, dd, n, r.max_out, r.d_dst, ds, bw);
                skip = tsx_zstd_blockmode_skip(bw, &skip_stride);
                c->blk_pieces.push_back({lo, n}); c->last_max_out = r.max_out;
            }
            t.unzstd_launches += tsx_launch_zstd_decompress(st, c->dev->d_zc, zsrc, r.enc ? 1 : 0, (uint64_t)c->mid_stride, dd, n, r.d_dst, ds, dzw, skip, skip_stride);
        } else if (!r.enc) {
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, dd, (const uint32_t*)dz, (uint64_t)0, 0, r.d_src, r.d_dst, ds, bpc);
        }
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n);
        if (r.enc && !r.comp) {            // decrypted straight into the caller's slots: nothing of a chunk that failed its tag check stays
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(scrub_failed_kernel, dim3(n * bpc), dim3(256), 0, st, (const tsx_chunk_desc*)dd, (const int32_t*)ds, r.d_dst, bpc);
        }
    }
    HIPCHK(hipEventRecord(e[3], st));
    if (r.mode == 1 && (flags & TSX_CRC)) {
        // CRC of the restored bytes; upper bound of a restored chunk is its slot capacity
        tsx_launch_crc32c(st, c->dev->d_crc, r.d_dst, dd, n, r.max_out, dpart, 1);
        t.crc_launches += 2;
    }
    if (!lean) HIPCHK(hipMemcpyAsync(c->h_descs + lo, dd, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(e[4], st));                                   // the caller's descriptors are filled in when this event has passed
    return TSX_OK;
}

// The bytes chunks [lo, lo + n) produced travel back on st_out (host-memory batches; their descriptors are on the host already).
// Exactly dst_len bytes per chunk: a slot's slack may hold bytes of an earlier batch on this (possibly pooled) context.
static int copy_back(const tsx_run& r, const tsx_sub& sb, size_t* packed_at, bool* packed_full, hipStream_t out_st) {
    tsx_ctx* c = r.c;
    size_t run_at = 0, run_len = 0;
    for (uint32_t i = sb.lo; i < sb.lo + sb.n; i++) {
        tsx_chunk_desc& d = r.descs[i];
        if (r.packed) {
            const size_t slot_off = d.dst_off;
            d.dst_off = *packed_at;
            if (d.status != TSX_OK) { d.dst_len = 0; continue; }
            if (*packed_full || *packed_at + d.dst_len > r.dst_size) { *packed_full = true; d.status = TSX_E_DST_TOO_SMALL; d.dst_len = 0; continue; }
            if (d.dst_len) {
                // packed output lands at any byte of the caller's buffer.  The copy engines move a D2H copy of ODD size to an ODD host
                // address at 12 GB/s instead of 31 (110 us instead of 42 per 1.3 MB chunk, profiles/r03_copy_engine_probe.txt; every
                // other combination of size and address is fast): such a copy goes as its 64-byte multiple + the last < 64 bytes.
                uint8_t* const hp = (uint8_t*)r.dst + *packed_at;
                const size_t tail = (((uintptr_t)hp & 1) && (d.dst_len & 1) && d.dst_len > 4096) ? (d.dst_len & 63) : 0;
                HIPCHK(hipMemcpyAsync(hp, c->d_out + slot_off, d.dst_len - tail, hipMemcpyDeviceToHost, out_st));
                if (tail) HIPCHK(hipMemcpyAsync(hp + (d.dst_len - tail), c->d_out + slot_off + (d.dst_len - tail), tail, hipMemcpyDeviceToHost, out_st));
            }
            *packed_at += d.dst_len;
        } else {
            // neighbours that produced back-to-back bytes (restored chunks fill their slots: the usual fetch) travel as ONE copy
            if (d.status != TSX_OK || d.dst_len == 0) continue;
            if (run_len && d.dst_off == run_at + run_len) { run_len += d.dst_len; continue; }
            if (run_len) HIPCHK(hipMemcpyAsync((uint8_t*)r.dst + run_at, c->d_out + run_at, run_len, hipMemcpyDeviceToHost, out_st));
            run_at = d.dst_off; run_len = d.dst_len;
        }
    }
    if (run_len) HIPCHK(hipMemcpyAsync((uint8_t*)r.dst + run_at, c->d_out + run_at, run_len, hipMemcpyDeviceToHost, out_st));
    return TSX_OK;
}

static uint32_t zstd_sched_from_env() {
    uint32_t sched = 0;                                              // the kernel's default speculation schedule
    if (const char* e = getenv("TSX_ZSTD_SCHED")) {                  // "k0,k1": explicit schedule (measurements; same bytes)
        unsigned a = 0, b = 0;
        if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && a <= 59 && b >= 1 && b <= 59) sched = a | b << 8;
    }
    return sched;
}

static int combiner_get(tsx_device* dev, tsx_combiner** out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!dev->comb) {
        std::unique_ptr<tsx_combiner> cb(new (std::nothrow) tsx_combiner);
        if (!cb) return TSX_E_NOMEM;
        // lanes: 3 with the runtime's default of 4 hardware queues (lanes + the two copy streams must not pile up on them); a process that
        // runs with GPU_MAX_HW_QUEUES = q >= 8 gets q / 2 lanes, at most 8 - a caller waits for a free lane 1 / lanes of a kernel's duration
        // on average, and what waits is not in flight.  TSX_LANES overrides (up to 24).  Round 4 swept queues x lanes with 20 / 32 callers
        // (profiles/r04_broker_shape_experiments.txt): 16 x 8 is as good as anything - 12-20 lanes on 16-24 queues measure the same within
        // the run-to-run spread, 32 queues lose 10-20 %.
        uint32_t nl = 3;
        if (const char* q = getenv("GPU_MAX_HW_QUEUES")) { const long v = atol(q); if (v >= 8) nl = (uint32_t)(v / 2 > 8 ? 8 : v / 2); }
        if (const char* e = getenv("TSX_LANES")) { const long v = atol(e); if (v >= 1 && v <= TSX_LANES_MAX) nl = (uint32_t)v; }
        // the copy streams first: whatever the runtime's stream -> hardware-queue assignment, the short copies and their event markers
        // are not the ones that end up behind a second-long kernel of a lane created later
        if (const char* e = getenv("TSX_COMBINER_MAX_CHUNKS")) { const long v = atol(e); if (v > 0) cb->cap = (uint32_t)v; }
        cb->n_in = 1; cb->n_out = 1;                                    // more streams measured WORSE (see tsx_combiner)
        if (const char* e = getenv("TSX_COPY_STREAMS")) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && a <= TSX_COPY_STREAMS_MAX && b >= 1 && b <= TSX_COPY_STREAMS_MAX) { cb->n_in = a; cb->n_out = b; } }
        bool ok = true;
        for (uint32_t i = 0; ok && i < cb->n_in; i++) ok = hipStreamCreateWithFlags(&cb->copy_in_s[i], hipStreamNonBlocking) == hipSuccess;
        for (uint32_t i = 0; ok && i < cb->n_out; i++) ok = hipStreamCreateWithFlags(&cb->copy_out_s[i], hipStreamNonBlocking) == hipSuccess;
        for (uint32_t i = 0; ok && i < nl; i++) {
            tsx_lane& l = cb->lane[i];
            ok = tsx_compressor_stream(&l.st, dev->hip_id) == hipSuccess && hipEventCreateWithFlags(&l.end, hipEventDisableTiming) == hipSuccess &&
                 hipHostMalloc((void**)&l.h_segs, TSX_GROUP_MAX_SEGS * sizeof(tsx_zseg), hipHostMallocMapped | hipHostMallocPortable) == hipSuccess &&
                 hipHostGetDevicePointer((void**)&l.d_segs, l.h_segs, 0) == hipSuccess;
            cb->nlanes = i + 1;
        }
        dev->comb = std::move(cb);
        if (!ok) { (void)hipGetLastError(); tsx_device& d = *dev; std::unique_ptr<tsx_combiner> dead = std::move(d.comb);
                   for (uint32_t i = 0; i < dead->nlanes; i++) { tsx_lane& l = dead->lane[i]; if (l.st) hipStreamDestroy(l.st); if (l.end) hipEventDestroy(l.end); if (l.h_segs) hipHostFree(l.h_segs); }
                   for (auto& q : dead->copy_in_s) if (q) hipStreamDestroy(q);
                   for (auto& q : dead->copy_out_s) if (q) hipStreamDestroy(q);
                   return TSX_E_DEVICE; }
    }
    *out = dev->comb.get();
    return TSX_OK;
}

// The leader's part, outside the lock: everything the members of one group need on lane `l`, in stream order.
static int combiner_launch(tsx_combiner* cb, tsx_lane& l, const std::vector<tsx_zreq*>& grp) {
    hipStream_t ls = l.st;
    uint32_t first = 0;
    for (size_t k = 0; k < grp.size(); k++) {
        tsx_zreq* q = grp[k]; tsx_ctx* c = q->c; const tsx_run& r = *q->r;
        const uint32_t n = r.n;
        if (q->in_ready) HIPCHK(hipStreamWaitEvent(ls, q->in_ready, 0));
        HIPCHK(hipEventRecord(c->ev[0], ls));
        memcpy(c->h_descs, r.descs, (size_t)n * sizeof(tsx_chunk_desc));
        // with encryption the launch is the group's ONLY operation on the lane: descriptors and key schedules stay in the members' pinned
        // memory (the waves read and write them in place), statuses are owned by the waves
        if (r.enc) {
            tsx_gcm_key_build_host(r.params->key, r.params->aad, r.params->aad_len, c->h_key);
            for (uint32_t i = 0; i < n; i++) { c->h_descs[i].status = TSX_E_DEVICE; c->h_descs[i].dst_len = 0; }      // the waves own them from here
        } else {
            HIPCHK(hipMemcpyAsync(c->d_descs, c->h_descs, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyHostToDevice, ls));
            hipLaunchKernelGGL(init_status_kernel, dim3((n + 255) / 256), dim3(256), 0, ls, c->d_status, n);
        }
        tsx_zseg& sg = l.h_segs[k];
        memset(&sg, 0, sizeof sg);
        sg.first = first; sg.n = n; sg.profile = r.params->zstd_profile;
        sg.src_base = r.d_src; sg.descs = r.enc ? c->hd_descs : c->d_descs; sg.mid = c->d_mid; sg.mid_stride = c->mid_stride; sg.zlen = c->d_zlen; sg.status = c->d_status;
        sg.work = (uint8_t*)c->d_zwork;
        sg.fuse.crc = (r.flags & TSX_CRC) ? c->dev->d_crc : nullptr;
        if (r.enc) {
            sg.fuse.aes = c->dev->d_aes; sg.fuse.key = c->hd_key; sg.fuse.out = r.d_dst; sg.fuse.self_status = 1; sg.fuse.key_on_host = 1;
            __atomic_store_n(c->h_segflag, 0u, __ATOMIC_RELEASE);
            sg.done = c->d_segdone; sg.flag = c->hd_segflag;             // this member's caller returns when ITS chunks are done (run_combined)
        }
        first += n;
    }
    (void)hipGetLastError();                                             // (hipErrorNotReady of the leader's lane queries)
    tsx_launch_zstd_compress_segments(ls, l.d_segs, l.h_segs, (uint32_t)grp.size(), first, zstd_sched_from_env());
    if (hipGetLastError() != hipSuccess) return TSX_E_DEVICE;           // every member gets the rc (combiner_submit)
    for (tsx_zreq* q : grp) {
        tsx_ctx* c = q->c; const tsx_run& r = *q->r;
        const uint32_t n = r.n;
        if (!r.enc) {                                                    // compression only: the frames go from the staging buffer to the caller's slots
            const uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, ls, c->d_descs, (const uint32_t*)c->d_zlen, (uint64_t)c->mid_stride, 1,
                               (const uint8_t*)c->d_mid, r.d_dst, c->d_status, bpc);
            hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, ls, c->d_descs, (const int32_t*)c->d_status, n);
            HIPCHK(hipMemcpyAsync(c->h_descs, c->d_descs, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyDeviceToHost, ls));
        }
        HIPCHK(hipEventRecord(c->ev[1], ls));
    }
    HIPCHK(hipEventRecord(l.end, ls));
    (void)cb;
    return TSX_OK;
}

// Hand one batch to the device's combiner and return when its group has been queued (q.done); the caller then waits for its own event.
static void combiner_submit(tsx_combiner* cb, tsx_zreq& q) {
    std::unique_lock<std::mutex> lk(cb->mu);
    cb->pending.push_back(&q);
    cb->cv.notify_all();
    while (!q.done) {
        if (cb->leader) { cb->cv.wait(lk); continue; }
        cb->leader = true;
        // ---- lead: wait for a free lane (requests keep arriving meanwhile and join the group), take a group, queue it ----
        int li = -1;
        uint32_t nfree = 0;
        for (;;) {
            nfree = 0;
            for (uint32_t i = 0; i < cb->nlanes; i++) {
                tsx_lane& l = cb->lane[i];
                if (l.busy && hipEventQuery(l.end) != hipErrorNotReady) l.busy = false;     // done - or failed: the launch on it will say so
                if (!l.busy) { if (li < 0) li = (int)i; nfree++; }
            }
            if (li >= 0) break;
            (void)hipGetLastError();                                     // hipErrorNotReady of the queries
            cb->cv.wait_for(lk, std::chrono::microseconds(200));
        }
        // What waits is shared out over the lanes that are free NOW (the next leader takes the next lane), in arrival order - which is the
        // order the members' input copies land in.  Everything onto the first free lane made one launch wait for the LAST member's copy,
        // and its members finish, come back and pile up together for good: 32 callers in step moved 12 GiB/s where 20 moved 17
        // (profiles/r03_bench_default_run_head.json before this).
        // (admission cap: wait for members to complete - combiner_member_done notifies - while the first waiting batch would exceed it;
        //  a batch larger than the cap goes alone on an empty device)
        while (cb->cap && cb->inflight && cb->inflight + cb->pending.front()->r->n > cb->cap) cb->cv.wait_for(lk, std::chrono::microseconds(500));
        const size_t share = (cb->pending.size() + nfree - 1) / nfree;
        std::vector<tsx_zreq*> grp;
        uint32_t chunks = 0;
        while (!cb->pending.empty() && grp.size() < share && grp.size() < TSX_GROUP_MAX_SEGS && (grp.empty() || chunks + cb->pending.front()->r->n <= TSX_GROUP_MAX_CHUNKS) &&
               (grp.empty() || !cb->cap || cb->inflight + chunks + cb->pending.front()->r->n <= cb->cap)) {
            grp.push_back(cb->pending.front()); chunks += cb->pending.front()->r->n;
            cb->pending.erase(cb->pending.begin());
        }
        tsx_lane& l = cb->lane[li];
        l.busy = true;
        cb->groups++; cb->members += grp.size();
        cb->inflight += chunks; if (cb->inflight > cb->inflight_peak) cb->inflight_peak = cb->inflight;
        lk.unlock();
        const int rc = combiner_launch(cb, l, grp);
        if (rc != TSX_OK) { (void)hipGetLastError(); (void)hipStreamSynchronize(l.st); }
        lk.lock();
        if (rc != TSX_OK) { l.busy = false; cb->inflight -= chunks; }
        for (tsx_zreq* m : grp) { m->rc = rc; m->done = true; }
        cb->leader = false;
        cb->cv.notify_all();
    }
}

// A member of a launch is done (or gave up waiting): its chunks leave the admission count.
static void combiner_member_done(tsx_combiner* cb, uint32_t n) {
    std::lock_guard<std::mutex> lk(cb->mu);
    cb->inflight = cb->inflight >= n ? cb->inflight - n : 0;
    if (cb->cap) cb->cv.notify_all();
}
// test hook: the most compressor chunks ever launched and not yet done on a device (admission cap)
extern "C" int tsx_debug_combiner_inflight_peak(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size() || !g_devs[device_index].comb) return TSX_E_INVAL;
    std::lock_guard<std::mutex> lk2(g_devs[device_index].comb->mu);
    return (int)g_devs[device_index].comb->inflight_peak;
}
// Where a context-less compressing call spends its time (test / measurement hook, tools/broker_probe.py): nanoseconds summed over calls -
// waiting for the own input copy, from asking for a launch to the own chunks being done (lane wait + kernel), output copies - and calls.
static std::atomic<uint64_t> g_phase_ns[4];
extern "C" void tsx_debug_combined_phases(uint64_t out[4], int reset) {
    for (int i = 0; i < 4; i++) { out[i] = g_phase_ns[i].load(); if (reset) g_phase_ns[i].store(0); }
}
// A ctx-less compressing batch: its buffers live in the pooled context, its work runs on the device's shared streams.
static int run_combined(tsx_run& r) {
    tsx_ctx* c = r.c;
    const uint32_t n = r.n;
    uint32_t max_len, max_out; size_t in_bytes; bool monotonic;
    int rc = validate(r.descs, n, r.src_size, r.dst_size, !r.packed, &max_len, &max_out, &in_bytes, &monotonic);
    if (rc) return rc;
    size_t out_bytes = r.dst_size;
    if (r.packed) {
        const size_t slot = (tsx_transformed_bound(max_len, r.flags) + 63) & ~(size_t)63;
        if (slot >= ((size_t)1 << 32)) return TSX_E_INVAL;
        for (uint32_t i = 0; i < n; i++) { r.descs[i].dst_off = (uint64_t)i * slot; r.descs[i].dst_cap = (uint32_t)slot; }
        out_bytes = (size_t)n * slot; max_out = (uint32_t)slot;
    }
    r.max_len = max_len; r.max_out = max_out;
    // Zero-copy output (round 4).  The waves that encrypt a chunk write IV || C || TAG straight into the caller's buffer over PCIe when the
    // device can address it (memory pinned with tsx_host_register - the JVM's reused direct buffers - or hipHostMalloc): posted writes of
    // a few ms of a second-long wave, released to system scope before the chunk is counted done.  No device output buffer, no copy-out
    // phase: with 32-48 callers a segment's 256 output copies stood 0.3-1.0 s in the copy engine's queue behind the other callers'
    // (profiles/r04_broker_shape_experiments.txt) - time in which that caller offered the chip nothing.  Slot layout (TSX_MEM_HOST, what
    // GpuTransformChunkEnumeration.java:201 issues): nothing is left to do on the host.  Packed layout: the waves fill bound-sized slots
    // in the caller's buffer when it has room for them and the host packs them down in place; otherwise the copy path below.
    uint8_t* zc_dst = nullptr;
    if (r.host && r.enc && !getenv("TSX_NO_ZERO_COPY_OUT") && (!r.packed || r.dst_size >= out_bytes)) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, r.dst, 0) == hipSuccess && dp) zc_dst = (uint8_t*)dp;
        else (void)hipGetLastError();                                    // pageable memory: not addressable from the device
    }
    rc = reserve_or_drain(c, n, max_len, 0, r.flags, r.host, in_bytes, zc_dst ? 0 : out_bytes);
    if (rc) return rc;
    tsx_combiner* cb = nullptr;
    if ((rc = combiner_get(c->dev, &cb))) return rc;
    r.d_src = r.host ? c->d_in : (const uint8_t*)r.src;
    r.d_dst = zc_dst ? zc_dst : r.host ? c->d_out : (uint8_t*)r.dst;
    memset(&c->timing, 0, sizeof c->timing);
    tsx_zreq q{c, &r, nullptr, TSX_OK, false};
    const auto t_in = std::chrono::steady_clock::now();
    if (r.host) {
        hipStream_t cin = cb->copy_in_s[cb->rr_in.fetch_add(1) % cb->n_in];
        HIPCHK(hipEventRecord(c->ev[2], cin));
        if (in_bytes) HIPCHK(hipMemcpyAsync(c->d_in, r.src, in_bytes, hipMemcpyHostToDevice, cin));
        HIPCHK(hipEventRecord(c->sub_ev[0][5], cin));
        // The caller waits for ITS input before it asks for a launch.  Round 3 queued the launch at once, behind a stream wait on the copy:
        // every caller's gigabyte travels on the one copy-in stream, so with 20-32 callers a launch sat on its lane for the tens to
        // hundreds of ms its copy stood in that queue - a lane (a hardware queue) held by a kernel that cannot start, while callers whose
        // input HAD landed waited for a lane.  Now what reaches the combiner is runnable, a lane is only ever occupied by running waves,
        // and the callers whose copies land while the lanes are busy leave together as one launch (VERDICT r3 #2a: a launch carried
        // ~1 segment, 8 lanes x 256 chunks left 60 % of the chip's wave slots empty at 32 callers).
        HIPCHK(hipEventSynchronize(c->sub_ev[0][5]));
    }
    const auto t_sub = std::chrono::steady_clock::now();
    combiner_submit(cb, q);
    if (q.rc != TSX_OK) return q.rc;                                    // (the leader has taken the group's chunks out of the admission count)
    struct member_guard { tsx_combiner* cb; uint32_t n; ~member_guard() { if (cb) combiner_member_done(cb, n); } void release() { combiner_member_done(cb, n); cb = nullptr; } } guard{cb, n};
    tsx_timing& t = c->timing;
    if (r.enc) {
        // The launch may carry other callers' segments and goes on until the last of THEIR chunks is done; this caller waits for its own:
        // the wave that finishes this member's last chunk raises the flag (zstd_compress_segments_kernel).  A short sleep between looks -
        // a chunk takes about a second; the launch's own event is the safety net (a launch that ended, or failed, without raising it).
        for (uint32_t look = 0;; look++) {
            if (__atomic_load_n(c->h_segflag, __ATOMIC_ACQUIRE)) break;
            if ((look & 31) == 31) {
                const hipError_t e = hipEventQuery(c->ev[1]);
                if (e == hipSuccess) {
                    if (__atomic_load_n(c->h_segflag, __ATOMIC_ACQUIRE)) break;
                    (void)hipMemset(c->d_segdone, 0, 64);                // the count is not to be trusted any more
                    snprintf(g_last_err, sizeof g_last_err, "combined launch ended without completing a member");
                    return TSX_E_DEVICE;
                }
                if (e != hipErrorNotReady) { (void)hipGetLastError(); tsx_set_err("combined launch", e); return TSX_E_DEVICE; }
                (void)hipGetLastError();
            }
            std::this_thread::sleep_for(std::chrono::microseconds(look < 64 ? 20 : 100));
        }
        t.zstd_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_sub).count();
    } else {
        HIPCHK(hipEventSynchronize(c->ev[1]));                          // this batch's descriptors are on the host (its group may still be running for others)
        t.zstd_ms = ev_ms(c->ev[0], c->ev[1]);
    }
    guard.release();                                                    // this member's chunks are done: the next launch may come
    memcpy(r.descs, c->h_descs, (size_t)n * sizeof(tsx_chunk_desc));
    t.zstd_launches = 1; t.total_ms = t.zstd_ms;
    if (zc_dst) {
        // the bytes are in the caller's buffer already; a packed batch is packed down in place (chunk i's slot starts at or behind its place)
        if (r.packed) {
            size_t at = 0;
            for (uint32_t i = 0; i < n; i++) {
                tsx_chunk_desc& d = r.descs[i];
                const size_t slot_off = d.dst_off;
                d.dst_off = at;
                if (d.status != TSX_OK) { d.dst_len = 0; continue; }
                if (d.dst_len && slot_off != at) memmove((uint8_t*)r.dst + at, (const uint8_t*)r.dst + slot_off, d.dst_len);
                at += d.dst_len;
            }
        }
        t.h2d_ms = ev_ms(c->ev[2], c->sub_ev[0][5]);
        t.d2h_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_sub).count() - t.zstd_ms;
        t.total_ms = t.h2d_ms + t.zstd_ms + t.d2h_ms;
    } else if (r.host) {
        size_t packed_at = 0; bool packed_full = false;
        const tsx_sub sb{0, n, 0, in_bytes};
        hipStream_t cout_ = cb->copy_out_s[cb->rr_out.fetch_add(1) % cb->n_out];
        if ((rc = copy_back(r, sb, &packed_at, &packed_full, cout_))) return rc;
        HIPCHK(hipEventRecord(c->ev[3], cout_));
        HIPCHK(hipEventSynchronize(c->ev[3]));
        t.h2d_ms = ev_ms(c->ev[2], c->sub_ev[0][5]);
        t.d2h_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_sub).count() - t.zstd_ms;
        t.total_ms = t.h2d_ms + t.zstd_ms + t.d2h_ms;
    }
    g_phase_ns[0] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_sub - t_in).count();
    g_phase_ns[1] += (uint64_t)(t.zstd_ms * 1e6);
    g_phase_ns[2] += (uint64_t)(t.d2h_ms * 1e6);
    g_phase_ns[3] += 1;
    return TSX_OK;
}

static int run_batch_inner(tsx_run& r) {
    tsx_ctx* c = r.c;
    const uint32_t n = r.n;
    uint32_t max_len, max_out; size_t in_bytes; bool monotonic;
    // src-side validation first: nothing of the caller's descriptors is touched by a call that fails with TSX_E_INVAL
    int rc = validate(r.descs, n, r.src_size, r.dst_size, r.mode != 2 && !r.packed, &max_len, &max_out, &in_bytes, &monotonic);
    if (rc) return rc;
    size_t out_bytes = r.dst_size;                                      // size of the output area the kernels see
    if (r.packed) {
        // the kernels still write one bound-sized slot per chunk - on the device; only the bytes produced cross PCIe, straight
        // to their final place in the caller's buffer
        const size_t slot = (tsx_transformed_bound(max_len, r.flags) + 63) & ~(size_t)63;
        if (slot >= ((size_t)1 << 32)) return TSX_E_INVAL;
        for (uint32_t i = 0; i < n; i++) { r.descs[i].dst_off = (uint64_t)i * slot; r.descs[i].dst_cap = (uint32_t)slot; }
        out_bytes = (size_t)n * slot; max_out = (uint32_t)slot;
    }
    r.max_len = max_len; r.max_out = max_out;
    // zero-copy output, as in run_combined: the compressor waves of a lean batch write into the caller's buffer when the device can address it.
    // Slot layout only: packing a whole 2048-chunk batch down in place is ~90 ms of one host thread behind the kernel (1004 against 935 ms
    // for the lone batch, measured), where the copy path packs piece by piece while later pieces run; TSX_ZERO_COPY_PACKED=1 takes it anyway.
    uint8_t* zc_dst = nullptr;
    if (r.mode == 0 && r.comp && r.enc && r.fuse_stages && r.host && !getenv("TSX_NO_ZERO_COPY_OUT") &&
        (!r.packed || (r.dst_size >= out_bytes && getenv("TSX_ZERO_COPY_PACKED")))) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, r.dst, 0) == hipSuccess && dp) zc_dst = (uint8_t*)dp;
        else (void)hipGetLastError();
    }
    rc = reserve_or_drain(c, n, max_len, r.mode == 1 ? max_out : 0, r.flags, r.host, in_bytes, zc_dst ? 0 : out_bytes);
    if (rc) return rc;
    r.d_src = r.host ? c->d_in : (const uint8_t*)r.src;
    r.d_dst = zc_dst ? zc_dst : r.host ? c->d_out : (uint8_t*)r.dst;
    hipStream_t st = c->st;
    if (r.mode == 0 && r.comp && reserved_cus()) {                     // a compressing batch leaves the reserved CUs alone
        if (!c->st_fwd) HIPCHK(tsx_compressor_stream(&c->st_fwd, c->dev->hip_id));
        st = c->st_fwd;
    }
    memset(&c->timing, 0, sizeof c->timing);
    // ---- sub-batches: a host-memory batch is cut into pieces whose H2D copy, kernels and D2H copy overlap.  Device-memory batches
    // have nothing to overlap.  Two shapes:
    //  * short kernels (no compression, or the inverse chain): many pieces of >= 64 MiB in order on ONE compute stream, three streams
    //    in all (the frame decoder is bound by per-chunk latency - ~30 ms however few chunks a launch has: its pieces are >= 512 chunks);
    //  * the compressing chain: every chunk is ~0.6 s of one wave whatever the batch size, so the pieces must CO-RESIDE - at most
    //    TSX_COMP_PIECES of them, each on its own compute stream, all copies and launches queued before the host waits for anything.
    //    Piece k's waves start when its share of the input has landed instead of when the whole batch has, and its output travels
    //    back while the later pieces still run; what stays serial is the input copy of the whole batch + one piece's kernel + one
    //    piece's output copy.
    std::vector<tsx_sub> subs;
    const bool comp_fwd = r.mode == 0 && r.comp;
    // Co-resident pieces need a hardware queue per compute stream: with the runtime's default of 4 queues for ALL streams of the process
    // the pieces' kernels and the copy streams' event markers end up behind one another (measured: 945 -> 1300-2400 ms per 2048-chunk
    // batch).  So only when the process runs with GPU_MAX_HW_QUEUES >= 8, or TSX_COMP_PIECES says so.
    uint32_t comp_pieces = 1;
    if (const char* e = getenv("TSX_COMP_PIECES")) { const long v = atol(e); if (v >= 1 && v <= TSX_COMP_PIECES) comp_pieces = (uint32_t)v; }
    else if (const char* q = getenv("GPU_MAX_HW_QUEUES")) { if (atol(q) >= 8) comp_pieces = TSX_COMP_PIECES; }
    if (comp_fwd && comp_pieces > 1 && r.host && !getenv("TSX_COMP_PIECES")) {
        // ... and only from / to pinned memory: a pageable copy is staged by the runtime with a copy kernel, and a copy kernel queued while
        // the chip is full of second-long compressor waves waits for them (measured: 2.4 s per 2048-chunk batch instead of 0.95)
        hipPointerAttribute_t a;
        // (the runtime answers for pageable memory too - hipMemoryTypeUnregistered - instead of failing)
        const bool src_pinned = hipPointerGetAttributes(&a, r.src) == hipSuccess && a.type == hipMemoryTypeHost;
        const bool dst_pinned = hipPointerGetAttributes(&a, r.dst) == hipSuccess && a.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        if (!src_pinned || !dst_pinned) comp_pieces = 1;
    }
    // With zero-copy output (round 4) there is no output copy left to overlap, and what the pieces still buy on the input side (~120 ms of
    // a lone batch's 160 ms copy) they lose in the kernels (four launches of 512 chunks: 770-800 ms against 743) - a lone batch reads 918-935 ms
    // with pieces; with several callers in flight their 4 x 4 compute streams collide on the hardware queues (14.2 against 16.3 GiB/s at
    // 4 callers, profiles/r04_bench_default_run_with_zero_copy_row.json): one launch per batch then.
    if (zc_dst && !getenv("TSX_COMP_PIECES")) comp_pieces = 1;
    const bool pipelined = r.host && monotonic && !(comp_fwd && (!r.fuse_stages || comp_pieces < 2)) && !getenv("TSX_NO_PIPELINE");
    // A fetch of 16 .. 256 chunks (a consumer catching up: ChunkCache.java:159-184 with a large prefetch.max.size) decodes in the block
    // form, whose cost is a ~1.5 ms chain of short kernels + a part proportional to the chunks: cut into up to 8 pieces of >= 8 chunks,
    // spread over the context's compute streams so that the pieces' chains overlap each other, the later pieces' copy-in and the earlier
    // pieces' copy-out.  (Round 3 only cut batches of >= 512 chunks: 64 chunks took 7 ms device resident and 17 ms host to host.)
    c->blk_pieces.clear(); c->last_used_blocks = false;
    const bool inv_blocks = r.mode == 1 && r.comp && pipelined && n >= 16 && c->d_bwork && dec_use_blocks(n, max_out) && !getenv("TSX_NO_DEC_PIECES");
    if (pipelined) {
        size_t budget = TSX_SUB_BYTES;
        size_t max_subs = TSX_MAX_SUBS;
        if (comp_fwd) { max_subs = comp_pieces; budget = in_bytes / comp_pieces + 1; if (budget < TSX_SUB_BYTES) budget = TSX_SUB_BYTES; }
        if (const char* e = getenv("TSX_SUB_BYTES")) { const long long v = atoll(e); if (v > 0) budget = (size_t)v; }     // tests / tuning
        if (in_bytes / budget + 1 > max_subs) budget = in_bytes / max_subs + 1;
        uint32_t min_chunks = (r.comp && !comp_fwd) ? 512u : 1u;
        if (inv_blocks) { max_subs = 8; budget = 0; min_chunks = (n + 7) / 8 < 8 ? 8u : (n + 7) / 8; }
        uint32_t lo = 0;
        while (lo < n) {
            uint32_t hi = lo; size_t bytes = 0;
            while (hi < n && (bytes < budget || hi - lo < min_chunks) && subs.size() + 1 <= max_subs) { bytes += r.descs[hi].src_len; hi++; }
            if (subs.size() + 1 == max_subs) hi = n;
            subs.push_back({lo, hi - lo, (size_t)r.descs[lo].src_off, (size_t)(r.descs[hi - 1].src_off + r.descs[hi - 1].src_len)});
            lo = hi;
        }
    } else subs.push_back({0, n, 0, in_bytes});
    const size_t ns = subs.size();
    const bool multi = (comp_fwd || inv_blocks) && ns > 1;             // pieces side by side: piece k on compute stream k mod TSX_COMP_PIECES
    for (size_t k = 1; k < ns; k++) for (auto& e : c->sub_ev[k]) if (!e) HIPCHK(hipEventCreate(&e));
    hipStream_t* const pcs = (comp_fwd && reserved_cus()) ? c->st_pcf : c->st_pc;
    if (multi) for (size_t k = 1; k < ns && k < TSX_COMP_PIECES; k++) if (!pcs[k - 1]) {
        if (pcs == c->st_pcf) HIPCHK(tsx_compressor_stream(&pcs[k - 1], c->dev->hip_id));
        else HIPCHK(hipStreamCreateWithFlags(&pcs[k - 1], hipStreamNonBlocking));
    }
    auto stream_of = [&](size_t k) { return (multi && k % TSX_COMP_PIECES) ? pcs[k % TSX_COMP_PIECES - 1] : st; };
    HIPCHK(hipEventRecord(c->ev[0], st));
    // A compressing batch whose waves run the whole chain needs no kernel besides the compressor's: the key schedule is built on the
    // host, every wave owns its chunk's status.  (Small kernels around a launch wait for a slot on a chip that is full of second-long
    // compressor waves: gcm_setup's workgroup wants 11 KiB of LDS where 3 KiB per CU are free.)
    const bool lean = comp_fwd && r.enc && r.fuse_stages;
    if (r.enc) {
        // The key schedule is built on the host for every encrypting / decrypting batch (~10 us with the host's carry-less multiplier).
        // Lean batches leave it in pinned memory (the waves fetch it); the batch GCM kernels - decryption on the fetch path, encryption
        // without compression - get it as ONE 21 KB copy in front of them instead of a raw-key copy + gcm_setup_kernel: that kernel was 0.17
        // of a single-chunk fetch's 1.7 ms and, on a busy device, one more small kernel waiting for a slot behind compressor waves
        // (VERDICT r3 #6).  TSX_GCM_SETUP_KERNEL=1 keeps the kernel (its tests; both produce the same schedule).
        static const bool setup_kernel = getenv("TSX_GCM_SETUP_KERNEL") != nullptr;
        if (lean || !setup_kernel) {
            tsx_gcm_key_build_host(r.params->key, r.params->aad, r.params->aad_len, c->h_key);
            if (!lean) HIPCHK(hipMemcpyAsync(c->d_key, c->h_key, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, st));
        } else {
            memcpy(c->h_keyraw, r.params->key, 32); memcpy(c->h_keyraw + 32, r.params->aad, 64);
            HIPCHK(hipMemcpyAsync(c->d_keyraw, c->h_keyraw, 96, hipMemcpyHostToDevice, st));
            tsx_launch_gcm_setup(st, c->dev->d_aes, c->d_keyraw, c->d_keyraw + 32, r.params->aad_len, c->d_key);
        }
        if (multi) HIPCHK(hipEventRecord(c->ev_key, st));
    }
    if (r.host) HIPCHK(hipEventRecord(c->ev[2], c->st_in));
    size_t packed_at = 0; bool packed_full = false;
    auto enqueue_piece = [&](size_t k) -> int {
        const tsx_sub& sb = subs[k];
        hipEvent_t* e = c->sub_ev[k];
        hipStream_t ks = stream_of(k);
        if (r.host) {
            // pageable memory is staged by the runtime (the call returns when the source has been read); memory pinned with
            // tsx_host_register goes by DMA and the call returns at once - either way the copy overlaps the kernels of earlier pieces
            if (sb.in_hi > sb.in_lo) HIPCHK(hipMemcpyAsync(c->d_in + sb.in_lo, (const uint8_t*)r.src + sb.in_lo, sb.in_hi - sb.in_lo, hipMemcpyHostToDevice, c->st_in));
            HIPCHK(hipEventRecord(e[5], c->st_in));
            HIPCHK(hipStreamWaitEvent(ks, e[5], 0));
        }
        if (multi && ks != st && r.enc) HIPCHK(hipStreamWaitEvent(ks, c->ev_key, 0));
        return launch_stages(r, sb, e, ks);
    };
    // The restored chunks of a fetch are what crosses PCIe (4 MiB each against 1.3 MB in): one copy stream moves them at ~31 GB/s - 8.7 of a
    // 64-chunk window's 10.4 ms; the pieces' copies alternate between two streams (TSX_DEC_OUT_STREAMS=1: one).
    bool out2 = inv_blocks && multi && r.host && !(getenv("TSX_DEC_OUT_STREAMS") && atoi(getenv("TSX_DEC_OUT_STREAMS")) < 2);
    if (out2 && !c->st_out2 && hipStreamCreateWithFlags(&c->st_out2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->st_out2 = nullptr; out2 = false; }
    auto collect_piece = [&](size_t k) -> int {
        const tsx_sub& sb = subs[k];
        HIPCHK(hipEventSynchronize(c->sub_ev[k][4]));                   // descriptors of piece k are on the host
        memcpy(r.descs + sb.lo, c->h_descs + sb.lo, (size_t)sb.n * sizeof(tsx_chunk_desc));
        if (zc_dst) {                                                   // the bytes are where they belong; a packed batch is packed down in place
            if (r.packed) for (uint32_t i = sb.lo; i < sb.lo + sb.n; i++) {
                tsx_chunk_desc& d = r.descs[i];
                const size_t slot_off = d.dst_off;
                d.dst_off = packed_at;
                if (d.status != TSX_OK) { d.dst_len = 0; continue; }
                if (d.dst_len && slot_off != packed_at) memmove((uint8_t*)r.dst + packed_at, (const uint8_t*)r.dst + slot_off, d.dst_len);
                packed_at += d.dst_len;
            }
            return TSX_OK;
        }
        if (r.host && r.mode != 2) return copy_back(r, sb, &packed_at, &packed_full, (out2 && (k & 1)) ? c->st_out2 : c->st_out);
        return TSX_OK;
    };
    if (multi) {
        for (size_t k = 0; k < ns; k++) if ((rc = enqueue_piece(k))) return rc;
        for (size_t k = 0; k < ns; k++) if ((rc = collect_piece(k))) return rc;
        for (size_t k = 1; k < ns; k++) HIPCHK(hipStreamWaitEvent(st, c->sub_ev[k][4], 0));   // the batch's end event covers every piece
    } else {
        // software pipeline on the host: copy-in + kernels of piece k are queued before the host waits for piece k - 1's descriptors
        // (they say how many bytes each chunk produced) and queues its copy-out
        for (size_t k = 0; k <= ns; k++) {
            if (k < ns && (rc = enqueue_piece(k))) return rc;
            if (k > 0 && (rc = collect_piece(k - 1))) return rc;
        }
    }
    if (r.host) HIPCHK(hipEventRecord(c->ev[3], c->st_out));
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipStreamSynchronize(st));
    if (r.host) { HIPCHK(hipStreamSynchronize(c->st_in)); HIPCHK(hipStreamSynchronize(c->st_out)); if (out2) HIPCHK(hipStreamSynchronize(c->st_out2)); }
    HIPCHK(hipGetLastError());
    tsx_timing& t = c->timing;
    float zmax = 0;
    for (size_t k = 0; k < ns; k++) {
        hipEvent_t* e = c->sub_ev[k];
        const float a = ev_ms(e[0], e[1]), b = ev_ms(e[1], e[2]), d = ev_ms(e[2], e[3]), f = ev_ms(e[3], e[4]);
        if (r.mode == 2) t.crc_ms += a;
        else if (r.mode == 0) { t.crc_ms += (r.flags & TSX_CRC) ? a : 0; if (multi) { if (b > zmax) zmax = b; } else t.zstd_ms += r.comp ? b : 0; t.gcm_ms += d; }
        else { t.gcm_ms += r.enc ? b : 0; t.unzstd_ms += d; t.crc_ms += (r.flags & TSX_CRC) ? f : 0; }
    }
    if (multi) t.zstd_ms = zmax;                                        // co-resident pieces: the longest launch, not their sum
    t.total_ms = ev_ms(c->ev[0], c->ev[1]);
    if (r.host) {
        // with pieces in flight the copies overlap the kernels: h2d_ms / d2h_ms are the spans of the copy streams, not additive
        t.h2d_ms = ev_ms(c->ev[2], c->sub_ev[ns - 1][5]);
        t.d2h_ms = r.mode != 2 ? ev_ms(c->sub_ev[0][4], c->ev[3]) : 0;
        const float tail = ev_ms(c->ev[0], c->ev[3]);
        if (r.mode != 2 && tail > t.total_ms) t.total_ms = tail;
    }
    return TSX_OK;
}

static int run_batch(tsx_ctx* c, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src, size_t src_size, void* dst,
                     size_t dst_size, int mem_kind, int mode /*0 transform, 1 detransform, 2 crc only*/, bool combined = false) {
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
    tsx_device_scope keep;
    if (hipSetDevice(c->dev->hip_id) != hipSuccess) return TSX_E_DEVICE;
    tsx_run r{};
    r.c = c; r.params = params; r.descs = descs; r.n = n; r.src = src; r.dst = dst; r.src_size = src_size; r.dst_size = dst_size; r.mem_kind = mem_kind; r.mode = mode;
    r.flags = flags; r.host = mem_kind != TSX_MEM_DEVICE; r.packed = packed;
    r.enc = mode != 2 && (flags & TSX_ENCRYPT); r.comp = mode != 2 && (flags & TSX_COMPRESS);
    r.fuse_stages = r.comp && !getenv("TSX_STAGES_SEPARATE");
    r.combined = combined && mode == 0 && r.comp && r.fuse_stages && !getenv("TSX_NO_COMBINE");
    const int rc = r.combined ? run_combined(r) : run_batch_inner(r);
    if (r.combined) {
        // nothing of the key was uploaded (every wave took and wiped its own copy of the schedule); what is left is the pinned original -
        // wiped once nothing of this call can still be running
        if (rc != TSX_OK) {
            (void)hipGetLastError();
            if (c->dev->comb) { for (auto& q : c->dev->comb->copy_in_s) if (q) (void)hipStreamSynchronize(q);
                                for (auto& q : c->dev->comb->copy_out_s) if (q) (void)hipStreamSynchronize(q);
                                for (uint32_t i = 0; i < c->dev->comb->nlanes; i++) (void)hipStreamSynchronize(c->dev->comb->lane[i].st); }
        }
        if (r.enc) { memset(c->h_keyraw, 0, 128); memset(c->h_key, 0, sizeof(tsx_gcm_key)); }
        return rc;
    }
    // Whatever happened: nothing of this call is still in flight when it returns (the copies reference the caller's buffers), and
    // the data key does not stay behind in a context that may serve another segment next (the native side zeroises its
    // copy; the round keys and H powers are as good as the key).
    if (r.enc) {
        hipStreamSynchronize(c->st);                                      // (a wave may still be reading the pinned key schedule on an error path)
        for (auto& q : c->st_pc) if (q) hipStreamSynchronize(q);
        for (auto& q : c->st_pcf) if (q) hipStreamSynchronize(q);
        if (c->st_fwd) hipStreamSynchronize(c->st_fwd);
        memset(c->h_keyraw, 0, 128); memset(c->h_key, 0, sizeof(tsx_gcm_key));
        if (!(mode == 0 && r.comp && r.fuse_stages)) {                    // lean batches uploaded nothing: every wave wiped its own copy of the schedule
            if (getenv("TSX_GCM_SETUP_KERNEL")) hipMemcpyAsync(c->d_keyraw, c->dev->h_zeros, 128, hipMemcpyHostToDevice, c->st);      // (the raw key only travels with the setup kernel)
            hipMemcpyAsync(c->d_key, c->dev->h_zeros, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, c->st);                           // wiped by copies, not kernels
        }
    }
    hipStreamSynchronize(c->st_in); hipStreamSynchronize(c->st); hipStreamSynchronize(c->st_out);
    if (c->st_out2) hipStreamSynchronize(c->st_out2);
    for (auto& q : c->st_pc) if (q) hipStreamSynchronize(q);
    for (auto& q : c->st_pcf) if (q) hipStreamSynchronize(q);
    if (c->st_fwd) hipStreamSynchronize(c->st_fwd);
    if (rc != TSX_OK) (void)hipGetLastError();
    return rc;
}

static int with_ctx(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src, size_t src_size, void* dst,
                    size_t dst_size, int mem_kind, int mode) {
    if (ctx) return run_batch(ctx, params, descs, n, src, src_size, dst, dst_size, mem_kind, mode);
    int rc = TSX_OK;
    tsx_ctx* c = pool_acquire(&rc);
    if (!c) return rc;
    rc = run_batch(c, params, descs, n, src, src_size, dst, dst_size, mem_kind, mode, true);
    pool_release(c);
    return rc;
}

extern "C" int tsx_transform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src,
                                   size_t src_size, void* dst, size_t dst_size, int mem_kind) {
    return with_ctx(ctx, params, descs, n, src, src_size, dst, dst_size, mem_kind, 0);
}

extern "C" int tsx_detransform_batch(tsx_ctx* ctx, const tsx_batch_params* params, tsx_chunk_desc* descs, uint32_t n, const void* src,
                                     size_t src_size, void* dst, size_t dst_size, int mem_kind) {
    return with_ctx(ctx, params, descs, n, src, src_size, dst, dst_size, mem_kind, 1);
}

extern "C" int tsx_crc32c_batch(tsx_ctx* ctx, tsx_chunk_desc* descs, uint32_t n, const void* src, size_t src_size, int mem_kind) {
    return with_ctx(ctx, nullptr, descs, n, src, src_size, nullptr, 0, mem_kind, 2);
}

// Test hook (not part of the ABI in include/tsxform.h): OR of every byte of the context's key material on the device - 0 after
// any batch, whatever its outcome.
extern "C" int tsx_debug_key_residue(tsx_ctx* c) {
    if (!c) return TSX_E_INVAL;
    tsx_device_scope keep;
    if (hipSetDevice(c->dev->hip_id) != hipSuccess) return TSX_E_DEVICE;
    std::vector<uint8_t> h(sizeof(tsx_gcm_key) + 128);
    if (hipMemcpy(h.data(), c->d_key, sizeof(tsx_gcm_key), hipMemcpyDeviceToHost) != hipSuccess) return TSX_E_DEVICE;
    if (hipMemcpy(h.data() + sizeof(tsx_gcm_key), c->d_keyraw, 128, hipMemcpyDeviceToHost) != hipSuccess) return TSX_E_DEVICE;
    int acc = 0;
    for (uint8_t b : h) acc |= b;
    return acc;
}

// Test hook (not part of the ABI): launches the device's combiner has made and the batches they carried.
extern "C" int tsx_debug_combiner_stats(int device_index, uint64_t* groups, uint64_t* members) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    tsx_combiner* cb = g_devs[device_index].comb.get();
    if (groups) *groups = cb ? cb->groups : 0;
    if (members) *members = cb ? cb->members : 0;
    return TSX_OK;
}

// Test hook (not part of the ABI): how many of the first n chunks of the context's LAST detransform batch were decoded by the
// block-parallel form (the rest went through the chunk-serial kernel); -1 when that batch did not use the form at all.
// test hook: how many idle pooled contexts of a device hold a block-form decoder workspace (bounded by TSX_POOL_MAX_IDLE_BWORK)
extern "C" int tsx_debug_pool_bwork(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    int k = 0;
    for (const tsx_ctx* c : g_devs[device_index].idle) if (c->d_bwork) k++;
    return k;
}
// test hook: block-form decoder launches of the context's last batch (> 1: the batch was cut into co-resident pieces)
extern "C" int tsx_debug_blockmode_pieces(tsx_ctx* c) { return c ? (int)c->blk_pieces.size() : TSX_E_INVAL; }
extern "C" int tsx_debug_blockmode_chunks(tsx_ctx* c, uint32_t n) {
    if (!c) return TSX_E_INVAL;
    if (!c->d_bwork || !c->last_used_blocks) return -1;
    tsx_device_scope keep;
    if (hipSetDevice(c->dev->hip_id) != hipSuccess) return TSX_E_DEVICE;
    int cnt = 0;
    for (const auto& pc : c->blk_pieces) {                              // every launch of the batch laid its chunks' headers out for itself
        uint32_t stride = 0;
        const uint32_t* skip = tsx_zstd_blockmode_skip((const uint8_t*)c->d_bwork + tsx_zstd_blockmode_bytes(pc.first, c->last_max_out), &stride);
        for (uint32_t i = 0; i < pc.second && pc.first + i < n; i++) {
            uint32_t w = 0;
            if (hipMemcpy(&w, skip + (size_t)i * stride, 4, hipMemcpyDeviceToHost) != hipSuccess) return TSX_E_DEVICE;
            cnt += w == 1;
        }
    }
    return cnt;
}

// Pins a caller buffer that will be used for TSX_MEM_HOST / TSX_MEM_HOST_PACKED batches again and again (the JVM side registers its
// per-thread direct ByteBuffers once): copies from / to it go by DMA and overlap fully instead of being staged by the runtime.
// Portable: the pinning holds for every device of the node, whichever one the pool picks for a batch.
extern "C" int tsx_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return TSX_E_INVAL;
    { std::lock_guard<std::mutex> lk(g_mu); if (g_devs.empty()) return TSX_E_DEVICE; }
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { tsx_set_err("hipHostRegister", e); (void)hipGetLastError(); return TSX_E_DEVICE; }
    return TSX_OK;
}
extern "C" int tsx_host_unregister(void* p) {
    if (!p) return TSX_E_INVAL;
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { tsx_set_err("hipHostUnregister", e); (void)hipGetLastError(); return TSX_E_DEVICE; }
    return TSX_OK;
}

// ---- device memory helpers -------------------------------------------------------------------------
static int set_dev(int device_index) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    return hipSetDevice(g_devs[device_index].hip_id) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_device_malloc(int device_index, size_t bytes, void** out) {
    if (!out) return TSX_E_INVAL;
    tsx_device_scope keep;
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMalloc(out, bytes) == hipSuccess ? TSX_OK : TSX_E_NOMEM;
}
extern "C" int tsx_device_free(int device_index, void* p) {
    tsx_device_scope keep;
    int rc = set_dev(device_index); if (rc) return rc;
    return hipFree(p) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_memcpy_h2d(int device_index, void* d, const void* s, size_t bytes) {
    tsx_device_scope keep;
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMemcpy(d, s, bytes, hipMemcpyHostToDevice) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_memcpy_d2h(int device_index, void* d, const void* s, size_t bytes) {
    tsx_device_scope keep;
    int rc = set_dev(device_index); if (rc) return rc;
    return hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost) == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
