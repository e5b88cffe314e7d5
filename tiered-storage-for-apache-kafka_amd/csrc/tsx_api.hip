// C-ABI front end of libtsxform: device/context management, the compressor service's host side and the batch pipelines.
// See include/tsxform.h for the contract and the reference call sites each entry point replaces.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
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

// ---- configuration -----------------------------------------------------------------------------------------------------------------
// Read ONCE, in tsx_init: tsx_init_ex's tsx_config first, then the environment of the process for the few settings a deployment may
// want to change without touching code (INTEGRATION.md 5 lists them).  No entry point on a data path reads the environment.  The
// fields under "test hooks" have no environment variable: tests and measurement tools set them through tsx_debug_config().
struct tsx_cfg {
    uint32_t reserved_cus = 0xFFFFFFFFu;  // compute units the compressor service leaves to everything else (0xFFFFFFFF: one per shader engine); TSX_FETCH_RESERVED_CUS
    uint32_t svc_max_launch_ms = 60000;   // age limit of one launch of the service kernel (0 = none); TSX_SERVICE_MAX_LAUNCH_MS
    uint32_t svc_idle_exit_us = 2000;     // the service kernel ends when it has had nothing to do for this long (callers in a closed loop need ~1 ms to come back)
    uint32_t fetch_quiet_ms = 2000;       // tsx_config.fetch_quiet_ms: the reserved CUs work for the compressor too (guest waves) once no fetch has been seen for this long; 0 = never; TSX_FETCH_QUIET_MS
    uint32_t svc_keep_waves = 0;          // tsx_config.fetch_shared_cu_waves: compressor waves that stay on a reserved CU all the same; TSX_FETCH_SHARED_CU_WAVES
    long long pool_idle_bytes = -1;       // idle pooled workspace kept per device (-1: 4/9 of its memory); TSX_POOL_IDLE_BYTES
    uint32_t zstd_sched = 0;              // parser speculation schedule k0 | k1 << 8 (0 = the kernel's default; same bytes); TSX_ZSTD_SCHED
    bool debug = false;                   // TSX_DEBUG: HIP failures go to stderr as they happen
    bool allow_any_arch = false;          // TSX_ALLOW_ANY_ARCH: the CPU test harness
    // ---- test hooks (tsx_debug_config) ----
    uint32_t dec_block_chunks = 256;      // largest detransform batch that takes the block-parallel decoder form (0 = never)
    uint32_t comp_pieces = 4;             // members a compressing host-memory batch is cut into (input copy of piece k + 1 overlaps piece k's waves)
    long long sub_bytes = 0;              // input bytes per piece of the staging pipeline (0 = TSX_SUB_BYTES)
    bool stages_separate = false;         // one launch per stage instead of the whole chain in the compressor wave
    bool no_pipeline = false;             // host-memory batches in one piece
    bool no_zero_copy_out = false;        // never let the waves write into the caller's buffer
    bool zero_copy_packed = false;        // explicit contexts: packed output in place too
    bool gcm_setup_kernel = false;        // key schedule by gcm_setup_kernel instead of on the host
    bool no_dec_pieces = false;           // block-form fetches in one piece
    uint32_t svc_waves_per_cu = 0;        // workgroups of a service launch per CU (0 = what the runtime says is resident at once; measurements only)
    bool svc_normal_priority = false;     // the service's stream like any other (default: the device's LOWEST stream priority, a hardware queue of its own pool)
    bool trace = false;                   // timestamps of a batch's phases on stderr (tools/fetch_block_probe.py)
    uint32_t svc_idle_nap_max = 0;        // measurements: tsx_svc_launch.idle_nap_max
    bool svc_no_spread = false;           // measurements: tickets to whoever asks first (no spreading of a partial load over the CUs)
    uint32_t svc_guest_looks = 3;         // measurements: which looks at the yield word guest waves make (1: before every block, 2: while idle)
};
static tsx_cfg g_cfg;

static thread_local char g_last_err[256];
static void tsx_set_err(const char* what, hipError_t e) {
    snprintf(g_last_err, sizeof g_last_err, "%s failed: %s", what, hipGetErrorString(e));
    if (g_cfg.debug) fprintf(stderr, "[tsxform] %s\n", g_last_err);
}

static void cfg_from_env(tsx_cfg& c) {
    if (const char* e = getenv("TSX_FETCH_RESERVED_CUS")) { const long v = atol(e); c.reserved_cus = (uint32_t)(v < 0 ? 0 : v > 128 ? 128 : v); }
    if (const char* e = getenv("TSX_FETCH_SHARED_CU_WAVES")) { const long v = atol(e); c.svc_keep_waves = (uint32_t)(v < 0 ? 0 : v > 8 ? 8 : v); }
    if (const char* e = getenv("TSX_FETCH_QUIET_MS")) { const long v = atol(e); if (v >= 0) c.fetch_quiet_ms = (uint32_t)v; }
    if (const char* e = getenv("TSX_SERVICE_MAX_LAUNCH_MS")) { const long v = atol(e); if (v >= 0) c.svc_max_launch_ms = (uint32_t)v; }
    if (const char* e = getenv("TSX_POOL_IDLE_BYTES")) { const long long v = atoll(e); if (v >= 0) c.pool_idle_bytes = v; }
    if (const char* e = getenv("TSX_ZSTD_SCHED")) { unsigned a = 0, b = 0; if (sscanf(e, "%u,%u", &a, &b) == 2 && a >= 1 && a <= 59 && b >= 1 && b <= 59) c.zstd_sched = a | b << 8; }
    c.debug = getenv("TSX_DEBUG") != nullptr;
    c.allow_any_arch = getenv("TSX_ALLOW_ANY_ARCH") != nullptr;
}

// Test / measurement hook (not part of the ABI in include/tsxform.h): set one configuration field, return its previous value
// (TSX_E_INVAL: no such field).  Takes effect for calls made afterwards; reserved_cus only for a tsx_init made afterwards.
extern "C" long long tsx_debug_config(const char* key, long long value) {
    if (!key) return TSX_E_INVAL;
#define CFG_FIELD(name, T) if (!strcmp(key, #name)) { const long long old = (long long)g_cfg.name; g_cfg.name = (T)value; return old; }
    CFG_FIELD(reserved_cus, uint32_t) CFG_FIELD(svc_max_launch_ms, uint32_t) CFG_FIELD(svc_idle_exit_us, uint32_t) CFG_FIELD(pool_idle_bytes, long long)
    CFG_FIELD(zstd_sched, uint32_t) CFG_FIELD(dec_block_chunks, uint32_t) CFG_FIELD(comp_pieces, uint32_t) CFG_FIELD(sub_bytes, long long)
    CFG_FIELD(stages_separate, bool) CFG_FIELD(no_pipeline, bool) CFG_FIELD(no_zero_copy_out, bool) CFG_FIELD(zero_copy_packed, bool)
    CFG_FIELD(gcm_setup_kernel, bool) CFG_FIELD(no_dec_pieces, bool) CFG_FIELD(debug, bool) CFG_FIELD(svc_normal_priority, bool) CFG_FIELD(svc_waves_per_cu, uint32_t) CFG_FIELD(svc_keep_waves, uint32_t) CFG_FIELD(fetch_quiet_ms, uint32_t) CFG_FIELD(trace, bool) CFG_FIELD(svc_guest_looks, uint32_t) CFG_FIELD(svc_no_spread, bool) CFG_FIELD(svc_idle_nap_max, uint32_t)
#undef CFG_FIELD
    return TSX_E_INVAL;
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
struct tsx_device;

// ---- the compressor service, host side (device side: zstd_enc.hip, zstd_service_kernel; layout: tsx_internal.h) -----------------------
// Rounds 2-4 launched one compressor kernel per batch (or per group of callers: the "launch combiner") and let the hardware's
// dispatcher hand workgroups to freed wave slots.  Two things followed from a launch being the unit of work: the chip ran in
// generations (a batch's stragglers held its hardware queue while slots sat empty; 18.0 GiB/s in a timed region against 18.7-20.4
// continuously fed), and everything that is not a compressor wave starved - a freed 6.7 KB slot is refilled by the dispatcher before a
// decoder workgroup finds three neighbouring ones: a fetch under upload load took 1-65 s (profiles/r04_mixed_load.txt).  Now a
// compressing batch is a MEMBER of its device's queue: tickets in pinned memory, persistent waves that pull them, per-member completion
// flags.  One kernel, on one stream that carries nothing else; it is (re)started by whoever publishes work and finds it gone, and by
// the waiting callers' watchdog (a launch that ended - idle, age limit, every wave on a reserved CU - with tickets still unserved).
struct tsx_svc_member { uint64_t id; uint32_t first, n; uint16_t slot; bool done; const uint32_t* h_flag; };
struct tsx_service {
    std::mutex mu; std::condition_variable cv;
    tsx_svc_host* h = nullptr; tsx_svc_host* hd = nullptr;          // the queue in pinned host memory: host view, device alias
    tsx_svc_dev* d = nullptr;
    uint32_t* h_zero = nullptr;                                      // pinned zero word (resets of device words travel as copies, not kernels)
    hipStream_t st = nullptr;                                        // the service kernel's stream: launches only - no events, no copies while it runs
    uint32_t launch_id = 0;                                          // id of the last launch made
    bool launched = false;                                           // a launch is out whose end this side has not seen yet
    bool stop_dirty = false;                                         // the device's stop word must be cleared in front of the next launch
    uint32_t paused = 0;                                             // > 0: no launches (memory management in progress)
    uint32_t grid = 0, cu_keys = 0, cus = 0, cus_reserved = 0, waves_per_cu = 0, resident = 0, engines = 0;
    uint32_t published = 0;
    uint64_t next_id = 1;
    std::deque<tsx_svc_member> out;                                  // members published and not yet retired, oldest first
    std::vector<uint16_t> free_slots; uint16_t slot_gen[TSX_SVC_MEMBERS] = {0};
    uint64_t launches = 0, watchdog_launches = 0, members = 0, chunks = 0, rotations = 0; double kernel_ms = 0;
    bool rotating = false;                                           // a waiting fetch has asked the running launch to end (svc_rotate)
    // The reservation follows the traffic.  "Foreground" = every batch that runs ordinary kernels (fetches above all): while one is in flight,
    // and for fetch_quiet_ms after the last one, tsx_svc_host.yield is raised - guest waves on the reserved CUs hand their chunks back and
    // leave (<= one block of their chunk later, ~30 ms), launches made meanwhile leave the reserved CUs alone.  A device that only uploads
    // compresses on every CU.
    std::atomic<uint32_t> fg_inflight{0};
    std::atomic<int64_t> fg_last_ns{INT64_MIN / 2};                  // steady clock at the end of the last foreground batch
    uint64_t guest_launches = 0, readmissions = 0;
    // Guests come in launches of their own, next to the launch they help (tsx_svc_launch.guest_launch): on a second stream of the lowest priority
    hipStream_t st_g = nullptr;
    uint32_t g_launch_id = 0; bool g_launched = false; int64_t g_last_ns = INT64_MIN / 2;
    int64_t launch_ns = 0;                                           // steady clock at the last launch of the service kernel itself
    std::vector<void*> deferred_dev, deferred_host;                  // frees that wait for the kernel to be gone (svc_free_*)
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
    std::vector<std::pair<void*, size_t>> spare_bwork;       // block-form decoder workspaces that left their context (pool_release), for the next one
    std::unique_ptr<tsx_service> svc;
    // copy streams of the context-less compressing calls, shared by the callers: ONE of each.  With a stream per caller a segment's copies
    // stood behind other callers' in the engines' queues anyway, and more streams measured worse (profiles/r04_broker_shape_experiments.txt).
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    uint32_t in_use = 0;
    uint64_t batches = 0;
};

#define TSX_MAX_SUBS 64                 /* sub-batches of one host-memory batch (staging pipeline) */
#define TSX_SUB_BYTES ((size_t)64 << 20) /* input bytes per sub-batch: >= 1000 workgroups of the GCM / CRC kernels */
#define TSX_POOL_MAX_IDLE 32            /* idle pooled contexts kept per device (a broker: >= 10 RLM threads + read-ahead helpers + the fetch pool) ... */
// ... as long as their workspaces together stay under tsx_device.idle_cap = 4/9 of the device's memory (128 of the MI355X's 288 GB; a smaller
// device or several processes per GPU get their share: tsx_config.pool_idle_bytes); the rest are destroyed on release, and an
// allocation that fails drains the idle pool and is tried again (reserve_or_drain) - cached memory is never the reason for TSX_E_NOMEM.
#define TSX_POOL_MAX_IDLE_BWORK 4       /* idle contexts that keep their block-form decoder workspace (37 MiB per 4 MiB chunk: 9.4 GiB for a segment) */
#define TSX_COMP_PIECES_MAX 8           /* members of one compressing host-memory batch (a completion counter + flag each) */

struct tsx_ctx {
    int dev_index = 0;
    tsx_device* dev = nullptr;
    hipStream_t st = nullptr;                      // kernels (+ descriptor copies)
    hipStream_t st_in = nullptr, st_out = nullptr; // H2D / D2H of the host-memory staging pipeline
    hipStream_t st_out2 = nullptr;                 // second D2H stream of a fetch cut into pieces (odd pieces; created on first use)
    hipStream_t st_pc[3] = {nullptr};              // compute streams of pieces 1.. of a block-form fetch cut into pieces (created on first use)
    hipEvent_t ev_key = nullptr;                   // key schedule ready (the piece streams wait for it)
    // device workspace (grown on demand)
    tsx_chunk_desc* d_descs = nullptr; size_t descs_cap = 0;
    tsx_chunk_desc* h_descs = nullptr;             // pinned mirror of the descriptors: no pageable copy ever sits in a stream
    tsx_chunk_desc* hd_descs = nullptr;            // ... as the device addresses it: compressor waves read and write it in place
    uint8_t* h_keyraw = nullptr;                   // pinned 128 bytes: key + aad on their way in (wiped after the batch)
    tsx_gcm_key* h_key = nullptr;                  // pinned: the key schedule built on the host (wiped after the batch)
    tsx_gcm_key* hd_key = nullptr;                 // ... as the device addresses it (every compressor wave takes its own copy, tsx_chain_fuse.key_on_host)
    uint32_t* d_segdone = nullptr;                 // per member of this context's batch: chunks that are done (device counters, self-resetting)
    uint32_t* h_segflag = nullptr;                 // ... and the words the last of them raise (pinned; hd_segflag = the device's address)
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
    uint32_t last_members = 0;                     // members the last compressing batch went as (test hook)
    bool last_zero_copy = false;                   // ... and whether its waves wrote into the caller's buffer (test hook)
    bool key_wiped = false;                        // the batch's own wipe_key_kernel has cleared d_key / d_keyraw
};

static std::mutex g_mu;
static std::vector<tsx_device> g_devs;
static uint32_t g_rr = 0;
static thread_local int t_dev_hint = -1;
static const char kUninitVersion[] = "tsxform 0.5 (gfx950 HIP; uninitialised)";
static char g_version_buf[2][512];
static unsigned g_version_gen = 0;
static std::atomic<const char*> g_version{kUninitVersion};
// buffers pinned through tsx_host_register: the only host ranges the device is KNOWN to address end to end (zero-copy output)
static std::mutex g_reg_mu;
static std::vector<std::pair<uintptr_t, size_t>> g_registered;

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

// ---- service: whose turn the reserved CUs are ----------------------------------------------------------------------------------------
static int64_t steady_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// A batch of ordinary kernels begins / is over (no lock: this is the fetch path).  The counter first, the word second - svc_launch_locked
// clears the word first and looks at the counter second, so one of the two always sees the other.
static void svc_foreground_begin(tsx_device* dev) {
    if (!dev->svc) return;
    dev->svc->fg_inflight.fetch_add(1, std::memory_order_seq_cst);
    __atomic_store_n(&dev->svc->h->yield, 1u, __ATOMIC_SEQ_CST);
}
// traffic = false: something that only needed room while it ran (a new context's first small copies, a helper copy) - it does not count as
// a fetch having been seen: the next member published finds the device quiet and asks the launch, whose guests have left, to make way for one
// that uses every CU again (svc_submit, readmission)
static void svc_foreground_end(tsx_device* dev, bool traffic = true) {
    if (!dev->svc) return;
    if (traffic) dev->svc->fg_last_ns.store(steady_ns(), std::memory_order_seq_cst);
    dev->svc->fg_inflight.fetch_sub(1, std::memory_order_seq_cst);
}
static bool svc_quiet(const tsx_service& s) {
    return g_cfg.fetch_quiet_ms != 0 && s.fg_inflight.load(std::memory_order_seq_cst) == 0 &&
           steady_ns() - s.fg_last_ns.load(std::memory_order_seq_cst) > (int64_t)g_cfg.fetch_quiet_ms * 1000000;
}

// A member is about to be published (mu held): the yield word follows the device's quietness - down when no fetch has been seen for fetch_quiet_ms
// (word first, counter second: svc_foreground_begin does it the other way round, one of the two sees the other), up otherwise.  Guests read it, and
// so does a wave of the launch itself that the hardware's scheduler has restored onto a reserved CU: on a quiet device it finishes its chunk there.
static void svc_note_quiet_locked(tsx_service& s) {
    if (!s.cus_reserved) return;
    if (svc_quiet(s)) {
        if (__atomic_load_n(&s.h->yield, __ATOMIC_SEQ_CST) == 0u) return;
        __atomic_store_n(&s.h->yield, 0u, __ATOMIC_SEQ_CST);
        if (s.fg_inflight.load(std::memory_order_seq_cst) != 0) __atomic_store_n(&s.h->yield, 1u, __ATOMIC_SEQ_CST);
    } else if (__atomic_load_n(&s.h->yield, __ATOMIC_SEQ_CST) == 0u) __atomic_store_n(&s.h->yield, 1u, __ATOMIC_SEQ_CST);      // (fetch_quiet_ms was changed under a lowered word)
}

// ---- service: lifetime ------------------------------------------------------------------------------------------------------------------
// Is the service kernel of this device still out?  (mu held.)  When its end is seen for the first time, its duration joins the statistics
// and what waited for it to be gone is freed: hipFree / hipHostFree wait for EVERY stream of the device, i.e. for a kernel that lives
// as long as uploads go on - nothing in this library frees device or pinned memory while that kernel may be running (svc_free_*).
static bool svc_guests_running_locked(tsx_service& s) {
    if (!s.g_launched) return false;
    if (__atomic_load_n(&s.h->g_ended_launch, __ATOMIC_ACQUIRE) != s.g_launch_id) return true;
    s.g_launched = false;
    return false;
}
static void svc_collect_locked(tsx_service& s) {                         // (nothing of the service is alive: what waited for that is freed)
    if (s.launched || s.g_launched) return;
    for (void* p : s.deferred_dev) (void)hipFree(p);
    for (void* p : s.deferred_host) (void)hipHostFree(p);
    s.deferred_dev.clear(); s.deferred_host.clear();
}
static bool svc_running_locked(tsx_service& s) {
    if (!s.launched) return false;
    // the launch's last wave says so itself (tsx_svc_host.ended_launch): nothing is queued behind the kernel that could be asked
    if (__atomic_load_n(&s.h->ended_launch, __ATOMIC_ACQUIRE) != s.launch_id) return true;
    s.kernel_ms += (double)(s.h->t_last - s.h->t_first) / 1e5;           // 100 MHz ticks
    s.launched = false;
    // a rotation is over with the launch it asked to end - also when a pause overlapped that end (the flag must not survive into the next
    // launch: svc_rotate would stay switched off for its whole life); only the host's stop word stays up while paused
    if (s.rotating) { s.rotating = false; if (!s.paused) __atomic_store_n(&s.h->stop, 0u, __ATOMIC_RELEASE); }   // (the device's copy of the word is cleared in front of the next launch)
    (void)svc_guests_running_locked(s);
    svc_collect_locked(s);
    return false;
}
// Is ANY kernel of the service still out - the launch or its guests?  (What must not free memory, or must wait for the device to be the caller's alone, asks this.)
static bool svc_alive_locked(tsx_service& s) {
    const bool a = svc_running_locked(s), g = svc_guests_running_locked(s);
    if (!a && !g) svc_collect_locked(s);
    return a || g;
}

// Start the service kernel (mu held, kernel known to be gone, device current).
static int svc_launch_locked(tsx_service& s) {
    if (s.paused) return TSX_OK;                                        // whoever paused the service starts it again (svc_resume)
    if (s.stop_dirty) {
        HIPCHK(hipMemcpyAsync(&s.d->stop, s.h_zero, 4, hipMemcpyHostToDevice, s.st));
        s.stop_dirty = false;
    }
    tsx_svc_launch a{};
    a.sched = g_cfg.zstd_sched;
#ifdef HIPEMU
    a.poll_ticks = 0; a.idle_exit_ticks = 0;                             // blocks run one after the other: nobody to wait for
#else
    a.poll_ticks = 500;                                                  // one look at the host's word per 5 us, device-wide
    a.idle_exit_ticks = g_cfg.svc_idle_exit_us * 100u;
#endif
    const uint64_t age = (uint64_t)g_cfg.svc_max_launch_ms * 100000ull;
    a.max_age_ticks_lo = (uint32_t)age; a.max_age_ticks_hi = (uint32_t)(age >> 32);
    a.keep_waves = g_cfg.svc_keep_waves;
    a.idle_nap_max = g_cfg.svc_idle_nap_max;
#ifdef HIPEMU
    a.guest_idle_ticks = 0;
#else
    a.guest_idle_ticks = 1000000;                                        // 10 ms
#endif
    a.spread_cus = g_cfg.svc_no_spread ? 0u : (s.cus > s.cus_reserved ? s.cus - s.cus_reserved : s.cus);
#ifdef HIPEMU
    // (the CPU harness runs one kernel at a time, its blocks one after the other: guests are part of the launch there, which is how its tests reach
    //  the guests' code; on the device they come in launches of their own - svc_try_guests_locked)
    if (s.cus_reserved && svc_quiet(s)) {
        __atomic_store_n(&s.h->yield, 0u, __ATOMIC_SEQ_CST);             // word first, counter second (svc_foreground_begin)
        if (s.fg_inflight.load(std::memory_order_seq_cst) != 0) __atomic_store_n(&s.h->yield, 1u, __ATOMIC_SEQ_CST);
        else a.guests = 1u | (g_cfg.svc_guest_looks << 1);               // (bit 1: guests look at the yield word before every block, bit 2: and while idle)
    }
#endif
    (void)hipGetLastError();
    a.launch_id = s.launch_id + 1;
    tsx_launch_zstd_service(s.st, s.hd, s.d, s.grid, a);
    if (hipGetLastError() != hipSuccess) { snprintf(g_last_err, sizeof g_last_err, "launch of the compressor service kernel failed"); return TSX_E_DEVICE; }
    s.launch_id = a.launch_id;
    s.launched = true; s.launches++; s.guest_launches += a.guests ? 1u : 0u;
    s.launch_ns = steady_ns();
    return TSX_OK;
}

// Guests (mu held, device current): a launch of as many one-wave workgroups as the reserved CUs hold, made when the running launch has more chunks
// queued than waves, no fetch has been seen for fetch_quiet_ms and no guests are out.  With every other slot of the chip taken the workgroups land
// on the reserved CUs; one that lands elsewhere (the main launch is still arriving, or the chip is not full after all) leaves at once, and so does
// every guest that finds the queue dry for ten milliseconds (callers that resubmit as soon as their batches complete leave the queue dry for 2 - 3 ms between
// two rounds: the guests stay through that) - so guests are there exactly while the chip is full AND busy, the one regime in which a
// chip without a free slot works well (profiles/r06_full_chip_with_idle_waves.txt).  The next fetch raises the yield word: they hand their chunks
// back and leave (tsx_svc_host.yield).
static void svc_try_guests_locked(tsx_service& s) {
#ifndef HIPEMU
    if (!s.cus_reserved || !g_cfg.fetch_quiet_ms || s.paused || s.rotating || !s.launched || !s.st_g) return;
    if (svc_guests_running_locked(s) || !svc_quiet(s)) return;
    const int64_t now = steady_ns();
    if (now - s.g_last_ns < 20000000) return;                            // (a launch whose workgroups all left at once is tried again 20 ms later)
    // not before the launch they are to help has arrived: its workgroups take a moment to be placed (~0.5 ms on a cold chip), and a guest that finds a
    // free slot anywhere but on a reserved CU takes it - and leaves (measured: guests launched 1 ms behind their launch, 41 of 768 stayed)
    if (now - s.launch_ns < 3000000) return;
    uint64_t outstanding = 0;
    for (const auto& m : s.out) if (!m.done && !__atomic_load_n(m.h_flag, __ATOMIC_ACQUIRE)) outstanding += m.n;
    const uint32_t gwaves = s.cus_reserved * s.waves_per_cu;
    if (outstanding <= (uint64_t)(s.grid > gwaves ? s.grid - gwaves : s.grid)) return;     // every queued chunk has a wave of the launch itself
    s.g_last_ns = now;
    __atomic_store_n(&s.h->yield, 0u, __ATOMIC_SEQ_CST);                 // word first, counter second (svc_foreground_begin)
    if (s.fg_inflight.load(std::memory_order_seq_cst) != 0) { __atomic_store_n(&s.h->yield, 1u, __ATOMIC_SEQ_CST); return; }
    tsx_svc_launch a{};
    a.sched = g_cfg.zstd_sched; a.poll_ticks = 500; a.idle_exit_ticks = g_cfg.svc_idle_exit_us * 100u;
    const uint64_t age = (uint64_t)g_cfg.svc_max_launch_ms * 100000ull;
    a.max_age_ticks_lo = (uint32_t)age; a.max_age_ticks_hi = (uint32_t)(age >> 32);
    a.guest_idle_ticks = 1000000; a.guest_launch = 1; a.guests = 1u | (g_cfg.svc_guest_looks << 1);
    a.spread_cus = 0;                                                    // (guests exist because everybody else is busy: nothing to spread)
    a.main_waves = s.grid > gwaves ? s.grid - gwaves : s.grid;
    a.launch_id = s.g_launch_id + 1;
    (void)hipGetLastError();
    tsx_launch_zstd_service(s.st_g, s.hd, s.d, gwaves, a);
    if (hipGetLastError() != hipSuccess) return;                         // (no guests this time)
    s.g_launch_id = a.launch_id; s.g_launched = true; s.guest_launches++;
#else
    (void)s;
#endif
}

static void svc_destroy(tsx_device& d) {
    if (!d.svc) return;
    tsx_service& s = *d.svc;
    if (s.h) __atomic_store_n(&s.h->stop, 1u, __ATOMIC_RELEASE);
    if (s.st) { (void)hipStreamSynchronize(s.st); (void)hipStreamDestroy(s.st); }
    if (s.st_g) { (void)hipStreamSynchronize(s.st_g); (void)hipStreamDestroy(s.st_g); }
    for (void* p : s.deferred_dev) (void)hipFree(p);
    for (void* p : s.deferred_host) (void)hipHostFree(p);
    if (s.h) (void)hipHostFree(s.h);
    if (s.h_zero) (void)hipHostFree(s.h_zero);
    if (s.d) (void)hipFree(s.d);
    d.svc.reset();
}

// The queue, the device words and - from a probe launch that covers the chip - the CU keys that exist and the ones the compressor leaves alone.
static int svc_create(tsx_device& d, int cus) {
    std::unique_ptr<tsx_service> sp(new (std::nothrow) tsx_service);
    if (!sp) return TSX_E_NOMEM;
    d.svc = std::move(sp);
    tsx_service& s = *d.svc;
    HIPCHK(hipHostMalloc((void**)&s.h, sizeof(tsx_svc_host), hipHostMallocMapped | hipHostMallocPortable));
    memset(s.h, 0, sizeof(tsx_svc_host));
    s.h->yield = 1;                                                      // (cleared when a member is published on a quiet device: svc_note_quiet_locked)
    HIPCHK(hipHostGetDevicePointer((void**)&s.hd, s.h, 0));
    HIPCHK(hipHostMalloc((void**)&s.h_zero, 64, hipHostMallocDefault));
    memset(s.h_zero, 0, 64);
    HIPCHK(hipMalloc((void**)&s.d, sizeof(tsx_svc_dev)));
    HIPCHK(hipMemset(s.d, 0, sizeof(tsx_svc_dev)));
    // The service's stream gets the LOWEST stream priority of the device.  Two reasons: the runtime keeps a pool of hardware queues per
    // priority and multiplexes a process's streams onto them - a stream that shared the service's hardware queue would sit behind a kernel
    // that lives as long as uploads go on, and nothing else in this library (or, normally, in the process) creates low-priority streams;
    // and between a compressor wave and a fetch's workgroup that could both be placed, the fetch's goes first.
    {
        int least = 0, greatest = 0;
        if (g_cfg.svc_normal_priority || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
            hipStreamCreateWithPriority(&s.st, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            s.st = nullptr;
            HIPCHK(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
        }
    }
    {   // ... and the guests' stream, of the same (lowest) priority: a hardware queue of its own in that pool (a second stream of normal priority would
        // share hardware queues with the fetch side's streams - and a kernel that lives for seconds blocks whatever is queued behind it)
        int least = 0, greatest = 0;
        if (g_cfg.svc_normal_priority || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest ||
            hipStreamCreateWithPriority(&s.st_g, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); s.st_g = nullptr; }     // (no guests then)
    }
    for (uint32_t i = TSX_SVC_MEMBERS; i-- > 0;) s.free_slots.push_back((uint16_t)i);
    // which compute units are there?  (HIP promises nothing about placement: a launch of three 48 KiB workgroups per CU that stay ~30 us each
    // has to spread over all of them; twice, in case the first one met a chip that was busy)
    std::vector<uint32_t> seen(128, 0);
    for (int pass = 0; pass < 2; pass++) {
        (void)hipGetLastError();
        tsx_launch_cu_probe(s.st, s.d, (uint32_t)cus * 3u * 2u);
        HIPCHK(hipStreamSynchronize(s.st));
        HIPCHK(hipMemcpy(seen.data(), s.d->seen, 512, hipMemcpyDeviceToHost));
        uint32_t k = 0; for (uint32_t w : seen) k += (uint32_t)__builtin_popcount(w);
        s.cu_keys = k;
        if ((int)k >= cus) break;
    }
    s.cus = (uint32_t)cus;
    // The reservation: ONE CU OF EVERY SHADER ENGINE first.  The hardware hands a kernel's workgroups to the shader engines in a fixed
    // rotation and a workgroup waits for room in ITS engine: next to waves that stay for seconds, a fetch kernel's workgroup that falls to
    // an engine without a free CU waits until the compressor launch ends (measured: kernels of a fetch each stuck for exactly one service
    // launch, with 8 reserved CUs as well as with a CU mask that kept the service off 8 CUs - profiles/r05_kernel_trace_blocked_fetch.csv.gz,
    // r05_cu_mask_variant_phases.txt).  A key's upper bits name the engine: xcc_id | se_id | sh_id (key >> 4).  Round r takes the r-th
    // highest CU of every engine: the default (0xFFFFFFFF = "one per engine") stops after round 0; a number asked for is spread the same way.
    // Never more than a quarter of the chip, and nothing at all when the probe did not find one key per CU (a key that stood for two CUs
    // would take both).
    std::vector<uint32_t> res(128, 0);
    uint32_t engines = 0;
    for (uint32_t g = 0; g < 256; g++) { bool any = false; for (uint32_t c = 0; c < 16; c++) { const uint32_t key = g << 4 | c; any |= ((seen[key >> 5] >> (key & 31)) & 1) != 0; } engines += any; }
    s.engines = engines;
    uint32_t want = g_cfg.reserved_cus == 0xFFFFFFFFu ? engines : g_cfg.reserved_cus;
    if (want > s.cus / 4) want = s.cus / 4;
    if (s.cu_keys != s.cus) {
        if (g_cfg.debug || want) fprintf(stderr, "[tsxform] device %d: %u CU keys seen for %u compute units - no CU reservation\n", d.hip_id, s.cu_keys, s.cus);
        want = 0;
    }
    uint32_t taken = 0;
    for (uint32_t round = 0; taken < want && round < 16; round++)
        for (uint32_t g = 0; g < 256 && taken < want; g++) {
            uint32_t nth = 0;
            for (int c = 15; c >= 0; c--) {
                const uint32_t key = g << 4 | (uint32_t)c;
                if (!((seen[key >> 5] >> (key & 31)) & 1)) continue;
                if (nth++ == round) { res[key >> 5] |= 1u << (key & 31); taken++; break; }
            }
        }
    s.cus_reserved = taken;
    HIPCHK(hipMemcpy(s.d->reserved, res.data(), 512, hipMemcpyHostToDevice));
    // A launch covers the chip exactly once - never more workgroups than are resident at the same time.  The waves stay for as long as there
    // is work, so workgroups that did not fit would stay PENDING for as long, and a dispatch that is still in progress holds its hardware
    // pipe: the first command of every stream whose queue sat on that pipe did not start until the launch was over.  Measured (gpurun
    // r05a-r05n, profiles/r05_service_resident_waves_and_pending_workgroups.txt): launches of 256 x 24 against the 256 x 21 that fit (the
    // kernel's 6704 bytes of LDS are allocated as 7680) kept 768 workgroups pending, and a fetch issued meanwhile came back when the launch
    // ended - up to its age limit later.
    // How many fit is MEASURED: a launch of 32 workgroups per CU whose waves stay until no workgroup has ARRIVED for 2 ms counts the most that
    // were ever resident at once (registers, LDS with its allocation granularity, scratch slots - whatever limits it; the runtime's occupancy
    // query said 24 where 21 fit).  The window was 500 us for most of round 6: the last few dozen workgroups of such a launch can arrive later
    // than that behind the others (6094 - 6133 resident "measured" on a chip that holds 6144), the division below rounded that down to 23 per
    // CU, and the service ran with 5888 waves instead of 6144 in most processes of some boxes - the "boxes that hold 23": they do not.  With
    // 2 ms every pass sees 6137 - 6144 and the best of the passes is 6144 (four processes of four).  Up to four passes, until the best is a
    // whole number of workgroups per CU twice in a row or at all after the second pass.  tsx_init runs on a device this process is not using yet.
    for (int pass = 0; pass < 4; pass++) {
        // (twice at least: the first launch of a process also loads the code object)
#ifdef HIPEMU
        tsx_svc_launch c{}; c.launch_id = ++s.launch_id; c.calibrate_ticks = 50000;       // (the CPU harness runs the workgroups one after the other: each waits its window out)
#else
        tsx_svc_launch c{}; c.launch_id = ++s.launch_id; c.calibrate_ticks = 200000;      // leave when nobody has arrived for 2 ms
#endif
        (void)hipGetLastError();
        tsx_launch_zstd_service(s.st, s.hd, s.d, s.cus * 32u, c);
        HIPCHK(hipStreamSynchronize(s.st));
        uint32_t lm[2] = {0, 0};
        HIPCHK(hipMemcpy(lm, &s.d->live, 8, hipMemcpyDeviceToHost));
        const uint32_t before = s.resident;
        if (lm[1] > s.resident) s.resident = lm[1];
        HIPCHK(hipMemcpy(&s.d->live_max, s.h_zero, 4, hipMemcpyHostToDevice));
        if (g_cfg.debug) fprintf(stderr, "[tsxform] device %d: calibration launch %d: %u workgroups resident at once\n", d.hip_id, pass, lm[1]);
        if (pass >= 1 && (lm[1] == before || s.resident % s.cus == 0)) break;
    }
    const uint32_t usable = s.cus;
    uint32_t per_cu = g_cfg.svc_waves_per_cu ? g_cfg.svc_waves_per_cu : s.resident / usable;
    if (per_cu == 0 || per_cu > 32) per_cu = 16;                        // (a measurement that cannot be: stay on the safe side)
    s.waves_per_cu = per_cu;
    s.grid = usable * per_cu;
    return TSX_OK;
}

// Frees that must not wait for the service kernel: freed at once when it is known to be gone (mu held meanwhile: no launch can begin),
// otherwise when its end is seen (svc_running_locked) or at shutdown.
static void svc_free_dev(tsx_device* dev, void* p) {
    if (!p) return;
    if (!dev->svc) { (void)hipFree(p); return; }
    std::lock_guard<std::mutex> lk(dev->svc->mu);
    if (svc_alive_locked(*dev->svc)) dev->svc->deferred_dev.push_back(p); else (void)hipFree(p);
}
static void svc_free_host(tsx_device* dev, void* p) {
    if (!p) return;
    if (!dev->svc) { (void)hipHostFree(p); return; }
    std::lock_guard<std::mutex> lk(dev->svc->mu);
    if (svc_alive_locked(*dev->svc)) dev->svc->deferred_host.push_back(p); else (void)hipHostFree(p);
}

// Memory management that needs the memory BACK (an allocation has failed): no launches until svc_resume, the running kernel is told to
// stop (its waves leave after their current chunk, <= ~1.3 s), deferred frees happen.  Members that wait meanwhile just wait.
static void svc_pause(tsx_device* dev) {
    if (!dev->svc) return;
    tsx_service& s = *dev->svc;
    std::unique_lock<std::mutex> lk(s.mu);
    s.paused++;
    __atomic_store_n(&s.h->stop, 1u, __ATOMIC_RELEASE);
    s.stop_dirty = true;
    while (svc_alive_locked(s)) { lk.unlock(); std::this_thread::sleep_for(std::chrono::microseconds(200)); lk.lock(); }
}
// The safety net of the fetch side.  One CU of every shader engine is kept free for it, and a fetch next to saturating uploads takes its
// ~2 ms - but once in a few hundred fetches (measured: 1 of 271, 2 of 12, 0 of 218 + 218 in four runs) a kernel of a fetch did not start
// until the compressor launch next to it ended, for a reason that was not found.  A batch that has waited for its own kernels for 200 ms
// while the service kernel is alive asks that launch to end: its waves leave after their current chunk (<= ~1.3 s), the waiting callers'
// watchdog starts the next one, and the fetch gets the chip in between.  Cost: one chunk time of a half-empty chip, only when it happens.
static void svc_rotate(tsx_device* dev) {
    if (!dev->svc) return;
    tsx_service& s = *dev->svc;
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.rotating || s.paused || !svc_running_locked(s)) return;
    s.rotating = true; s.rotations++;
    __atomic_store_n(&s.h->stop, 1u, __ATOMIC_RELEASE);
    s.stop_dirty = true;
}
// Waits for an event of a batch that is NOT the compressor's, with that safety net.
static hipError_t wait_event_watching(tsx_ctx* c, hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    bool asked = false;
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        (void)hipGetLastError();
        const auto age = std::chrono::steady_clock::now() - t0;
        if (!asked && age > std::chrono::milliseconds(200)) { svc_rotate(c->dev); asked = true; }
        // A single-chunk fetch is 1.6 ms.  Only the first 200 us are polled without a sleep (a piece that is nearly done); after that the thread
        // sleeps between looks - 20 us while the batch is young (a sleep's real granularity is ~60 us: + <= 4 % on a single-chunk fetch), 50 us up
        // to 20 ms, 500 us beyond.  Every ChunkCache worker (ForkJoinPool, parallelism = #cores by default) waits in here: round 5 spun for 3 ms,
        // i.e. a whole core per fetch in flight.
        if (age > std::chrono::microseconds(200))
            std::this_thread::sleep_for(std::chrono::microseconds(age > std::chrono::milliseconds(20) ? 500 : age > std::chrono::milliseconds(3) ? 50 : 20));
    }
}
static void svc_resume(tsx_device* dev) {
    if (!dev->svc) return;
    tsx_service& s = *dev->svc;
    std::lock_guard<std::mutex> lk(s.mu);
    if (s.paused && --s.paused == 0) {
        __atomic_store_n(&s.h->stop, 0u, __ATOMIC_RELEASE);
        if (!s.out.empty() && !svc_running_locked(s)) (void)svc_launch_locked(s);     // (a failure shows up in the waiting members' watchdog)
    }
}

static void device_free_consts(tsx_device& d) {
    if (d.hip_id < 0) return;
    hipSetDevice(d.hip_id);
    svc_destroy(d);
    for (auto& b : d.spare_bwork) (void)hipFree(b.first);
    d.spare_bwork.clear();
    if (d.copy_in) hipStreamDestroy(d.copy_in);
    if (d.copy_out) hipStreamDestroy(d.copy_out);
    if (d.d_crc) hipFree(d.d_crc);
    if (d.d_aes) hipFree(d.d_aes);
    if (d.d_zc) hipFree(d.d_zc);
    if (d.h_zeros) hipHostFree(d.h_zeros);
    d.d_crc = nullptr; d.d_aes = nullptr; d.d_zc = nullptr; d.h_zeros = nullptr; d.copy_in = nullptr; d.copy_out = nullptr;
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
        if (strncmp(d.arch, "gfx950", 6) != 0 && !g_cfg.allow_any_arch) {
            snprintf(g_last_err, sizeof g_last_err, "device %d is %s, this library is built for gfx950 only", d.hip_id, d.arch);
            return TSX_E_DEVICE;
        }
        HIPCHK(hipSetDevice(d.hip_id));
        {   size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) { (void)hipGetLastError(); total_b = prop.totalGlobalMem; }
            d.idle_cap = total_b / 9 * 4;
            if (g_cfg.pool_idle_bytes >= 0) d.idle_cap = (size_t)g_cfg.pool_idle_bytes;
        }
        HIPCHK(hipMalloc((void**)&d.d_crc, sizeof(tsx_crc_tables)));
        HIPCHK(hipMalloc((void**)&d.d_aes, sizeof(tsx_aes_tables)));
        HIPCHK(hipMalloc((void**)&d.d_zc, tsx_zstd_consts_bytes()));
        HIPCHK(hipHostMalloc((void**)&d.h_zeros, sizeof(tsx_gcm_key), hipHostMallocDefault));
        memset(d.h_zeros, 0, sizeof(tsx_gcm_key));
        HIPCHK(hipMemcpy(d.d_crc, hc, sizeof(tsx_crc_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_aes, ha, sizeof(tsx_aes_tables), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d.d_zc, hz, tsx_zstd_consts_bytes(), hipMemcpyHostToDevice));
        // the copy streams first, then the service's stream: whatever the runtime's stream -> hardware-queue assignment, the short copies and
        // their event markers are not the ones that end up behind the long-lived kernel
        HIPCHK(hipStreamCreateWithFlags(&d.copy_in, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&d.copy_out, hipStreamNonBlocking));
        const int rc = svc_create(d, prop.multiProcessorCount);
        if (rc) return rc;
    }
    return TSX_OK;
}

extern "C" int tsx_init_ex(int device_count, const int* device_ids, const tsx_config* cfg) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_devs.empty()) return (int)g_devs.size();
    tsx_device_scope keep;
    {   // what the process has (defaults, or what tsx_debug_config set), then the caller's structure, then the environment - a deployment's last word
        tsx_cfg c = g_cfg;
        if (cfg && cfg->struct_size >= offsetof(tsx_config, fetch_quiet_ms)) {      // (the struct as it was before fetch_quiet_ms is accepted too)
            if (cfg->fetch_reserved_cus != TSX_CFG_DEFAULT) c.reserved_cus = cfg->fetch_reserved_cus > 128 ? 128 : cfg->fetch_reserved_cus;
            if (cfg->service_max_launch_ms != TSX_CFG_DEFAULT) c.svc_max_launch_ms = cfg->service_max_launch_ms;
            if (cfg->fetch_shared_cu_waves != TSX_CFG_DEFAULT) c.svc_keep_waves = cfg->fetch_shared_cu_waves > 8 ? 8 : cfg->fetch_shared_cu_waves;
            if (cfg->struct_size >= offsetof(tsx_config, fetch_quiet_ms) + 4 && cfg->fetch_quiet_ms != TSX_CFG_DEFAULT) c.fetch_quiet_ms = cfg->fetch_quiet_ms;
            if (cfg->pool_idle_bytes != TSX_CFG_DEFAULT64) c.pool_idle_bytes = (long long)cfg->pool_idle_bytes;
        } else if (cfg) return TSX_E_INVAL;
        cfg_from_env(c);
        g_cfg = c;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        snprintf(g_last_err, sizeof g_last_err, "no HIP device visible");
        return TSX_E_DEVICE;
    }
    int want = device_count <= 0 ? visible : device_count;
    if (device_ids) { for (int i = 0; i < want; i++) if (device_ids[i] < 0 || device_ids[i] >= visible) return TSX_E_INVAL; }
    else if (want > visible) return TSX_E_INVAL;
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
             "tsxform 0.5 (gfx950 HIP; CRC32C, AES-256-GCM, Zstd level-3 frames; zstd parity target libzstd 1.5.7 / 1.5.6 profile; %d device(s): %s; "
             "compressor service: %u waves, %u of %u CUs reserved for fetches)",
             (int)g_devs.size(), g_devs[0].name, g_devs[0].svc->grid, g_devs[0].svc->cus_reserved, g_devs[0].svc->cus);
    g_version.store(vb, std::memory_order_release);
    return (int)g_devs.size();
}

extern "C" int tsx_init(int device_count, const int* device_ids) { return tsx_init_ex(device_count, device_ids, nullptr); }

extern "C" int tsx_device_count(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_devs.size();
}

static void ctx_free_device_mem(tsx_ctx* c) {
    hipSetDevice(c->dev->hip_id);
    // the key schedule and the raw key never outlive the context in readable form (copies of zeros, not kernels)
    if (c->st && c->d_key) { hipMemcpyAsync(c->d_key, c->dev->h_zeros, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, c->st); }
    if (c->st && c->d_keyraw) { hipMemcpyAsync(c->d_keyraw, c->dev->h_zeros, 128, hipMemcpyHostToDevice, c->st); }
    if (c->st) hipStreamSynchronize(c->st);
    void* ptrs[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen, c->d_partials, c->d_key, c->d_keyraw, c->d_in, c->d_out, c->d_mid, c->d_zwork, c->d_bwork, c->d_segdone};
    for (void* p : ptrs) svc_free_dev(c->dev, p);
    svc_free_host(c->dev, c->h_segflag);
    svc_free_host(c->dev, c->h_descs);
    if (c->h_keyraw) { memset(c->h_keyraw, 0, 128); svc_free_host(c->dev, c->h_keyraw); }
    if (c->h_key) { memset(c->h_key, 0, sizeof(tsx_gcm_key)); svc_free_host(c->dev, c->h_key); }
    for (auto& e : c->ev) if (e) hipEventDestroy(e);
    for (auto& row : c->sub_ev) for (auto& e : row) if (e) hipEventDestroy(e);
    if (c->st) hipStreamDestroy(c->st);
    if (c->st_in) hipStreamDestroy(c->st_in);
    if (c->st_out) hipStreamDestroy(c->st_out);
    if (c->st_out2) hipStreamDestroy(c->st_out2);
    for (auto& q : c->st_pc) if (q) hipStreamDestroy(q);
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
    { std::lock_guard<std::mutex> lr(g_reg_mu); g_registered.clear(); }
    g_version.store(kUninitVersion, std::memory_order_release);
}

template <class T>
static int grow(tsx_device* dev, T** p, size_t* cap, size_t need) {
    if (need <= *cap && *p) return TSX_OK;
    if (*p) { svc_free_dev(dev, *p); *p = nullptr; *cap = 0; }
    size_t want = need + need / 8 + 256;
    hipError_t e = hipMalloc((void**)p, want * sizeof(T));
    if (e != hipSuccess) { tsx_set_err("hipMalloc(workspace)", e); return TSX_E_NOMEM; }
    *cap = want;
    return TSX_OK;
}

// Small detransform batches (a fetch: one chunk, a prefetch window) decode one workgroup per BLOCK instead of per chunk: the
// chunk-serial decoder needs 25-50 ms for a chunk however idle the chip is.  Up to 256 chunks (measured: 27.6 ms at 256 chunks against
// the chunk form's 33, profiles/r03_dec_latency_block_form.jsonl); the test hook dec_block_chunks moves the limit (0 = never).
static bool dec_use_blocks(uint32_t n, uint32_t max_out) { return n <= g_cfg.dec_block_chunks && tsx_zstd_blockmode_takes(max_out); }

// max_out: largest output slot of the batch (detransform: the CRC of the restored bytes runs over dst_cap-sized slots)
static int ctx_reserve(tsx_ctx* c, uint32_t n, uint32_t max_len, uint32_t max_out, uint32_t flags, bool host_mem, size_t in_bytes, size_t out_bytes) {
    HIPCHK(hipSetDevice(c->dev->hip_id));
    if (n > c->descs_cap || !c->d_descs) {
        void* olds[] = {c->d_descs, c->d_gchunks, c->d_status, c->d_zlen};
        for (void* p : olds) svc_free_dev(c->dev, p);
        svc_free_host(c->dev, c->h_descs);
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
    int rc = grow(c->dev, &c->d_partials, &c->partials_cap, (size_t)n * per_chunk);
    if (rc) return rc;
    c->partials_per_chunk = per_chunk;
    if (host_mem) {
        if ((rc = grow(c->dev, &c->d_in, &c->in_cap, in_bytes + 64))) return rc;
        if ((rc = grow(c->dev, &c->d_out, &c->out_cap, out_bytes + 64))) return rc;
    }
    if (flags & TSX_COMPRESS) {
        size_t stride = (tsx_transformed_bound(max_len, TSX_COMPRESS) + 63) & ~(size_t)63;
        if ((rc = grow(c->dev, &c->d_mid, &c->mid_cap, stride * n))) return rc;
        c->mid_stride = stride;
        size_t zw = tsx_zstd_workspace_bytes(n, max_len);
        uint8_t* zp = (uint8_t*)c->d_zwork;
        rc = grow(c->dev, &zp, &c->zwork_cap, zw);
        c->d_zwork = zp;                                     // also when grow failed: it has freed the old block
        if (rc) return rc;
        if (max_out && dec_use_blocks(n, max_out)) {         // inverse chain, small batch
            const size_t need = tsx_zstd_blockmode_bytes(n, max_out);
            if (!c->d_bwork || c->bwork_cap < need) {
                // a workspace another context left behind (pool_release) before a fresh allocation
                std::lock_guard<std::mutex> lk(g_mu);
                auto& sp = c->dev->spare_bwork;
                for (size_t k = 0; k < sp.size(); k++) if (sp[k].second >= need) {
                    if (c->d_bwork) sp.push_back({c->d_bwork, c->bwork_cap});
                    c->d_bwork = sp[k].first; c->bwork_cap = sp[k].second;
                    sp.erase(sp.begin() + (long)k);
                    break;
                }
            }
            uint8_t* bp = (uint8_t*)c->d_bwork;
            rc = grow(c->dev, &bp, &c->bwork_cap, need);
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
    HIPCHK(hipMalloc((void**)&c->d_segdone, 64));
    // Nothing an allocator left behind: what tsx_debug_key_residue reads, and the completion counters, start from zero.  Copies of pinned
    // zeros on the context's OWN stream - it is non-blocking (a memset on the null stream could land behind the first batch's key copy: seen
    // on the device), and a memset would be a kernel, which waits for a wave slot on a chip full of compressor waves.
    HIPCHK(hipMemcpyAsync(c->d_key, c->dev->h_zeros, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(c->d_keyraw, c->dev->h_zeros, 128, hipMemcpyHostToDevice, c->st));
    HIPCHK(hipMemcpyAsync(c->d_segdone, c->dev->h_zeros, 64, hipMemcpyHostToDevice, c->st));
    HIPCHK(hipHostMalloc((void**)&c->h_keyraw, 128, hipHostMallocDefault));
    HIPCHK(hipHostMalloc((void**)&c->h_key, sizeof(tsx_gcm_key), hipHostMallocMapped | hipHostMallocPortable));
    HIPCHK(hipHostGetDevicePointer((void**)&c->hd_key, c->h_key, 0));
    HIPCHK(hipHostMalloc((void**)&c->h_segflag, 64, hipHostMallocMapped | hipHostMallocPortable));
    HIPCHK(hipHostGetDevicePointer((void**)&c->hd_segflag, c->h_segflag, 0));
    memset(c->h_segflag, 0, 64);
    HIPCHK(hipStreamSynchronize(c->st));                                // (the counters are zero before a wave of the service can touch them)
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
    // a new context zeroes a few device words with small copies (blit kernels of the runtime): like a fetch, it asks guest waves for room
    svc_foreground_begin(dev);
    int rc = ctx_init_device_objects(c);
    svc_foreground_end(dev, false);
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
// Idle pooled contexts of a device give their memory back (an allocation has just failed: what is cached must not be the reason).  The
// service is paused meanwhile: a free only happens with its kernel gone.
static bool pool_drain(tsx_device* dev) {
    std::vector<tsx_ctx*> dead;
    std::vector<std::pair<void*, size_t>> spare;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        dead.swap(dev->idle); dev->idle_bytes = 0;
        spare.swap(dev->spare_bwork);
    }
    if (dead.empty() && spare.empty()) return false;
    tsx_device_scope keep;
    svc_pause(dev);
    for (tsx_ctx* c : dead) { ctx_free_device_mem(c); delete c; }
    for (auto& b : spare) { hipSetDevice(dev->hip_id); (void)hipFree(b.first); }
    svc_resume(dev);
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
    void* surplus = nullptr;
    tsx_device* const dev = c->dev;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        tsx_device& d = *c->dev;
        d.in_use--;
        if (c->d_bwork) {
            // the fetch side's ForkJoinPool issues from dozens of threads (ChunkCache.java:140): without a bound every idle context would
            // park a block-form workspace.  Beyond a few, the workspace leaves its context for the device's spare list - the next small fetch
            // on a context without one takes it from there (ctx_reserve).  Never a hipFree here: that call waits for every stream of the
            // device, i.e. for second-long compressor waves, on the path of a fetch.
            uint32_t with = 0;
            for (const tsx_ctx* o : d.idle) if (o->d_bwork) with++;
            if (with >= TSX_POOL_MAX_IDLE_BWORK) {
                if (d.spare_bwork.size() < TSX_POOL_MAX_IDLE_BWORK) d.spare_bwork.push_back({c->d_bwork, c->bwork_cap});
                else surplus = c->d_bwork;                               // freed when the service kernel is gone (svc_free_dev), not here
                c->d_bwork = nullptr; c->bwork_cap = 0;
            }
        }
        const size_t b = ctx_workspace_bytes(c);
        if (d.idle.size() < TSX_POOL_MAX_IDLE && (d.idle.empty() || d.idle_bytes + b <= d.idle_cap)) { d.idle.push_back(c); d.idle_bytes += b; c = nullptr; }
    }
    tsx_device_scope keep;
    if (surplus) { hipSetDevice(dev->hip_id); svc_free_dev(dev, surplus); }
    if (!c) return;
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

// First kernel of a batch of ordinary kernels (fetches above all).  A copy of 48 bytes, or of the 21 KB key schedule, is a blit KERNEL of
// the runtime (__amd_rocclr_copyBuffer, 512-thread workgroups): every fetch used to queue five of them (descriptors up, key schedule up,
// descriptors down, two wipes), and they were the kernels that got stuck next to the compressor service (all six stuck dispatches of
// profiles/r05_kernel_trace_blocked_fetch.csv.gz are copyBuffer kernels; next to guest waves such a copy waits for the launch to end).
// So nothing small is copied any more: this kernel reads the batch's descriptors - and, for the first piece of an encrypted batch, the key
// schedule the host built - straight from the context's pinned memory into their device-side places and starts every status from
// TSX_OK; publish_status_kernel / crc32c_final_kernel write the results back there; wipe_key_kernel clears the key material.
__global__ __launch_bounds__(256) void begin_batch_kernel(const tsx_chunk_desc* __restrict__ hd_descs, tsx_chunk_desc* __restrict__ descs, int32_t* __restrict__ status,
                                                          uint32_t n, const uint4* __restrict__ hd_key, uint4* __restrict__ d_key, uint32_t key_words16) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint4* s = reinterpret_cast<const uint4*>(hd_descs + i);
        uint4* d = reinterpret_cast<uint4*>(descs + i);
        const uint4 a = s[0], b = s[1], c = s[2];
        d[0] = a; d[1] = b; d[2] = c;
        status[i] = TSX_OK;
    }
    if (hd_key) for (uint32_t k = i; k < key_words16; k += gridDim.x * blockDim.x) d_key[k] = hd_key[k];
}
static_assert(sizeof(tsx_chunk_desc) == 48 && sizeof(tsx_gcm_key) % 16 == 0, "begin_batch_kernel moves descriptors and the key schedule as 16-byte words");

// Last kernel of an encrypted batch: the key schedule and the raw key bytes do not stay behind in the context
__global__ __launch_bounds__(256) void wipe_key_kernel(uint4* __restrict__ d_key, uint32_t key_words16, uint4* __restrict__ d_keyraw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 z; z.x = z.y = z.z = z.w = 0;
    for (uint32_t k = i; k < key_words16; k += gridDim.x * blockDim.x) d_key[k] = z;
    if (i < 8) d_keyraw[i] = z;
}

__global__ void init_status_kernel(int32_t* status, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[i] = TSX_OK;
}

// mirror != nullptr: the whole descriptor as it stands goes to the batch's pinned copy as well (what the caller gets back; a CRC stage that
// runs afterwards adds its word there itself)
__global__ void publish_status_kernel(tsx_chunk_desc* descs, const int32_t* status, uint32_t n, tsx_chunk_desc* mirror) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    descs[i].status = status[i];
    if (status[i] != TSX_OK) descs[i].dst_len = 0;
    if (mirror) {
        const uint4* s = reinterpret_cast<const uint4*>(descs + i);
        uint4* d = reinterpret_cast<uint4*>(mirror + i);
        const uint4 a = s[0], b = s[1], c = s[2];
        d[0] = a; d[1] = b; d[2] = c;
    }
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

// test hook `trace`: where a batch spends its time, as seen from the calling thread (microseconds since the first mark of the call)
struct tsx_trace {
    std::chrono::steady_clock::time_point t0; bool on;
    tsx_trace() : t0(std::chrono::steady_clock::now()), on(g_cfg.trace) {}
    void mark(const char* what) const { if (on) fprintf(stderr, "[tsx trace %p] %9.0f us  %s\n", (const void*)this, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what); }
};

static float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; hipEventElapsedTime(&ms, a, b); return ms; }

struct tsx_sub { uint32_t lo, n; size_t in_lo, in_hi; };     // chunks [lo, lo + n), their input bytes [in_lo, in_hi) of src

struct tsx_run {                                              // what one batch needs everywhere below
    tsx_ctx* c; const tsx_batch_params* params; tsx_chunk_desc* descs; uint32_t n; const void* src; void* dst; size_t src_size, dst_size;
    int mem_kind, mode; uint32_t flags, max_len, max_out; bool host, packed, enc, comp, fuse_stages, pooled;
    const uint8_t* d_src; uint8_t* d_dst;
};

// ---- service: members ---------------------------------------------------------------------------------------------------------------
// Publish one member: `proto` names its buffers (n, done and flag included); its n chunks become the next n tickets.  Returns the member's
// id (for svc_retire) through *id.  Blocks while the ticket ring or the member slots are full (members retire one by one).
static int svc_submit(tsx_device* dev, const tsx_zseg& proto, const uint32_t* h_flag, uint64_t* id) {
    tsx_service& s = *dev->svc;
    const uint32_t n = proto.n;
    if (!n || n > TSX_SVC_MEMBER_MAX) return TSX_E_INVAL;
    std::unique_lock<std::mutex> lk(s.mu);
    // room: a ticket record is reused TSX_SVC_TICKETS tickets later - by then every member up to it must be complete ON THE DEVICE (its flag
    // raised; whether its caller has come back for it yet does not matter: a caller that publishes its pieces one after the other must
    // never wait here for a piece of its own that only it can retire)
    for (;;) {
        uint32_t oldest = s.published; bool any = false;
        for (const auto& m : s.out) if (!m.done && !__atomic_load_n(m.h_flag, __ATOMIC_ACQUIRE)) { oldest = m.first; any = true; break; }
        if (!s.free_slots.empty() && (!any || (uint32_t)(s.published + n - oldest) <= TSX_SVC_TICKETS)) break;
        // whoever waits for room is also the service's watchdog (as svc_wait is): the kernel may have ended - idle, age limit, a rotation -
        // with the members that hold the room still unserved, and their callers may all be in here
        if (!s.paused && !s.out.empty() && !svc_running_locked(s)) { s.watchdog_launches++; (void)svc_launch_locked(s); }
        s.cv.wait_for(lk, std::chrono::milliseconds(1));
    }
    svc_note_quiet_locked(s);
    const uint16_t slot = s.free_slots.back(); s.free_slots.pop_back();
    const uint16_t gen = ++s.slot_gen[slot];
    tsx_zseg e = proto;
    e.gen = gen; e.pad = 0;
    s.h->member[slot] = e;
    const uint32_t first = s.published;
    for (uint32_t i = 0; i < n; i++) { tsx_svc_ticket& t = s.h->ticket[(first + i) & (TSX_SVC_TICKETS - 1)]; t.member_gen = (uint32_t)gen << 16 | slot; t.chunk = i; }
    s.published = first + n;
    __atomic_store_n(&s.h->published, s.published, __ATOMIC_RELEASE);   // the waves' poll picks it up (a few microseconds)
    *id = s.next_id++;
    s.out.push_back({*id, first, n, slot, false, h_flag});
    s.members++;
    if (!svc_running_locked(s)) { const int rc = svc_launch_locked(s); if (rc == TSX_OK) svc_try_guests_locked(s); return rc; }     // (on failure the caller abandons the member: svc_retire)
    svc_try_guests_locked(s);                                           // the queue has just grown: deeper than the launch has waves?  (and quiet?)
    return TSX_OK;
}

// The member is over - completed, or abandoned (`abandon`: its tickets, should a later launch ever reach them, name a stale generation).
static void svc_retire(tsx_device* dev, uint64_t id, bool abandon) {
    tsx_service& s = *dev->svc;
    std::lock_guard<std::mutex> lk(s.mu);
    for (auto& m : s.out) if (m.id == id) {
        m.done = true;
        if (abandon) { s.slot_gen[m.slot]++; __atomic_store_n(&s.h->member[m.slot].gen, (uint32_t)s.slot_gen[m.slot], __ATOMIC_RELEASE); }
        else s.chunks += m.n;
        break;
    }
    while (!s.out.empty() && s.out.front().done) { s.free_slots.push_back(s.out.front().slot); s.out.pop_front(); }
    s.cv.notify_all();
}

// Wait for a member's flag.  A chunk takes about a second: short sleeps only while the member is young (tests, tiny chunks), a
// millisecond between looks afterwards - dozens of callers must not burn the cores next to the GPU's NUMA node.  Every look that follows
// a real sleep is also the service's watchdog: a kernel that has ended with this member unfinished is started again.
static int svc_wait(tsx_device* dev, uint32_t* h_flag) {
    tsx_service& s = *dev->svc;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t failures = 0;
    for (uint32_t look = 0;; look++) {
        if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE)) return TSX_OK;
        const auto age = std::chrono::steady_clock::now() - t0;
        const bool young = age < std::chrono::milliseconds(2);
        if (!young || (look & 7) == 7) {
            std::lock_guard<std::mutex> lk(s.mu);
            if (!s.paused && !svc_running_locked(s)) {
                if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE)) return TSX_OK;
                s.watchdog_launches++;
                if (svc_launch_locked(s) != TSX_OK && ++failures >= 3) return TSX_E_DEVICE;
            } else if (!young) svc_try_guests_locked(s);                // (the callers may all be in here: whoever waits also asks whether guests are due)
        }
        std::this_thread::sleep_for(young ? std::chrono::microseconds(20) : age < std::chrono::milliseconds(50) ? std::chrono::microseconds(250) : std::chrono::microseconds(1000));
    }
}

// Returns when the device's service kernel has ended (its waves leave a moment after the last chunk): measurement tools bracket a timed
// region with it so that the region's chunks and the kernel launches that did them can be set against each other (tsx_service_stats).
extern "C" int tsx_service_quiesce(int device_index) {
    tsx_device* dev;
    { std::lock_guard<std::mutex> lk(g_mu); if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL; dev = &g_devs[device_index]; }
    tsx_service& s = *dev->svc;
    for (;;) {
        { std::lock_guard<std::mutex> lk(s.mu); if (!svc_alive_locked(s)) return TSX_OK; }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

static_assert(sizeof(tsx_service_info) == 112, "relocated_waves took the struct's tail padding: callers compiled before it hand in 112 bytes too");
extern "C" int tsx_service_stats(int device_index, tsx_service_info* out) {
    if (!out) return TSX_E_INVAL;
    tsx_device* dev;
    { std::lock_guard<std::mutex> lk(g_mu); if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL; dev = &g_devs[device_index]; }
    tsx_service& s = *dev->svc;
    tsx_device_scope keep;
    if (hipSetDevice(dev->hip_id) != hipSuccess) return TSX_E_DEVICE;
    std::lock_guard<std::mutex> lk(s.mu);
    const bool running = svc_running_locked(s);
    memset(out, 0, sizeof *out);
    out->launches = s.launches; out->watchdog_launches = s.watchdog_launches; out->rotations = (uint32_t)s.rotations; out->members = s.members; out->chunks = s.chunks;
    out->kernel_ms = s.kernel_ms; out->running = running ? 1u : 0u;
    out->waves = s.grid; out->compute_units = s.cus; out->cu_keys_seen = s.cu_keys; out->reserved_cus = s.cus_reserved; out->shader_engines = s.engines;
    out->guest_launches = (uint32_t)s.guest_launches; out->readmissions = (uint32_t)s.readmissions;
    if (running) {
        // the launch is alive: the words its waves mirror into pinned memory (a copy out of device memory is a blit kernel for sizes like
        // these and, next to guest waves, waits for the launch to end - tsx_internal.h, tsx_svc_host.m_*)
        out->device_chunks = __atomic_load_n(&s.h->m_chunks, __ATOMIC_RELAXED); out->live_waves = __atomic_load_n(&s.h->m_live, __ATOMIC_RELAXED);
        out->live_waves_max = __atomic_load_n(&s.h->m_live_max, __ATOMIC_RELAXED); out->wave_starts = __atomic_load_n(&s.h->m_wave_starts, __ATOMIC_RELAXED);
        out->reserved_exits = __atomic_load_n(&s.h->m_reserved_exits, __ATOMIC_RELAXED); out->skipped_tickets = __atomic_load_n(&s.h->m_skipped, __ATOMIC_RELAXED);
        out->yielded_waves = __atomic_load_n(&s.h->m_yields, __ATOMIC_RELAXED); out->returned_chunks = __atomic_load_n(&s.h->m_returned, __ATOMIC_RELAXED);
        out->relocated_waves = __atomic_load_n(&s.h->m_relocated, __ATOMIC_RELAXED);
        return TSX_OK;
    }
    uint32_t w[4] = {0, 0, 0, 0};
    if (hipMemcpy(w, &s.d->stat_yields, 8, hipMemcpyDeviceToHost) == hipSuccess) { out->yielded_waves = w[0]; out->returned_chunks = w[1]; } else (void)hipGetLastError();
    if (hipMemcpy(w, &s.d->stat_chunks, sizeof w, hipMemcpyDeviceToHost) == hipSuccess) {
        out->device_chunks = w[0]; out->wave_starts = w[1]; out->reserved_exits = w[2]; out->skipped_tickets = w[3];
    } else (void)hipGetLastError();
    if (hipMemcpy(w, &s.d->live, 8, hipMemcpyDeviceToHost) == hipSuccess) { out->live_waves = w[0]; out->live_waves_max = w[1]; } else (void)hipGetLastError();
    if (hipMemcpy(w, &s.d->stat_relocated, 4, hipMemcpyDeviceToHost) == hipSuccess) out->relocated_waves = w[0]; else (void)hipGetLastError();
    return TSX_OK;
}

// Test hook (not part of the ABI): put the device's ticket counters at `published` (an idle service only) - the wrap-around of the 32-bit
// counters is ten days of full-rate compression away otherwise.
extern "C" int tsx_debug_service_seed(int device_index, uint32_t published) {
    tsx_device* dev;
    { std::lock_guard<std::mutex> lk(g_mu); if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL; dev = &g_devs[device_index]; }
    tsx_service& s = *dev->svc;
    tsx_device_scope keep;
    if (hipSetDevice(dev->hip_id) != hipSuccess) return TSX_E_DEVICE;
    std::lock_guard<std::mutex> lk(s.mu);
    if (svc_alive_locked(s) || !s.out.empty()) return TSX_E_INVAL;
    const uint32_t w[2] = {published, published};
    if (hipMemcpy(&s.d->next, w, 8, hipMemcpyHostToDevice) != hipSuccess) return TSX_E_DEVICE;      // next, pub
    const uint32_t fa[2] = {published, 0};
    static_assert(offsetof(tsx_svc_dev, avail) == offsetof(tsx_svc_dev, fin) + 4, "fin, avail");
    if (hipMemcpy(&s.d->fin, fa, 8, hipMemcpyHostToDevice) != hipSuccess) return TSX_E_DEVICE;      // fin, avail (nothing outstanding: no right to a ticket)
    s.published = published;
    __atomic_store_n(&s.h->published, published, __ATOMIC_RELEASE);
    return TSX_OK;
}

// Can the device address ALL of [p, p + bytes)?  Zero-copy output lets the compressor waves write there; a buffer of which only a
// prefix is registered, or that spans two registrations, would fault the process (the broker's JVM) at the first byte behind the
// mapping.  Known extents only: a range inside ONE tsx_host_register'ed buffer, or inside one allocation the runtime reports
// (hipHostMalloc'ed memory); anything else takes the copy path.
static uint8_t* device_alias_of_range(void* p, size_t bytes) {
    if (!p || !bytes) return nullptr;
    bool inside = false;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (const auto& r : g_registered) if ((uintptr_t)p >= r.first && (uintptr_t)p - r.first <= r.second && bytes <= r.second - ((uintptr_t)p - r.first)) { inside = true; break; }
    }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || !dp) { (void)hipGetLastError(); return nullptr; }      // pageable memory
    if (inside) return (uint8_t*)dp;
#ifndef HIPEMU
    void* base = nullptr; size_t size = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)dp) == hipSuccess && base && (uintptr_t)dp >= (uintptr_t)base &&
        bytes <= size - ((uintptr_t)dp - (uintptr_t)base)) return (uint8_t*)dp;
    (void)hipGetLastError();
#endif
    return nullptr;
}

static int copy_back(const tsx_run& r, const tsx_sub& sb, size_t* packed_at, bool* packed_full, hipStream_t out_st);

// ---- the compressing forward chain: members of the device's service ---------------------------------------------------------------------
// Any compressing batch - explicit context or pooled, device or host memory - takes this way.  The batch is cut into at most comp_pieces
// members when its input comes from host memory (piece k + 1's input copy overlaps piece k's waves; a member is published when ITS input
// has landed: what is in the queue is runnable), otherwise it is one member (several beyond TSX_SVC_MEMBER_MAX chunks).  The waves run the
// whole chain of a chunk and own its descriptor in the context's pinned mirror: nothing but tickets goes to the device - no descriptor
// upload, no status kernels, no key upload (a small copy or kernel queued next to second-long waves waits for them: measured in rounds 3-4).
static int run_compress(tsx_run& r) {
    tsx_ctx* c = r.c;
    tsx_device* dev = c->dev;
    const uint32_t n = r.n;
    uint32_t max_len, max_out; size_t in_bytes; bool monotonic;
    int rc = validate(r.descs, n, r.src_size, r.dst_size, !r.packed, &max_len, &max_out, &in_bytes, &monotonic);
    if (rc) return rc;
    size_t out_bytes = r.dst_size;
    if (r.packed) {
        // the waves still write one bound-sized slot per chunk; only the bytes produced end up, back to back, in the caller's buffer
        const size_t slot = (tsx_transformed_bound(max_len, r.flags) + 63) & ~(size_t)63;
        if (slot >= ((size_t)1 << 32)) return TSX_E_INVAL;
        for (uint32_t i = 0; i < n; i++) { r.descs[i].dst_off = (uint64_t)i * slot; r.descs[i].dst_cap = (uint32_t)slot; }
        out_bytes = (size_t)n * slot; max_out = (uint32_t)slot;
    }
    r.max_len = max_len; r.max_out = max_out;
    // Zero-copy output (round 4).  The wave that finishes a chunk writes its bytes straight into the caller's buffer over PCIe when the device
    // can address ALL of it (device_alias_of_range): posted writes of a few ms of a second-long wave, released to system scope before the
    // chunk is counted done.  No device output buffer, no copy-out phase: with 32-48 callers a segment's 256 output copies stood 0.3-1.0 s
    // in the copy engine's queue behind the other callers' (profiles/r04_broker_shape_experiments.txt).  Slot layout (TSX_MEM_HOST, what
    // GpuTransformChunkEnumeration.java issues): nothing is left to do on the host.  Packed layout: pooled contexts (256-chunk segments) let
    // the waves fill bound-sized slots in the caller's buffer when it has room for them and pack them down in place; an explicit context's
    // 2048-chunk batch would spend ~90 ms of one host thread on that behind the kernel - it keeps the copy path, which packs piece by piece
    // while later pieces run (test hook zero_copy_packed takes it anyway).
    uint8_t* zc_dst = nullptr;
    if (r.host && r.fuse_stages && !g_cfg.no_zero_copy_out && (!r.packed || (r.dst_size >= out_bytes && (r.pooled || g_cfg.zero_copy_packed))))
        zc_dst = device_alias_of_range(r.dst, r.packed ? out_bytes : r.dst_size);
    rc = reserve_or_drain(c, n, max_len, 0, r.flags, r.host, in_bytes, zc_dst ? 0 : out_bytes);
    if (rc) return rc;
    r.d_src = r.host ? c->d_in : (const uint8_t*)r.src;
    r.d_dst = zc_dst ? zc_dst : r.host ? c->d_out : (uint8_t*)r.dst;
    memset(&c->timing, 0, sizeof c->timing);
    tsx_timing& t = c->timing;
    c->last_zero_copy = zc_dst != nullptr;
    // ---- the pieces ----
    std::vector<tsx_sub> subs;
    {
        uint32_t pieces = 1;
        if (r.host && monotonic && !g_cfg.no_pipeline) { pieces = g_cfg.comp_pieces < 1 ? 1 : g_cfg.comp_pieces > TSX_COMP_PIECES_MAX ? TSX_COMP_PIECES_MAX : g_cfg.comp_pieces; if (n < 8 * pieces && !g_cfg.sub_bytes) pieces = n >= 16 ? 2 : 1; }
        uint32_t per = (n + pieces - 1) / pieces;
        if (per > TSX_SVC_MEMBER_MAX) per = TSX_SVC_MEMBER_MAX;
        if ((n + per - 1) / per > TSX_COMP_PIECES_MAX) return TSX_E_INVAL;                // (> 131072 chunks in one call)
        for (uint32_t lo = 0; lo < n; lo += per) {
            const uint32_t cnt = n - lo < per ? n - lo : per;
            size_t a = r.descs[lo].src_off, b = a;
            if (monotonic) b = (size_t)(r.descs[lo + cnt - 1].src_off + r.descs[lo + cnt - 1].src_len);
            else { a = 0; b = in_bytes; }
            subs.push_back({lo, cnt, a, b});
        }
    }
    const size_t ns = subs.size();
    c->last_members = (uint32_t)ns;
    for (size_t k = 1; k < ns; k++) for (auto& e : c->sub_ev[k]) if (!e) HIPCHK(hipEventCreate(&e));
    // ---- descriptors and key schedule, where the waves read and write them: the context's pinned memory ----
    memcpy(c->h_descs, r.descs, (size_t)n * sizeof(tsx_chunk_desc));
    const bool self_status = r.fuse_stages;
    if (self_status) for (uint32_t i = 0; i < n; i++) { c->h_descs[i].status = TSX_E_DEVICE; c->h_descs[i].dst_len = 0; }   // the waves own them from here: a
                                                                        // stale TSX_OK the caller handed in must not survive a chunk that never ran
    if (r.enc && r.fuse_stages) tsx_gcm_key_build_host(r.params->key, r.params->aad, r.params->aad_len, c->h_key);
    hipStream_t cin = r.pooled ? dev->copy_in : c->st_in, cout_ = r.pooled ? dev->copy_out : c->st_out;
    const auto t_begin = std::chrono::steady_clock::now();
    uint64_t ids[TSX_COMP_PIECES_MAX] = {0};
    size_t submitted = 0;
    auto abandon_all = [&](int code) {
        // nothing of this call may still be running when it returns with an error: members that were published are taken back (their
        // tickets become stale) only once the service kernel is gone; copies that were queued (they read the caller's source) have drained
        if (submitted) { svc_pause(dev); for (size_t k = 0; k < submitted; k++) if (ids[k]) svc_retire(dev, ids[k], true); svc_resume(dev);
                         (void)hipMemcpy(c->d_segdone, dev->h_zeros, 64, hipMemcpyHostToDevice); memset(c->h_segflag, 0, 64); }
        (void)hipGetLastError();
        if (r.host) { (void)hipStreamSynchronize(cin); (void)hipStreamSynchronize(cout_); }
        (void)hipStreamSynchronize(c->st);
        return code;
    };
    // every HIP failure from the first queued copy on leaves through abandon_all (a bare return would leave copies of the caller's buffer,
    // or published members that read a key about to be wiped, in flight)
#define HIPCHK_AB(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { tsx_set_err(#x, e_); return abandon_all(TSX_E_DEVICE); } } while (0)
    if (!self_status) {
        // test hook stages_separate: CRC, compressor, GCM (or the copy into the slots) as separate launches around the service's members
        HIPCHK_AB(hipMemcpyAsync(c->d_descs, c->h_descs, (size_t)n * sizeof(tsx_chunk_desc), hipMemcpyHostToDevice, c->st));
        hipLaunchKernelGGL(init_status_kernel, dim3((n + 255) / 256), dim3(256), 0, c->st, c->d_status, n);
    }
    // ---- input copies, all queued at once; every piece is published when its own bytes are on the device ----
    if (r.host) {
        HIPCHK_AB(hipEventRecord(c->ev[2], cin));
        for (size_t k = 0; k < ns; k++) {
            const tsx_sub& sb = subs[k];
            if ((k == 0 || monotonic) && sb.in_hi > sb.in_lo) HIPCHK_AB(hipMemcpyAsync(c->d_in + sb.in_lo, (const uint8_t*)r.src + sb.in_lo, sb.in_hi - sb.in_lo, hipMemcpyHostToDevice, cin));
            HIPCHK_AB(hipEventRecord(c->sub_ev[k][5], cin));
        }
    }
    for (size_t k = 0; k < ns; k++) {
        const tsx_sub& sb = subs[k];
        if (r.host && hipEventSynchronize(c->sub_ev[k][5]) != hipSuccess) return abandon_all(TSX_E_DEVICE);
        if (k == 0) t.h2d_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        if (!self_status) {
            if ((r.flags & TSX_CRC)) { tsx_launch_crc32c(c->st, dev->d_crc, r.d_src, c->d_descs + sb.lo, sb.n, r.max_len, c->d_partials + (size_t)sb.lo * c->partials_per_chunk, 0); t.crc_launches += 2; }
            if (hipStreamSynchronize(c->st) != hipSuccess) return abandon_all(TSX_E_DEVICE);
        }
        tsx_zseg m; memset(&m, 0, sizeof m);
        m.n = sb.n; m.profile = r.params->zstd_profile;
        m.src_base = r.d_src; m.descs = (self_status ? c->hd_descs : c->d_descs) + sb.lo; m.mid = c->d_mid + (size_t)sb.lo * c->mid_stride; m.mid_stride = c->mid_stride;
        m.zlen = c->d_zlen + sb.lo; m.status = c->d_status + sb.lo; m.work = (uint8_t*)c->d_zwork + (size_t)sb.lo * tsx_zstd_workspace_bytes(1, 0);
        if (self_status) {
            m.fuse.crc = (r.flags & TSX_CRC) ? dev->d_crc : nullptr;
            m.fuse.out = r.d_dst; m.fuse.self_status = 1;
            if (r.enc) { m.fuse.aes = dev->d_aes; m.fuse.key = c->hd_key; m.fuse.key_on_host = 1; }
        }
        __atomic_store_n(&c->h_segflag[k], 0u, __ATOMIC_RELEASE);
        m.done = c->d_segdone + k; m.flag = c->hd_segflag + k;
        rc = svc_submit(dev, m, &c->h_segflag[k], &ids[k]);
        if (rc == TSX_E_INVAL) return abandon_all(rc);
        submitted = k + 1;
        if (rc) return abandon_all(rc);
    }
    // ---- completions, in order; a piece's bytes travel back (or are packed down) while the later pieces still run ----
    size_t packed_at = 0; bool packed_full = false;
    const auto t_pub = std::chrono::steady_clock::now();
    for (size_t k = 0; k < ns; k++) {
        const tsx_sub& sb = subs[k];
        if ((rc = svc_wait(dev, &c->h_segflag[k]))) return abandon_all(rc);
        svc_retire(dev, ids[k], false);
        ids[k] = 0;
        if (k + 1 == ns) t.zstd_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_pub).count();
        if (!self_status) {
            // stages_separate: the frames are in the staging buffer; GCM (or the copy into the slots) and the status publication follow
            tsx_chunk_desc* dd = c->d_descs + sb.lo; int32_t* ds = c->d_status + sb.lo; uint32_t* dz = c->d_zlen + sb.lo;
            uint8_t* dmid = c->d_mid + (size_t)sb.lo * c->mid_stride;
            if (r.enc) {
                if (k == 0) {
                    tsx_gcm_key_build_host(r.params->key, r.params->aad, r.params->aad_len, c->h_key);
                    HIPCHK_AB(hipMemcpyAsync(c->d_key, c->h_key, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, c->st));
                }
                hipLaunchKernelGGL(plan_gcm_kernel, dim3((sb.n + 255) / 256), dim3(256), 0, c->st, dd, sb.n, (const uint32_t*)dz, (uint64_t)c->mid_stride, 1, 0, 0, c->d_gchunks + sb.lo, ds);
                tsx_launch_gcm(c->st, dev->d_aes, c->d_key, c->d_gchunks + sb.lo, sb.n, (uint32_t)tsx_transformed_bound(r.max_len, TSX_COMPRESS), dmid, r.d_dst,
                               c->d_partials + (size_t)sb.lo * c->partials_per_chunk, ds, 0);
                t.gcm_launches += 2;
            } else {
                const uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
                hipLaunchKernelGGL(copy_chunks_kernel, dim3(sb.n * bpc), dim3(256), 0, c->st, dd, (const uint32_t*)dz, (uint64_t)c->mid_stride, 1, (const uint8_t*)dmid, r.d_dst, ds, bpc);
            }
            hipLaunchKernelGGL(publish_status_kernel, dim3((sb.n + 255) / 256), dim3(256), 0, c->st, dd, (const int32_t*)ds, sb.n, (tsx_chunk_desc*)nullptr);
            HIPCHK_AB(hipMemcpyAsync(c->h_descs + sb.lo, dd, (size_t)sb.n * sizeof(tsx_chunk_desc), hipMemcpyDeviceToHost, c->st));
            if (hipStreamSynchronize(c->st) != hipSuccess) return abandon_all(TSX_E_DEVICE);
        }
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
        } else if (r.host) {
            if ((rc = copy_back(r, sb, &packed_at, &packed_full, cout_))) return abandon_all(rc);
        }
    }
    const auto t_done = std::chrono::steady_clock::now();
    if (r.host && !zc_dst) { HIPCHK_AB(hipEventRecord(c->ev[3], cout_)); HIPCHK_AB(hipEventSynchronize(c->ev[3])); }
#undef HIPCHK_AB
    t.zstd_launches = (uint32_t)ns;
    t.d2h_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_done).count();
    t.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (!r.host) { t.h2d_ms = 0; t.d2h_ms = 0; }
    return TSX_OK;
}

// ---- everything else: forward chain without compression, the inverse chain, CRC only ---------------------------------------------------
// Enqueues the kernels of chunks [lo, lo + n) on compute stream st; e[0..4] are recorded at the stage boundaries.
// stage_key: this piece's first kernel also brings the key schedule the host built (c->h_key) to the device (c->d_key); c->ev_key is
// recorded behind it.  Nothing here is a copy: descriptors and key come in through begin_batch_kernel, results leave through the kernels
// that produce them (tsx_chunk_desc mirror in pinned memory).
static int launch_stages(const tsx_run& r, const tsx_sub& sb, hipEvent_t* e, hipStream_t st, bool stage_key) {
    tsx_ctx* c = r.c;
    const uint32_t n = sb.n, lo = sb.lo, flags = r.flags;
    tsx_chunk_desc* dd = c->d_descs + lo;
    tsx_chunk_desc* const hm = c->hd_descs + lo;                       // the batch's descriptors in pinned memory, as the device addresses them
    int32_t* ds = c->d_status + lo;
    uint32_t* dz = c->d_zlen + lo;
    tsx_gcm_chunk* dg = c->d_gchunks + lo;
    uint8_t* dmid = c->d_mid ? c->d_mid + (size_t)lo * c->mid_stride : nullptr;
    // the Zstd workspace of chunk i is slot i of the batch, whichever piece it travels in: pieces of one batch co-reside
    void* dzw = c->d_zwork ? (uint8_t*)c->d_zwork + (size_t)lo * tsx_zstd_workspace_bytes(1, 0) : nullptr;
    uint32_t* const dpart = c->d_partials + (size_t)lo * c->partials_per_chunk;     // this piece's slice of the CRC / GHASH partial sums
    tsx_timing& t = c->timing;
    memcpy(c->h_descs + lo, r.descs + lo, (size_t)n * sizeof(tsx_chunk_desc));
    {
        const uint32_t kw = stage_key ? (uint32_t)(sizeof(tsx_gcm_key) / 16) : 0u;
        const uint32_t blocks = std::max((n + 255u) / 256u, stage_key ? 6u : 1u);
        hipLaunchKernelGGL(begin_batch_kernel, dim3(blocks), dim3(256), 0, st, (const tsx_chunk_desc*)hm, dd, ds, n,
                           stage_key ? (const uint4*)c->hd_key : (const uint4*)nullptr, (uint4*)c->d_key, kw);
        if (stage_key) HIPCHK(hipEventRecord(c->ev_key, st));
    }
    HIPCHK(hipEventRecord(e[0], st));
    if (r.mode == 2) {
        tsx_launch_crc32c(st, c->dev->d_crc, r.d_src, dd, n, r.max_len, dpart, 0);
        HIPCHK(hipEventRecord(e[1], st)); HIPCHK(hipEventRecord(e[2], st));
        t.crc_launches += 2;
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n, hm);      // a reused descriptor must not keep an old status
    } else if (r.mode == 0) {
        // forward chain WITHOUT compression (producers compress: RemoteStorageManager.java:381-398 leaves Zstd out): CRC32C of the chunk,
        // then AES-256-GCM (or the plain copy) as batch kernels - milliseconds of work, no residency to protect
        if (flags & TSX_CRC) { tsx_launch_crc32c(st, c->dev->d_crc, r.d_src, dd, n, r.max_len, dpart, 0); t.crc_launches += 2; }
        HIPCHK(hipEventRecord(e[1], st));
        HIPCHK(hipEventRecord(e[2], st));
        if (r.enc) {
            hipLaunchKernelGGL(plan_gcm_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, n, (const uint32_t*)dz, (uint64_t)c->mid_stride, 0, 0, 0, dg, ds);
            tsx_launch_gcm(st, c->dev->d_aes, c->d_key, dg, n, r.max_len, r.d_src, r.d_dst, dpart, ds, 0);
            t.gcm_launches += 2;
        } else {
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, dd, (const uint32_t*)dz, (uint64_t)c->mid_stride, 0, r.d_src, r.d_dst, ds, bpc);
        }
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n, hm);
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
            // (a batch decode next to running uploads takes the scratch-free build of the chunk-serial kernel: a dispatch that needs scratch
            //  waits for the compressor service's kernel - and its scratch - to go away)
            bool busy_service = false;
            if (!skip) { std::lock_guard<std::mutex> lk(c->dev->svc->mu); busy_service = svc_running_locked(*c->dev->svc) || !c->dev->svc->out.empty(); }
            t.unzstd_launches += tsx_launch_zstd_decompress(st, c->dev->d_zc, zsrc, r.enc ? 1 : 0, (uint64_t)c->mid_stride, dd, n, r.d_dst, ds, dzw, skip, skip_stride, busy_service);
        } else if (!r.enc) {
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(copy_chunks_kernel, dim3(n * bpc), dim3(256), 0, st, dd, (const uint32_t*)dz, (uint64_t)0, 0, r.d_src, r.d_dst, ds, bpc);
        }
        hipLaunchKernelGGL(publish_status_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dd, (const int32_t*)ds, n, hm);
        if (r.enc && !r.comp) {            // decrypted straight into the caller's slots: nothing of a chunk that failed its tag check stays
            uint32_t bpc = r.max_len > (1u << 20) ? 16 : 1;
            hipLaunchKernelGGL(scrub_failed_kernel, dim3(n * bpc), dim3(256), 0, st, (const tsx_chunk_desc*)dd, (const int32_t*)ds, r.d_dst, bpc);
        }
    }
    HIPCHK(hipEventRecord(e[3], st));
    if (r.mode == 1 && (flags & TSX_CRC)) {
        // CRC of the restored bytes; upper bound of a restored chunk is its slot capacity
        tsx_launch_crc32c(st, c->dev->d_crc, r.d_dst, dd, n, r.max_out, dpart, 1, hm);
        t.crc_launches += 2;
    }
    HIPCHK(hipEventRecord(e[4], st));                                   // the batch's descriptors in pinned memory are filled in when this event has passed
    return TSX_OK;
}

// The bytes chunks [lo, lo + n) produced travel back on out_st (host-memory batches; their descriptors are on the host already).
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
    const tsx_trace tr;
    // Zero-copy output for the encrypt-only forward chain (producers compress: RemoteStorageManager.java:381-398 leaves Zstd out): the GCM
    // kernel's waves write IV || C || TAG straight into the caller's slots when the device can address the WHOLE buffer - the compress path
    // got this in round 4.  Slot layout only (a packed batch still travels through the device buffer: its chunks are packed by the copies).
    uint8_t* zc_dst = nullptr;
    if (r.mode == 0 && r.enc && r.host && !r.packed && !g_cfg.no_zero_copy_out) zc_dst = device_alias_of_range(r.dst, r.dst_size);
    c->last_zero_copy = zc_dst != nullptr;
    rc = reserve_or_drain(c, n, max_len, r.mode == 1 ? max_out : 0, r.flags, r.host, in_bytes, zc_dst ? 0 : out_bytes);
    if (rc) return rc;
    tr.mark("workspace reserved");
    r.d_src = r.host ? c->d_in : (const uint8_t*)r.src;
    r.d_dst = zc_dst ? zc_dst : r.host ? c->d_out : (uint8_t*)r.dst;
    hipStream_t st = c->st;
    memset(&c->timing, 0, sizeof c->timing);
    // ---- sub-batches: a host-memory batch is cut into pieces whose H2D copy, kernels and D2H copy overlap (device-memory batches have
    // nothing to overlap): pieces of >= 64 MiB in order on ONE compute stream, three streams in all (the chunk-serial frame decoder is bound
    // by per-chunk latency - ~30 ms however few chunks a launch has: its pieces are >= 512 chunks).
    std::vector<tsx_sub> subs;
    const bool pipelined = r.host && monotonic && !g_cfg.no_pipeline;
    // A fetch of 16 .. 256 chunks (a consumer catching up: ChunkCache.java:159-184 with a large prefetch.max.size) decodes in the block
    // form, whose cost is a ~1.5 ms chain of short kernels + a part proportional to the chunks: cut into up to 8 pieces of >= 8 chunks,
    // spread over the context's compute streams so that the pieces' chains overlap each other, the later pieces' copy-in and the earlier
    // pieces' copy-out.  (Round 3 only cut batches of >= 512 chunks: 64 chunks took 7 ms device resident and 17 ms host to host.)
    c->blk_pieces.clear(); c->last_used_blocks = false;
    const bool inv_blocks = r.mode == 1 && r.comp && pipelined && n >= 16 && c->d_bwork && dec_use_blocks(n, max_out) && !g_cfg.no_dec_pieces;
    if (pipelined) {
        size_t budget = g_cfg.sub_bytes > 0 ? (size_t)g_cfg.sub_bytes : TSX_SUB_BYTES;
        size_t max_subs = TSX_MAX_SUBS;
        if (in_bytes / budget + 1 > max_subs) budget = in_bytes / max_subs + 1;
        uint32_t min_chunks = r.comp ? 512u : 1u;
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
    const bool multi = inv_blocks && ns > 1;                            // pieces side by side: piece k on compute stream k mod 4
    for (size_t k = 1; k < ns; k++) for (auto& e : c->sub_ev[k]) if (!e) HIPCHK(hipEventCreate(&e));
    if (multi) for (size_t k = 1; k < ns && k < 4; k++) if (!c->st_pc[k - 1]) HIPCHK(hipStreamCreateWithFlags(&c->st_pc[k - 1], hipStreamNonBlocking));
    auto stream_of = [&](size_t k) { return (multi && k % 4) ? c->st_pc[k % 4 - 1] : st; };
    HIPCHK(hipEventRecord(c->ev[0], st));
    bool keyraw_written = false, stage_key = false;
    if (r.enc) {
        // The key schedule is built on the host (~10 us with the host's carry-less multiplier) and travels as ONE 21 KB copy in front of the
        // batch's GCM kernels instead of a raw-key copy + gcm_setup_kernel: that kernel was 0.17 of a single-chunk fetch's 1.7 ms and, on a
        // busy device, one more small kernel waiting for a slot.  Test hook gcm_setup_kernel keeps the kernel (both
        // produce the same schedule).
        if (!g_cfg.gcm_setup_kernel) {
            tsx_gcm_key_build_host(r.params->key, r.params->aad, r.params->aad_len, c->h_key);    // piece 0's first kernel takes it to the device
            stage_key = true;
        } else {
            memcpy(c->h_keyraw, r.params->key, 32); memcpy(c->h_keyraw + 32, r.params->aad, 64);
            HIPCHK(hipMemcpyAsync(c->d_keyraw, c->h_keyraw, 96, hipMemcpyHostToDevice, st));
            keyraw_written = true;
            tsx_launch_gcm_setup(st, c->dev->d_aes, c->d_keyraw, c->d_keyraw + 32, r.params->aad_len, c->d_key);
            if (multi) HIPCHK(hipEventRecord(c->ev_key, st));
        }
    }
    (void)keyraw_written;                                               // (run_batch wipes both device copies whatever was written)
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
        if (multi && ks != st && r.enc) HIPCHK(hipStreamWaitEvent(ks, c->ev_key, 0));      // (piece 0 went first: its begin kernel staged the key)
        return launch_stages(r, sb, e, ks, stage_key && k == 0);
    };
    // The restored chunks of a fetch are what crosses PCIe (4 MiB each against 1.3 MB in): one copy stream moves them at ~31 GB/s - 8.7 of a
    // 64-chunk window's 10.4 ms; the pieces' copies alternate between two streams.
    bool out2 = inv_blocks && multi && r.host;
    if (out2 && !c->st_out2 && hipStreamCreateWithFlags(&c->st_out2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->st_out2 = nullptr; out2 = false; }
    auto collect_piece = [&](size_t k) -> int {
        const tsx_sub& sb = subs[k];
        if (tr.on) {                                                    // (test hook: which stage a waiting batch is waiting in)
            static const char* const names[6] = {"descriptors up + status init", "stage 1", "stage 2", "stage 3", "descriptors down", "copy-in landed"};
            if (r.host) { (void)hipEventSynchronize(c->sub_ev[k][5]); tr.mark(names[5]); }
            for (int q = 0; q < 5; q++) { (void)hipEventSynchronize(c->sub_ev[k][q]); tr.mark(names[q]); }
        }
        HIPCHK(wait_event_watching(c, c->sub_ev[k][4]));                // descriptors of piece k are on the host
        memcpy(r.descs + sb.lo, c->h_descs + sb.lo, (size_t)sb.n * sizeof(tsx_chunk_desc));
        if (zc_dst) return TSX_OK;                                      // the bytes are in the caller's buffer already
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
            if (k < ns) tr.mark("piece enqueued (copy-in + kernels)");
            if (k > 0 && (rc = collect_piece(k - 1))) return rc;
            if (k > 0) tr.mark("piece collected (descriptors back, copy-out queued)");
        }
    }
    if (r.host) HIPCHK(hipEventRecord(c->ev[3], c->st_out));
    if (r.enc) {
        // the key material leaves the device behind the batch's last kernel (every piece's chain has been joined into st above); the pinned
        // originals are wiped by run_batch
        hipLaunchKernelGGL(wipe_key_kernel, dim3(6), dim3(256), 0, st, (uint4*)c->d_key, (uint32_t)(sizeof(tsx_gcm_key) / 16), (uint4*)c->d_keyraw);
        c->key_wiped = true;
    }
    HIPCHK(hipEventRecord(c->ev[1], st));
    HIPCHK(hipStreamSynchronize(st));
    tr.mark("compute stream idle");
    if (r.host) { HIPCHK(hipStreamSynchronize(c->st_in)); HIPCHK(hipStreamSynchronize(c->st_out)); if (out2) HIPCHK(hipStreamSynchronize(c->st_out2)); }
    tr.mark("copy streams idle");
    HIPCHK(hipGetLastError());
    tsx_timing& t = c->timing;
    for (size_t k = 0; k < ns; k++) {
        hipEvent_t* e = c->sub_ev[k];
        const float a = ev_ms(e[0], e[1]), b = ev_ms(e[1], e[2]), d = ev_ms(e[2], e[3]), f = ev_ms(e[3], e[4]);
        if (r.mode == 2) t.crc_ms += a;
        else if (r.mode == 0) { t.crc_ms += (r.flags & TSX_CRC) ? a : 0; t.gcm_ms += d; }
        else { t.gcm_ms += r.enc ? b : 0; t.unzstd_ms += d; t.crc_ms += (r.flags & TSX_CRC) ? f : 0; }
    }
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
                     size_t dst_size, int mem_kind, int mode /*0 transform, 1 detransform, 2 crc only*/, bool pooled = false) {
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
    r.flags = flags; r.host = mem_kind != TSX_MEM_DEVICE; r.packed = packed; r.pooled = pooled;
    r.enc = mode != 2 && (flags & TSX_ENCRYPT); r.comp = mode != 2 && (flags & TSX_COMPRESS);
    r.fuse_stages = r.comp && !g_cfg.stages_separate;
    const bool service = mode == 0 && r.comp;
    c->key_wiped = false;
    struct fg_scope {                                                   // every batch of ordinary kernels claims the reserved CUs for its duration (+ fetch_quiet_ms)
        tsx_device* d;
        explicit fg_scope(tsx_device* dev) : d(dev) { if (d) svc_foreground_begin(d); }
        ~fg_scope() { if (d) svc_foreground_end(d); }
    } fg(service && r.fuse_stages ? nullptr : c->dev);
    const int rc = service ? run_compress(r) : run_batch_inner(r);
    // Whatever happened: nothing of this call is still in flight when it returns (the copies reference the caller's buffers), and
    // the data key does not stay behind in a context that may serve another segment next (the native side zeroises its
    // copy; the round keys and H powers are as good as the key).
    if (service && r.fuse_stages) {
        // nothing of the key was uploaded (every wave took and wiped its own copy of the schedule); what is left is the pinned original,
        // and run_compress returns only when none of the batch's members can still be running
        if (r.enc) { memset(c->h_keyraw, 0, 128); memset(c->h_key, 0, sizeof(tsx_gcm_key)); }
        if (rc != TSX_OK) (void)hipGetLastError();
        return rc;
    }
    if (r.enc) {
        hipStreamSynchronize(c->st);
        for (auto& q : c->st_pc) if (q) hipStreamSynchronize(q);
        memset(c->h_keyraw, 0, 128); memset(c->h_key, 0, sizeof(tsx_gcm_key));
        if (!c->key_wiped) {
            // the batch failed before its wipe kernel was queued: both device copies are cleared here, whichever of them it wrote
            hipMemcpyAsync(c->d_keyraw, c->dev->h_zeros, 128, hipMemcpyHostToDevice, c->st);
            hipMemcpyAsync(c->d_key, c->dev->h_zeros, sizeof(tsx_gcm_key), hipMemcpyHostToDevice, c->st);
        }
    }
    hipStreamSynchronize(c->st_in); hipStreamSynchronize(c->st); hipStreamSynchronize(c->st_out);
    if (c->st_out2) hipStreamSynchronize(c->st_out2);
    for (auto& q : c->st_pc) if (q) hipStreamSynchronize(q);
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
    for (size_t i = 0; i < sizeof(tsx_gcm_key); i++) acc |= ((const uint8_t*)c->h_key)[i];
    return acc;
}

// Test hooks (not part of the ABI): members the context's last compressing batch went as; whether its waves wrote into the caller's buffer.
extern "C" int tsx_debug_last_members(tsx_ctx* c) { return c ? (int)c->last_members : TSX_E_INVAL; }
extern "C" int tsx_debug_last_zero_copy(tsx_ctx* c) { return c ? (int)c->last_zero_copy : TSX_E_INVAL; }

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
// per-thread direct ByteBuffers once): copies from / to it go by DMA and overlap fully instead of being staged by the runtime, and
// compressing batches write their output straight into it (zero-copy output: the registered extent is what the device may touch).
// Portable: the pinning holds for every device of the node, whichever one the pool picks for a batch.
extern "C" int tsx_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return TSX_E_INVAL;
    { std::lock_guard<std::mutex> lk(g_mu); if (g_devs.empty()) return TSX_E_DEVICE; }
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { tsx_set_err("hipHostRegister", e); (void)hipGetLastError(); return TSX_E_DEVICE; }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_registered.push_back({(uintptr_t)p, bytes});
    return TSX_OK;
}
extern "C" int tsx_host_unregister(void* p) {
    if (!p) return TSX_E_INVAL;
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        for (size_t i = 0; i < g_registered.size(); i++) if (g_registered[i].first == (uintptr_t)p) { g_registered.erase(g_registered.begin() + (long)i); break; }
    }
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { tsx_set_err("hipHostUnregister", e); (void)hipGetLastError(); return TSX_E_DEVICE; }
    return TSX_OK;
}

// ---- device memory helpers -------------------------------------------------------------------------
static int set_dev(int device_index, tsx_device** dev = nullptr) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (device_index < 0 || device_index >= (int)g_devs.size()) return TSX_E_INVAL;
    if (dev) *dev = &g_devs[device_index];
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
    tsx_device* dev = nullptr;
    int rc = set_dev(device_index, &dev); if (rc) return rc;
    svc_free_dev(dev, p);                                               // (given back when the compressor service's kernel is gone: a free waits for every stream)
    return TSX_OK;
}
// (copies of pageable memory and small copies run as kernels of the runtime: guest waves make room for them as for a fetch)
extern "C" int tsx_memcpy_h2d(int device_index, void* d, const void* s, size_t bytes) {
    tsx_device_scope keep;
    tsx_device* dev = nullptr;
    int rc = set_dev(device_index, &dev); if (rc) return rc;
    svc_foreground_begin(dev);
    const hipError_t e = hipMemcpy(d, s, bytes, hipMemcpyHostToDevice);
    svc_foreground_end(dev, false);
    return e == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
extern "C" int tsx_memcpy_d2h(int device_index, void* d, const void* s, size_t bytes) {
    tsx_device_scope keep;
    tsx_device* dev = nullptr;
    int rc = set_dev(device_index, &dev); if (rc) return rc;
    svc_foreground_begin(dev);
    const hipError_t e = hipMemcpy(d, s, bytes, hipMemcpyDeviceToHost);
    svc_foreground_end(dev, false);
    return e == hipSuccess ? TSX_OK : TSX_E_DEVICE;
}
