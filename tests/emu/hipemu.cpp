// TEST HARNESS ONLY — scheduler / runtime half of the HIP emulator declared in hip/hip_runtime.h.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <vector>

extern "C" void hipemu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

// ThreadSanitizer flavour (make emu-tsan): the lane fibers are announced to the tool, or it would take a stack switch for a wild jump
#if defined(__SANITIZE_THREAD__)
extern "C" { void* __tsan_get_current_fiber(void); void* __tsan_create_fiber(unsigned flags); void __tsan_destroy_fiber(void* fiber);
             void __tsan_switch_to_fiber(void* fiber, unsigned flags); }
#define TSAN_FIBER_NEW() __tsan_create_fiber(0)
#define TSAN_FIBER_FREE(f_) __tsan_destroy_fiber(f_)
#define TSAN_FIBER_GO(f_) __tsan_switch_to_fiber(f_, 0)
#define TSAN_FIBER_SELF() __tsan_get_current_fiber()
#else
#define TSAN_FIBER_NEW() nullptr
#define TSAN_FIBER_FREE(f_) do {} while (0)
#define TSAN_FIBER_GO(f_) do {} while (0)
#define TSAN_FIBER_SELF() nullptr
#endif

namespace hipemu {

static const size_t kStack = 256 * 1024;

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool done = false;
    void* tsan = nullptr;
    Ctx ctx{};
};
struct Wave {
    unsigned arrive = 0, gen = 0, live = 0;
    uint64_t live_mask = 0;
    uint64_t slots[64];
};
struct Block {
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    unsigned bar_arrive = 0, bar_gen = 0, live = 0;
    const std::function<void()>* entry = nullptr;
};

Ctx* g_ctx = nullptr;
static Block* g_blk = nullptr;
static Fiber* g_cur = nullptr;
static void* g_sched_sp = nullptr;
static void* g_sched_tsan = nullptr;
static std::vector<char*> g_stack_pool;

static void yield() { TSAN_FIBER_GO(g_sched_tsan); hipemu_switch(&g_cur->sp, g_sched_sp); }

static void release_block_barrier_if_complete() {
    Block& b = *g_blk;
    if (b.live && b.bar_arrive == b.live) { b.bar_arrive = 0; b.bar_gen++; }
}
static void release_wave_barrier_if_complete(Wave& w) {
    if (w.live && w.arrive == w.live) { w.arrive = 0; w.gen++; }
}

static void trampoline() {
    Fiber* f = g_cur;
    (*g_blk->entry)();
    f->done = true;
    Block& b = *g_blk;
    Wave& w = b.waves[f->ctx.flat_tid / 64];
    w.live--; w.live_mask &= ~(1ull << (f->ctx.flat_tid & 63));
    b.live--;
    release_wave_barrier_if_complete(w);
    release_block_barrier_if_complete();
    TSAN_FIBER_GO(g_sched_tsan);
    hipemu_switch(&f->sp, g_sched_sp);
    std::abort();  // never resumed
}

void block_barrier() {
    Block& b = *g_blk;
    unsigned gen = b.bar_gen;
    b.bar_arrive++;
    release_block_barrier_if_complete();
    while (b.bar_gen == gen) yield();
}
void wave_barrier() {
    Wave& w = g_blk->waves[g_cur->ctx.flat_tid / 64];
    unsigned gen = w.gen;
    w.arrive++;
    release_wave_barrier_if_complete(w);
    while (w.gen == gen) yield();
}
uint64_t* wave_slots() { return g_blk->waves[g_cur->ctx.flat_tid / 64].slots; }
uint64_t wave_live_mask() { return g_blk->waves[g_cur->ctx.flat_tid / 64].live_mask; }
unsigned lane_id() { return g_cur->ctx.flat_tid & 63; }

static void init_fiber(Fiber& f) {
    if (!f.stack) {
        if (!g_stack_pool.empty()) { f.stack = g_stack_pool.back(); g_stack_pool.pop_back(); }
        else f.stack = (char*)aligned_alloc(64, kStack);
    }
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)(top - 16);   // sp[0] = return address slot, 16-byte aligned
    sp[0] = (void*)&trampoline;
    sp -= 6;                          // r15 r14 r13 r12 rbx rbp
    for (int i = 0; i < 6; i++) sp[i] = nullptr;
    f.sp = sp;
    f.done = false;
}

static std::atomic<int> g_force_reserved{0};
}  // namespace hipemu
uint64_t hipemu_clock_100mhz() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10; }
static std::atomic<int> g_hipemu_key_shift{0}, g_hipemu_yield_after{0}, g_hipemu_relocate_after{0}, g_hipemu_relocated_block{-1};
uint32_t hipemu_cu_key() {
    if (hipemu::g_force_reserved.load(std::memory_order_relaxed) > 0) return 3u;
    // hipemu_relocate_after(n): the n-th look from now finds its block on the reserved CU, and so does every later look of that block in this launch -
    // a wave that the hardware's scheduler saved and restored somewhere else
    if (g_hipemu_relocate_after.load(std::memory_order_relaxed) > 0 && g_hipemu_relocate_after.fetch_sub(1) == 1) g_hipemu_relocated_block = (int)hipemu::g_ctx->bid.x;
    if (g_hipemu_relocated_block.load(std::memory_order_relaxed) == (int)hipemu::g_ctx->bid.x) return 3u;
    return (hipemu::g_ctx->bid.x + (unsigned)g_hipemu_key_shift.load(std::memory_order_relaxed)) % 4u;
}
extern "C" void hipemu_relocate_after(int n) { g_hipemu_relocate_after = n; g_hipemu_relocated_block = -1; }
extern "C" void hipemu_force_reserved_launches(int k) { hipemu::g_force_reserved = k; }
extern "C" void hipemu_cu_key_shift(int k) { g_hipemu_key_shift = k; }
extern "C" void hipemu_force_yield_after(int n) { g_hipemu_yield_after = n; }
uint32_t hipemu_yield_probe(const uint32_t* p) {
    const uint32_t v = __atomic_load_n(p, __ATOMIC_ACQUIRE);
    if (v) return v;
    if (g_hipemu_yield_after.load(std::memory_order_relaxed) > 0 && g_hipemu_yield_after.fetch_sub(1) == 1) {
        __atomic_store_n(const_cast<uint32_t*>(p), 1u, __ATOMIC_RELEASE);     // as if a fetch had arrived on the host at this very moment
        return 1u;
    }
    return 0u;
}
namespace hipemu {

// One grid at a time: the scheduler state (and `__shared__` = static storage) is process-wide, while the front end under test is
// called from several host threads (pooled contexts, device hints).
static std::mutex g_grid_mu;
void run_grid(dim3 grid, dim3 block, const std::function<void()>& entry) {
    std::lock_guard<std::mutex> grid_lock(g_grid_mu);
    unsigned nthreads = block.x * block.y * block.z;
    Block blk;
    blk.fibers.resize(nthreads);
    blk.entry = &entry;
    Block* saved_blk = g_blk;
    g_blk = &blk;
    g_sched_tsan = TSAN_FIBER_SELF();
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                blk.waves.assign((nthreads + 63) / 64, Wave{});
                blk.bar_arrive = 0; blk.live = nthreads;
                for (unsigned t = 0; t < nthreads; t++) {
                    Fiber& f = blk.fibers[t];
                    init_fiber(f);
                    if (!f.tsan) f.tsan = TSAN_FIBER_NEW();
                    f.ctx.tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
                    f.ctx.bid = {bx, by, bz};
                    f.ctx.bdim = block; f.ctx.gdim = grid; f.ctx.flat_tid = t;
                    Wave& w = blk.waves[t / 64];
                    w.live++; w.live_mask |= 1ull << (t & 63);
                }
                unsigned remaining = nthreads;
                while (remaining) {
                    // wave-major round robin: lanes of one wave advance together between rendezvous points
                    for (unsigned t = 0; t < nthreads; t++) {
                        Fiber& f = blk.fibers[t];
                        if (f.done) continue;
                        g_cur = &f; g_ctx = &f.ctx;
                        TSAN_FIBER_GO(f.tsan);
                        hipemu_switch(&g_sched_sp, f.sp);
                        if (f.done) remaining--;
                    }
                }
            }
    for (auto& f : blk.fibers) { if (f.stack) g_stack_pool.push_back(f.stack); if (f.tsan) TSAN_FIBER_FREE(f.tsan); }
    if (g_force_reserved.load(std::memory_order_relaxed) > 0) g_force_reserved--;
    g_blk = saved_blk; g_cur = nullptr; g_ctx = nullptr;
}

}  // namespace hipemu

// ---- host runtime ---------------------------------------------------------------------------------
struct hipemu_stream { int dummy; };
struct hipemu_event { std::chrono::steady_clock::time_point t; };

// HIPEMU_DEVICES=n makes the harness report n identical "devices" (one address space), so the multi-device dispatch of the
// front end (context pools per device, device hints) can be exercised without hardware.
static int emu_devices() { const char* e = getenv("HIPEMU_DEVICES"); int n = e ? atoi(e) : 1; return n < 1 ? 1 : n > 16 ? 16 : n; }
static thread_local int g_emu_dev = 0;
hipError_t hipGetDeviceCount(int* n) { *n = emu_devices(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= emu_devices()) return hipErrorInvalidValue; g_emu_dev = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_emu_dev; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    std::memset(p, 0, sizeof *p);
    std::snprintf(p->name, sizeof p->name, "hipemu (CPU test harness)");
    std::snprintf(p->gcnArchName, sizeof p->gcnArchName, "emu");
    p->multiProcessorCount = 4;
    p->totalGlobalMem = (size_t)8 << 30;
    return hipSuccess;
}
// Fault injection for the front end's out-of-memory paths: hipemu_fail_alloc_at(k) makes the k-th allocation from now on (device
// or pinned host, k >= 1) fail once; 0 disarms.  hipemu_alloc_calls() counts allocations since the last arming.
static std::atomic<long> g_fail_at{0}, g_alloc_calls{0};
extern "C" void hipemu_fail_alloc_at(long k) { g_alloc_calls = 0; g_fail_at = k; }
extern "C" long hipemu_alloc_calls() { return g_alloc_calls; }
hipError_t hipMalloc(void** p, size_t n) {
    const long k = ++g_alloc_calls;
    if (g_fail_at > 0 && k == g_fail_at) { g_fail_at = 0; *p = nullptr; return hipErrorOutOfMemory; }
    *p = aligned_alloc(256, (n + 255) & ~(size_t)255);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)6 << 30; *total_b = (size_t)8 << 30; return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeHost; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipStreamCreate(hipStream_t* s) { *s = new hipemu_stream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
