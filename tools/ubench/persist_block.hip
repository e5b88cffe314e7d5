// What does a kernel that stays on the chip for seconds do to the FIRST operation other streams issue meanwhile?
// (gpurun r05a-r05h: the first fetch after uploads began came back when the compressor service's kernel ended; every later one in 2 ms.)
//   persist_block <scratch_bytes_per_lane> <low|normal> <waves> <seconds> [lds_bytes] [repeat]
// A "persistent" kernel - `waves` one-wave workgroups, `lds_bytes` of LDS each, each lane touching a private array of scratch_bytes (0:
// none) - spins for `seconds` on a stream of the given priority, launched TWICE in a row when repeat = 1 (the second launch behind the
// first, as the service's watchdog relaunches).  Meanwhile the main thread, on another stream, times every 250 ms: an empty kernel, an
// event record + synchronize, a 48-byte pinned H2D copy, a 1 MiB pinned H2D copy.  One line per operation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int WORDS>
__global__ __launch_bounds__(64) void persist(const volatile unsigned* stop, unsigned long long ticks, unsigned* sink, unsigned lds_words) {
    extern __shared__ unsigned lds[];
    unsigned priv[WORDS > 0 ? WORDS : 1];
    if (WORDS > 0) for (int i = 0; i < WORDS; i++) priv[i] = threadIdx.x * 31u + i;
    if (lds_words) lds[threadIdx.x % lds_words] = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    unsigned acc = 0, k = threadIdx.x;
    while (!*stop && wall_clock64() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(127);
        if (WORDS > 0) { k = (k * 1103515245u + 12345u); acc += priv[k % WORDS]; priv[(k >> 8) % WORDS] = acc; }     // dynamic index: stays in scratch
    }
    if (acc == 0xFFFFFFFFu) sink[0] = acc + (lds_words ? lds[0] : 0);
}
__global__ void empty_kernel(unsigned* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: persist_block <scratch_bytes> <low|normal> <waves> <seconds> [lds_bytes] [repeat]\n"); return 2; }
    const int scratch = atoi(argv[1]); const bool low = !strcmp(argv[2], "low"); const unsigned waves = (unsigned)atoi(argv[3]); const double secs = atof(argv[4]);
    const unsigned lds = argc > 5 ? (unsigned)atoi(argv[5]) : 6704; const int repeat = argc > 6 ? atoi(argv[6]) : 0;
    unsigned* stop; CK(hipHostMalloc((void**)&stop, 64, hipHostMallocMapped)); *stop = 0;
    unsigned* dstop; CK(hipHostGetDevicePointer((void**)&dstop, stop, 0));
    unsigned* sink; CK(hipMalloc((void**)&sink, 64));
    hipStream_t ps, ws; int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (low) CK(hipStreamCreateWithPriority(&ps, hipStreamNonBlocking, least)); else CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
    uint8_t *hbuf, *dbuf; CK(hipHostMalloc((void**)&hbuf, 1 << 20, hipHostMallocDefault)); CK(hipMalloc((void**)&dbuf, 1 << 20));
    hipEvent_t ev; CK(hipEventCreate(&ev));
    auto ops = [&](const char* tag, double t0) -> int {
        const char* names[4] = {"empty kernel", "event record + sync", "48 B H2D", "1 MiB H2D"};
        for (int k = 0; k < 4; k++) {
            const double a = now_s();
            if (k == 0) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, ws, (unsigned*)nullptr); CK(hipStreamSynchronize(ws)); }
            else if (k == 1) { CK(hipEventRecord(ev, ws)); CK(hipEventSynchronize(ev)); }
            else if (k == 2) { CK(hipMemcpyAsync(dbuf, hbuf, 48, hipMemcpyHostToDevice, ws)); CK(hipStreamSynchronize(ws)); }
            else { CK(hipMemcpyAsync(dbuf, hbuf, 1 << 20, hipMemcpyHostToDevice, ws)); CK(hipStreamSynchronize(ws)); }
            printf("%s t=%.2f s  %-22s %10.3f ms\n", tag, a - t0, names[k], (now_s() - a) * 1e3); fflush(stdout);
        }
        return 0;
    };
    const double t00 = now_s();
    if (ops("idle", t00)) return 1;
    if (ops("idle", t00)) return 1;
    const unsigned long long ticks = (unsigned long long)(secs * 1e8);
    for (int r = 0; r <= repeat; r++) {
        if (scratch >= 1024) hipLaunchKernelGGL(persist<268>, dim3(waves), dim3(64), lds, ps, dstop, ticks, sink, lds / 4);
        else if (scratch >= 256) hipLaunchKernelGGL(persist<64>, dim3(waves), dim3(64), lds, ps, dstop, ticks, sink, lds / 4);
        else if (scratch > 0) hipLaunchKernelGGL(persist<16>, dim3(waves), dim3(64), lds, ps, dstop, ticks, sink, lds / 4);
        else hipLaunchKernelGGL(persist<0>, dim3(waves), dim3(64), lds, ps, dstop, ticks, sink, lds / 4);
    }
    CK(hipGetLastError());
    const double t0 = now_s();
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    while (now_s() - t0 < secs * (repeat + 1) + 1.0) {
        char tag[64]; snprintf(tag, sizeof tag, "scratch=%d %s waves=%u launch%d", scratch, low ? "low" : "normal", waves, (int)((now_s() - t0) / secs));
        if (ops(tag, t0)) return 1;
        std::this_thread::sleep_for(std::chrono::milliseconds(250));
    }
    *stop = 1;
    CK(hipStreamSynchronize(ps));
    printf("done %.2f s\n", now_s() - t0);
    return 0;
}
