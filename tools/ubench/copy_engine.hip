// Which engine moves a host<->device copy, and how fast?  copy_engine <d2h|h2d> <malloc|registered|portable> <bytes> <count> [hog] [host_misalign] [bg]
// bg = 1: another thread keeps 64 MiB copies of the OTHER direction going on its own stream meanwhile; 2: a long kernel runs meanwhile;
// 3: a 50 ms kernel on the SAME stream in front of the copies; 4: an event record behind every copy; 5: both directions alternate on the same stream
// host_misalign = bytes added to every host address (TSX_MEM_HOST_PACKED writes chunk i right behind chunk i - 1: any alignment);
// hog = number of OTHER streams that do one small copy in each direction first (does the runtime hand its SDMA engines to the first
// streams that copy, and blit kernels to the rest?)
// One non-blocking stream, `count` hipMemcpyAsync of `bytes` each between a device buffer and pinned host memory (hipHostMalloc'ed, or
// malloc'ed + hipHostRegister'ed as tsx_host_register does), timed with events.  Run under
//   rocprofv3 --kernel-trace --memory-copy-trace --stats
// to see whether the copies were SDMA transfers (memory-copy records) or __amd_rocclr_copyBuffer kernels (which need CU slots - on a
// chip full of compressor waves they wait).  profiles/r03_copy_engine_probe.txt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <atomic>
__global__ void spin_kernel(unsigned long long cycles, unsigned* out) { const unsigned long long t0 = clock64(); while (clock64() - t0 < cycles) ; if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = 1; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: copy_engine <d2h|h2d> <malloc|registered|portable> <bytes> <count> [hog]\n"); return 2; }
    const bool d2h = !strcmp(argv[1], "d2h"), portable = !strcmp(argv[2], "portable"), reg = portable || !strcmp(argv[2], "registered");
    const size_t bytes = strtoull(argv[3], nullptr, 0); const int count = atoi(argv[4]);
    const size_t span = bytes * (size_t)(count < 64 ? count : 64) + 64;
    uint8_t* dev = nullptr; CK(hipMalloc((void**)&dev, span)); CK(hipMemset(dev, 1, span));
    uint8_t* host = nullptr;
    if (reg) { if (posix_memalign((void**)&host, 4096, span)) return 3; memset(host, 2, span); CK(hipHostRegister(host, span, portable ? hipHostRegisterPortable : hipHostRegisterDefault)); }
    else { CK(hipHostMalloc((void**)&host, span, hipHostMallocDefault)); memset(host, 2, span); }
    const int hog = argc > 5 ? atoi(argv[5]) : 0;
    const size_t mis = argc > 6 ? (size_t)atoi(argv[6]) : 0;
    for (int h = 0; h < hog; h++) {
        hipStream_t hs; CK(hipStreamCreateWithFlags(&hs, hipStreamNonBlocking));
        CK(hipMemcpyAsync(host, dev, 1u << 20, hipMemcpyDeviceToHost, hs)); CK(hipMemcpyAsync(dev, host, 1u << 20, hipMemcpyHostToDevice, hs));   // (big enough not to be a blit kernel anyway)
        CK(hipStreamSynchronize(hs));                                   // the stream stays alive
    }
    const int bg = argc > 7 ? atoi(argv[7]) : 0;
    std::atomic<bool> stop{false};
    std::thread bgt;
    uint8_t* dev2 = nullptr; uint8_t* host2 = nullptr; hipStream_t bs = nullptr;
    if (bg == 1 || bg == 2) { CK(hipMalloc((void**)&dev2, 64u << 20)); CK(hipHostMalloc((void**)&host2, 64u << 20, hipHostMallocDefault)); CK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking)); }
    if (bg == 1) bgt = std::thread([&] { while (!stop) { for (int k = 0; k < 4; k++) { if (d2h) (void)hipMemcpyAsync(dev2, host2, 64u << 20, hipMemcpyHostToDevice, bs); else (void)hipMemcpyAsync(host2, dev2, 64u << 20, hipMemcpyDeviceToHost, bs); } (void)hipStreamSynchronize(bs); } });
    if (bg == 2) hipLaunchKernelGGL(spin_kernel, dim3(256 * 32), dim3(256), 0, bs, 2400000000ull, (unsigned*)nullptr);      // ~1 s on every CU slot
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; w++) {                                       // first round warms up
        CK(hipEventRecord(a, st));
        if (bg == 3) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, st, 120000000ull, (unsigned*)nullptr);
        for (int i = 0; i < count; i++) {
            const size_t off = (size_t)(i % 64) * bytes;
            if (bg == 4 && i) CK(hipEventRecord(b, st));
            if (bg == 5 && (i & 1)) { if (d2h) CK(hipMemcpyAsync(dev + off, host + mis + off, bytes, hipMemcpyHostToDevice, st)); else CK(hipMemcpyAsync(host + mis + off, dev + off, bytes, hipMemcpyDeviceToHost, st)); continue; }
            if (d2h) CK(hipMemcpyAsync(host + mis + off, dev + off, bytes, hipMemcpyDeviceToHost, st));
            else CK(hipMemcpyAsync(dev + off, host + mis + off, bytes, hipMemcpyHostToDevice, st));
        }
        CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    }
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    stop = true; if (bgt.joinable()) bgt.join();
    printf("%s %-10s hog %d misalign %zu bg %d %10zu B x %4d: %8.3f ms  %7.2f GB/s  %8.1f us per copy\n", argv[1], argv[2], hog, mis, bg, bytes, count, ms, bytes * (double)count / ms / 1e6, ms * 1e3 / count);
    return 0;
}
