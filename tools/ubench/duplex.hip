// Do an H2D and a D2H copy on two streams overlap on this box (PCIe full duplex through the SDMA engines)?  Pinned host memory
// (hipHostMalloc) and registered malloc'ed memory (hipHostRegister), 1 GiB each way, also cut into 64 MiB pieces.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t B = (size_t)1 << 30, P = (size_t)64 << 20;
    void *d0, *d1; CHK(hipMalloc(&d0, B)); CHK(hipMalloc(&d1, B));
    hipStream_t s0, s1; CHK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int kind = 0; kind < 2; kind++) {
        char *h0, *h1;
        if (kind == 0) { CHK(hipHostMalloc((void**)&h0, B)); CHK(hipHostMalloc((void**)&h1, B)); }
        else { h0 = (char*)aligned_alloc(4096, B); h1 = (char*)aligned_alloc(4096, B); for (size_t i = 0; i < B; i += 4096) { h0[i] = 1; h1[i] = 2; }
               CHK(hipHostRegister(h0, B, hipHostRegisterDefault)); CHK(hipHostRegister(h1, B, hipHostRegisterDefault)); }
        for (int rep = 0; rep < 2; rep++) {
            double t = now(); CHK(hipMemcpyAsync(d0, h0, B, hipMemcpyHostToDevice, s0)); CHK(hipStreamSynchronize(s0)); double a = now() - t;
            t = now(); CHK(hipMemcpyAsync(h1, d1, B, hipMemcpyDeviceToHost, s1)); CHK(hipStreamSynchronize(s1)); double b = now() - t;
            t = now(); CHK(hipMemcpyAsync(d0, h0, B, hipMemcpyHostToDevice, s0)); CHK(hipMemcpyAsync(h1, d1, B, hipMemcpyDeviceToHost, s1));
            CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1)); double c = now() - t;
            t = now();
            for (size_t o = 0; o < B; o += P) { CHK(hipMemcpyAsync((char*)d0 + o, h0 + o, P, hipMemcpyHostToDevice, s0)); CHK(hipMemcpyAsync(h1 + o, (char*)d1 + o, P, hipMemcpyDeviceToHost, s1)); }
            CHK(hipStreamSynchronize(s0)); CHK(hipStreamSynchronize(s1)); double e = now() - t;
            printf("%s: H2D %.1f ms (%.1f GB/s)  D2H %.1f ms (%.1f GB/s)  both at once %.1f ms  both, 64 MiB pieces %.1f ms\n", kind ? "registered" : "hipHostMalloc",
                   a * 1e3, B / a / 1e9, b * 1e3, B / b / 1e9, c * 1e3, e * 1e3);
        }
    }
    return 0;
}
