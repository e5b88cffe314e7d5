// Does the latency of a random table probe depend on how far apart the 2048 per-chunk tables are (TLB reach)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void k(uint32_t* __restrict__ base, size_t stride_words, uint32_t table_words, unsigned long long* out, int iters, int lanes) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t* tab = base + (size_t)wg * stride_words;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t idx = (x >> 8) % table_words; uint32_t v = (int)lane < lanes ? tab[idx] : 0; x ^= v; }
    unsigned long long t1 = clock64();
    if (lane == 0) out[wg] = (t1 - t0) / iters;
    if (x == 0x12345677u) out[wg] = 1;
}
int main() {
    const uint32_t table_words = 196608;   // 768 KiB
    const int nwg = 2048, iters = 3000;
    size_t strides[] = {196608, 524288, 1048576, 2097152};   // words: 768 KiB, 2 MiB, 4 MiB, 8 MiB
    unsigned long long* out; CHK(hipMalloc(&out, nwg * 8));
    for (size_t sw : strides) {
        uint32_t* base; size_t bytes = (size_t)nwg * sw * 4;
        CHK(hipMalloc(&base, bytes)); CHK(hipMemset(base, 1, bytes));
        for (int lanes : {9, 18, 64}) {
            for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, base, sw, table_words, out, iters, lanes); CHK(hipDeviceSynchronize()); }
            std::vector<unsigned long long> h(nwg); CHK(hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost));
            double m = 0; for (auto v : h) m += v; m /= nwg;
            printf("stride %7.2f MiB (footprint %6.2f GiB) lanes %2d: %6.0f cycles/probe  -> %.1f G lines/s\n", sw * 4 / 1048576.0, bytes / 1073741824.0, lanes, m, nwg * (double)lanes / (m / 2.4e9) / 1e9);
        }
        CHK(hipFree(base));
    }
    return 0;
}
