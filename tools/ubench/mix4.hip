// Does WHERE the tables live matter?  mix.hip (3.75 GiB allocation, tables packed at a 768 KiB stride) sustains 44 G lines/s for the
// parser mix, mix2.hip's word layout (same pattern inside a 60 GiB allocation) 37.  The compressor's tables sit inside the per-chunk
// workspace: 768 KiB of every 2.03 MiB (ZS_WS_BYTES), one allocation per context.  Variants: allocation size, stride, one allocation
// or three (three contexts in flight).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(64, 5) void k(uint32_t* __restrict__ b0, uint32_t* __restrict__ b1, uint32_t* __restrict__ b2, uint32_t per, size_t stride_words,
                                          uint32_t table_words, unsigned long long* out, int iters, int R, int W, int X) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t* base = wg / per == 0 ? b0 : wg / per == 1 ? b1 : b2;
    uint32_t* tab = base + (size_t)(wg % per) * stride_words;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 8) % table_words;
        const uint32_t v = (int)lane < R ? tab[idx] : 0;
        if ((int)lane < W) tab[idx] = v + 1;
        if ((int)lane >= 32 && (int)lane < 32 + X) tab[(idx * 7 + 13) % table_words] = x;
        x ^= __shfl_xor(v, 1) + v;
    }
    if (x == 0x12345677u) out[wg] = 1;
}
int main() {
    const uint32_t table_words = 196608;   // 768 KiB
    const int iters = 3000;
    unsigned long long* out; CHK(hipMalloc(&out, 8192 * 8));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct { const char* name; size_t stride; size_t alloc_gib; int nalloc; } cfg[] = {
        {"packed 768 KiB stride, exact allocation", 196608, 0, 1}, {"packed 768 KiB stride, 60 GiB allocation", 196608, 60, 1},
        {"workspace stride 2.03 MiB, exact allocation", 532608, 0, 1}, {"workspace stride 2.03 MiB, three allocations (3 contexts)", 532608, 0, 3},
        {"packed 768 KiB stride, three allocations", 196608, 0, 3}, {"stride 1 MiB (power of two), exact", 262144, 0, 1}};
    for (auto& c : cfg)
        for (int nwg : {2048, 5120, 6144}) {
            const uint32_t per = (uint32_t)((nwg + c.nalloc - 1) / c.nalloc);
            size_t bytes = c.alloc_gib ? (c.alloc_gib << 30) : (size_t)per * c.stride * 4;
            uint32_t* b[3] = {nullptr, nullptr, nullptr};
            for (int a = 0; a < c.nalloc; a++) { CHK(hipMalloc(&b[a], bytes)); CHK(hipMemset(b[a], 1, (size_t)per * c.stride * 4)); }
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, b[0], b[1] ? b[1] : b[0], b[2] ? b[2] : b[0], per, c.stride, table_words, out, iters, 18, 9, 4);
                CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("%5d waves  %-58s %8.2f ms -> %6.1f G lines/s (parser mix 18r + 9rw + 4st)\n", nwg, c.name, ms, (double)nwg * iters * 31 / ms / 1e6);
            fflush(stdout);
            for (int a = 0; a < c.nalloc; a++) CHK(hipFree(b[a]));
        }
    return 0;
}
