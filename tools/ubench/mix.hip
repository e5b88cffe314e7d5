// What does the chip sustain for the PARSER'S access mix?  Per step a wave reads R random 4-byte table entries (one 64-B line each),
// overwrites W of them (same lines: dirty in L2, written back later) and stores X more entries to other random lines (the
// complementary insertions) - the next step depends on the loaded values.  Per-chunk tables of 768 KiB as in the compressor;
// 2048 / 4096 / 5120 single-wave workgroups (1 / 2 / 2.5 batches resident).  Prints lines/s at the L2<->memory level by
// construction of the pattern (R reads + W + X write-backs per step, table >> caches) and wall time (hipEvents).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(64, 5) void k(uint32_t* __restrict__ base, size_t stride_words, uint32_t table_words, unsigned long long* out, int iters, int R, int W, int X) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t* tab = base + (size_t)wg * stride_words;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 8) % table_words;
        const uint32_t v = (int)lane < R ? tab[idx] : 0;
        if ((int)lane < W) tab[idx] = v + 1;
        if ((int)lane >= 32 && (int)lane < 32 + X) tab[(idx * 7 + 13) % table_words] = x;
        x ^= __shfl_xor(v, 1) + v;                                    // next addresses depend on the loaded values
    }
    if (x == 0x12345677u) out[wg] = 1;
}
int main() {
    const uint32_t table_words = 196608;   // 768 KiB
    const int iters = 4000;
    const int maxwg = 5120;
    unsigned long long* out; CHK(hipMalloc(&out, maxwg * 8));
    uint32_t* base; size_t bytes = (size_t)maxwg * table_words * 4;
    CHK(hipMalloc(&base, bytes)); CHK(hipMemset(base, 1, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct { int R, W, X; const char* name; } mixes[] = {{18, 0, 0, "reads only (18 lines/step)"}, {18, 9, 0, "18 reads + 9 rewrites"}, {18, 9, 4, "18 reads + 9 rewrites + 4 stores (parser mix)"},
                                                        {30, 14, 6, "30 reads + 14 rewrites + 6 stores"}, {64, 0, 0, "reads only (64 lines/step)"}, {0, 0, 24, "stores only (24 lines/step)"}};
    for (int nwg : {2048, 4096, 5120})
        for (auto& m : mixes) {
            float ms = 0;
            for (int rep = 0; rep < 2; rep++) {
                CHK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(nwg), dim3(64), 0, 0, base, (size_t)table_words, table_words, out, iters, m.R, m.W, m.X);
                CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double lines = (double)nwg * iters * (m.R + m.W + m.X);
            printf("%5d waves  %-48s %8.2f ms  -> %6.1f G lines/s (reads %.1f + write-backs %.1f)\n", nwg, m.name, ms, lines / ms / 1e6,
                   (double)nwg * iters * m.R / ms / 1e6, (double)nwg * iters * (m.W + m.X) / ms / 1e6);
        }
    return 0;
}
