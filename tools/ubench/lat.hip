// Latency calibration for the Zstd parser design (tools/, not product): what does one dependent memory round trip cost a
// single wave on MI355X when 2048 single-wave workgroups each own a private 768 KiB table + 4 MiB source?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void lat_kernel(uint32_t* __restrict__ tables, const uint8_t* __restrict__ src, unsigned long long* out, int iters, size_t table_words, size_t src_bytes) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t* tab = tables + (size_t)wg * table_words;
    const uint8_t* s = src + (size_t)wg * src_bytes;
    __shared__ volatile uint32_t lds[1024];
    unsigned long long t0, t1;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;
    // (A) dependent random table loads (one lane active -> pure latency)
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t idx = (x >> 8) % table_words; uint32_t v = tab[idx]; x ^= v; }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 0] = (t1 - t0) / iters;
    // (B) dependent random table loads, 16 lanes diverged addresses
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t idx = (x >> 8) % table_words; uint32_t v = lane < 16 ? tab[idx] : 0; x ^= v; }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 1] = (t1 - t0) / iters;
    // (C) sequential-ish source loads: 8 B per lane at consecutive byte offsets, advancing 24 B per iteration (L1/L2 hot)
    uint32_t ip = 0; uint64_t acc = 0;
    t0 = clock64();
    for (int i = 0; i < iters; i++) { uint64_t v; __builtin_memcpy(&v, s + ip + lane, 8); acc += v; ip += 24 + (uint32_t)(acc & 1); ip = __builtin_amdgcn_readfirstlane(ip); }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 2] = (t1 - t0) / iters;
    // (D) random store then wait (store round trip)
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t idx = (x >> 8) % table_words; if (lane < 16) tab[idx] = x; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 3] = (t1 - t0) / iters;
    // (E) LDS scoreboard: write lane id, read back, ballot
    uint32_t coll = 0;
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t sl = (x >> 10) & 1023; lds[sl] = lane; __builtin_amdgcn_wave_barrier(); coll += __ballot(lds[sl] != lane) != 0; }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 4] = (t1 - t0) / iters;
    // (F) random load from the 4 MiB source (candidate bytes), 16 lanes
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t off = (x >> 6) % (uint32_t)(src_bytes - 8); uint64_t v = 0; if (lane < 16) __builtin_memcpy(&v, s + off, 8); x ^= (uint32_t)v; }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 5] = (t1 - t0) / iters;
    // (H) random load from the LAST 256 KiB of source only (near candidates), 16 lanes
    t0 = clock64();
    for (int i = 0; i < iters; i++) { x = x * 1664525u + 1013904223u; uint32_t off = (x >> 6) % (uint32_t)((256u << 10) - 8); uint64_t v = 0; if (lane < 16) __builtin_memcpy(&v, s + off, 8); x ^= (uint32_t)v; }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 7] = (t1 - t0) / iters;
    // (G) clock64 overhead
    t0 = clock64();
    for (int i = 0; i < iters; i++) { acc += clock64(); }
    t1 = clock64();
    if (lane == 0) out[wg * 8 + 6] = (t1 - t0) / iters;
    if (x == 0x12345 && acc == 77 && coll == 99999) out[wg * 8 + 6] = x;
}

int main(int argc, char** argv) {
    int iters = 2000;
    const size_t table_words = 196608, src_bytes = 4u << 20;
    for (int nwg : {256, 512, 1024, 2048, 4096}) {
        uint32_t* tables; uint8_t* src; unsigned long long* out;
        CHK(hipMalloc(&tables, (size_t)nwg * table_words * 4)); CHK(hipMalloc(&src, (size_t)nwg * src_bytes)); CHK(hipMalloc(&out, nwg * 64));
        CHK(hipMemset(tables, 1, (size_t)nwg * table_words * 4)); CHK(hipMemset(src, 3, (size_t)nwg * src_bytes)); CHK(hipMemset(out, 0, nwg * 64));
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(lat_kernel, dim3(nwg), dim3(64), 0, 0, tables, src, out, iters, table_words, src_bytes);
            CHK(hipDeviceSynchronize());
        }
        std::vector<unsigned long long> h(nwg * 8);
        CHK(hipMemcpy(h.data(), out, nwg * 64, hipMemcpyDeviceToHost));
        const char* names[8] = {"A dep random table load (1 lane)", "B dep random table load (16 lanes)", "C sequential src 8B/lane", "D random store + vmcnt(0)", "E LDS scoreboard", "F random src load (16 lanes)", "G clock64", "H random src load in 256 KiB (16 lanes)"};
        printf("== %d workgroups (cycles per op, mean over waves)\n", nwg);
        for (int k = 0; k < 8; k++) { double m = 0; for (int w = 0; w < nwg; w++) m += h[w * 8 + k]; printf("  %-40s %8.0f\n", names[k], m / nwg); }
        hipFree(tables); hipFree(src); hipFree(out);
    }
    return 0;
}
