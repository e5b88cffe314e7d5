// Cache-policy sweep for the table accesses (follow-up to mix2.hip): the same dependent read / rewrite / store mix, with the
// sc0 / sc1 / nt bits of the gfx950 global loads and stores varied through inline assembly.  Layouts: "word" (4-byte entries,
// 768 KiB per wave) and "line" (one 64-byte line per entry, 12 MiB per wave, full-line rewrites).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define DEFK(NAME, LDM, STM)                                                                                                      \
__global__ __launch_bounds__(64, 5) void NAME(uint32_t* __restrict__ base, size_t stride_words, uint32_t entries, unsigned long long* out, \
                                              int iters, int R, int W, int X, int line) {                                         \
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;                                                                           \
    uint32_t* tab = base + (size_t)wg * stride_words;                                                                             \
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;                                                                            \
    for (int i = 0; i < iters; i++) {                                                                                             \
        x = x * 1664525u + 1013904223u;                                                                                           \
        const uint32_t idx = (x >> 8) % entries;                                                                                  \
        uint32_t v = 0;                                                                                                           \
        if (line) {                                                                                                               \
            uint32_t* p = tab + (size_t)idx * 16;                                                                                 \
            if ((int)lane < R) { u32x4 a; asm volatile("global_load_dwordx4 %0, %1, off " LDM "\n s_waitcnt vmcnt(0)" : "=v"(a) : "v"(p) : "memory"); v = a.x + a.w; } \
            u32x4 s = {v + 1, x, (uint32_t)i, lane};                                                                              \
            if ((int)lane < W) asm volatile("global_store_dwordx4 %0, %1, off " STM "\n global_store_dwordx4 %0, %1, off offset:16 " STM "\n" \
                                            "global_store_dwordx4 %0, %1, off offset:32 " STM "\n global_store_dwordx4 %0, %1, off offset:48 " STM :: "v"(p), "v"(s) : "memory"); \
            if ((int)lane >= 32 && (int)lane < 32 + X) {                                                                          \
                uint32_t* q = tab + (size_t)((idx * 7 + 13) % entries) * 16;                                                      \
                asm volatile("global_store_dwordx4 %0, %1, off " STM "\n global_store_dwordx4 %0, %1, off offset:16 " STM "\n"    \
                             "global_store_dwordx4 %0, %1, off offset:32 " STM "\n global_store_dwordx4 %0, %1, off offset:48 " STM :: "v"(q), "v"(s) : "memory"); \
            }                                                                                                                     \
        } else {                                                                                                                  \
            uint32_t* p = tab + idx;                                                                                              \
            if ((int)lane < R) asm volatile("global_load_dword %0, %1, off " LDM "\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); \
            const uint32_t s = v + 1;                                                                                             \
            if ((int)lane < W) asm volatile("global_store_dword %0, %1, off " STM :: "v"(p), "v"(s) : "memory");                  \
            if ((int)lane >= 32 && (int)lane < 32 + X) { uint32_t* q = tab + (idx * 7 + 13) % entries; asm volatile("global_store_dword %0, %1, off " STM :: "v"(q), "v"(x) : "memory"); } \
        }                                                                                                                         \
        x ^= __shfl_xor(v, 1) + v;                                                                                                \
    }                                                                                                                             \
    if (x == 0x12345677u) out[wg] = 1;                                                                                            \
}
DEFK(k_00, "", "")
DEFK(k_n0, "nt", "")
DEFK(k_10, "sc1", "")
DEFK(k_s0, "sc0 sc1", "")
DEFK(k_a0, "sc0 sc1 nt", "")
DEFK(k_01, "", "sc1")
DEFK(k_0s, "", "sc0 sc1")
DEFK(k_n1, "nt", "sc1")
DEFK(k_ns, "nt", "sc0 sc1")
DEFK(k_ss, "sc0 sc1", "sc0 sc1")
DEFK(k_00b, "sc0", "sc0")
typedef void (*kfn)(uint32_t*, size_t, uint32_t, unsigned long long*, int, int, int, int, int);
int main(int argc, char** argv) {
    const uint32_t entries = 196608;
    const int iters = argc > 1 ? atoi(argv[1]) : 2500;
    const int maxwg = 5120;
    unsigned long long* out; CHK(hipMalloc(&out, maxwg * 8));
    uint32_t* base; size_t bytes = (size_t)maxwg * entries * 64;
    CHK(hipMalloc(&base, bytes)); CHK(hipMemset(base, 1, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct { kfn f; const char* name; } ks[] = {{k_00, "ld -        st -"}, {k_n0, "ld nt       st -"}, {k_10, "ld sc1      st -"}, {k_s0, "ld sc0sc1   st -"}, {k_a0, "ld sc0sc1nt st -"},
                                                {k_01, "ld -        st sc1"}, {k_0s, "ld -        st sc0sc1"}, {k_n1, "ld nt       st sc1"}, {k_ns, "ld nt       st sc0sc1"},
                                                {k_ss, "ld sc0sc1   st sc0sc1"}, {k_00b, "ld sc0      st sc0"}};
    struct { int R, W, X; const char* name; } mixes[] = {{18, 0, 0, "reads only"}, {18, 9, 4, "18r+9rw+4st"}, {12, 9, 4, "12r+9rw+4st"}, {0, 0, 24, "24 stores"}};
    for (int nwg : {5120, 2048})
        for (int line = 0; line < 2; line++)
            for (auto& kk : ks)
                for (auto& m : mixes) {
                    float ms = 0;
                    for (int rep = 0; rep < 2; rep++) {
                        CHK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(kk.f, dim3(nwg), dim3(64), 0, 0, base, line ? (size_t)entries * 16 : (size_t)entries, entries, out, iters, m.R, m.W, m.X, line);
                        CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    const double lines = (double)nwg * iters * (m.R + m.W + m.X);
                    printf("%5d waves %s  %-22s %-12s %8.2f ms -> %6.1f G lines/s\n", nwg, line ? "line" : "word", kk.name, m.name, ms, lines / ms / 1e6);
                    fflush(stdout);
                }
    return 0;
}
