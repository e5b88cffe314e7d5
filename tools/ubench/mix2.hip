// Follow-up to mix.hip: does the chip move more table lines per second when an insertion rewrites a WHOLE 64-byte line instead of
// 4 bytes of it?  Layouts: "word" = 4-byte entries, 768 KiB of tables per wave (the round-1 compressor); "line" = one 64-byte line
// per entry (index word + 60 bytes of the bytes around the position), 12 MiB per wave, so a probe brings the candidate's bytes
// with it and an insertion is a full-line write (no byte-masked partial write, no read-modify-write in the memory controller).
// Per step a wave reads R random entries, rewrites W of them and stores X more entries elsewhere; the next step's addresses depend
// on the loaded values.  Prints entries (= lines) per second and wall time.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT> __device__ inline void st16(uint4* p, uint4 v) { u32x4 t = {v.x, v.y, v.z, v.w}; if (NT) __builtin_nontemporal_store(t, (u32x4*)p); else *(u32x4*)p = t; }
template <int NT> __device__ inline void st4(uint32_t* p, uint32_t v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NT> __device__ inline uint4 ld16(const uint4* p) { u32x4 t = (NT & 2) ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p; return make_uint4(t.x, t.y, t.z, t.w); }
template <int NT> __device__ inline uint32_t ld4(const uint32_t* p) { return (NT & 2) ? __builtin_nontemporal_load(p) : *p; }
template <int LINE, int RD16, int NT>   // LINE: 0 word entries, 1 line entries;  RD16: 16-byte pieces read per probed line;  NT: 1 nontemporal stores, 2 loads too
__global__ __launch_bounds__(64, 5) void k(uint32_t* __restrict__ base, size_t stride_words, uint32_t entries, unsigned long long* out, int iters, int R, int W, int X) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t* tab = base + (size_t)wg * stride_words;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 8) % entries;
        uint32_t v = 0;
        if (LINE) {
            uint4* p = (uint4*)(tab + (size_t)idx * 16);
            if ((int)lane < R) {
                uint4 a = ld16<NT>(p); v = a.x + a.w;
                if (RD16 > 1) { uint4 b = ld16<NT>(p + 1); v += b.y; }
                if (RD16 > 2) { uint4 c = ld16<NT>(p + 2), d = ld16<NT>(p + 3); v += c.z + d.w; }
            }
            if ((int)lane < W) { uint4 s; s.x = v + 1; s.y = x; s.z = i; s.w = lane; st16<NT & 1>(p, s); st16<NT & 1>(p + 1, s); st16<NT & 1>(p + 2, s); st16<NT & 1>(p + 3, s); }
            if ((int)lane >= 32 && (int)lane < 32 + X) {
                uint4* q = (uint4*)(tab + (size_t)((idx * 7 + 13) % entries) * 16);
                uint4 s; s.x = x; s.y = x; s.z = i; s.w = lane; st16<NT & 1>(q, s); st16<NT & 1>(q + 1, s); st16<NT & 1>(q + 2, s); st16<NT & 1>(q + 3, s);
            }
        } else {
            v = (int)lane < R ? ld4<NT>(tab + idx) : 0;
            if ((int)lane < W) st4<NT & 1>(tab + idx, v + 1);
            if ((int)lane >= 32 && (int)lane < 32 + X) st4<NT & 1>(tab + (idx * 7 + 13) % entries, x);
        }
        x ^= __shfl_xor(v, 1) + v;
    }
    if (x == 0x12345677u) out[wg] = 1;
}
int main(int argc, char** argv) {
    const uint32_t entries = 196608;
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    const int maxwg = 5120;
    unsigned long long* out; CHK(hipMalloc(&out, maxwg * 8));
    uint32_t* base; size_t bytes = (size_t)maxwg * entries * 64;                  // 60 GiB for the line layout
    CHK(hipMalloc(&base, bytes)); CHK(hipMemset(base, 1, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    struct { int R, W, X; const char* name; } mixes[] = {{18, 0, 0, "reads only"}, {18, 9, 0, "18 reads + 9 rewrites"}, {18, 9, 4, "18 r + 9 rw + 4 st (parser mix)"},
                                                        {0, 0, 24, "stores only (24)"}, {12, 9, 4, "12 r + 9 rw + 4 st (lean mix)"}};
    for (int nwg : {2048, 5120})
        for (int layout = 0; layout < 7; layout++)
            for (auto& m : mixes) {
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    CHK(hipEventRecord(e0, 0));
                    #define L_(LN, RD, NTV, SW) hipLaunchKernelGGL((k<LN, RD, NTV>), dim3(nwg), dim3(64), 0, 0, base, (size_t)entries * SW, entries, out, iters, m.R, m.W, m.X)
                    switch (layout) { case 0: L_(0, 1, 0, 1); break; case 1: L_(0, 1, 1, 1); break; case 2: L_(0, 1, 3, 1); break; case 3: L_(1, 1, 0, 16); break;
                                      case 4: L_(1, 1, 1, 16); break; case 5: L_(1, 1, 3, 16); break; default: L_(1, 4, 1, 16); }
                    CHK(hipEventRecord(e1, 0)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
                }
                const char* ln[] = {"word", "word nt-st", "word nt-ld+st", "line r16", "line r16 nt-st", "line r16 nt-ld+st", "line r64 nt-st"};
                const double lines = (double)nwg * iters * (m.R + m.W + m.X);
                printf("%5d waves  %-26s %-34s %8.2f ms -> %6.1f G lines/s (reads %.1f + writes %.1f)  %.0f cycles/step@2.4GHz\n", nwg, ln[layout], m.name, ms, lines / ms / 1e6,
                       (double)nwg * iters * m.R / ms / 1e6, (double)nwg * iters * (m.W + m.X) / ms / 1e6, ms * 1e-3 * 2.4e9 / iters);
                fflush(stdout);
            }
    return 0;
}
