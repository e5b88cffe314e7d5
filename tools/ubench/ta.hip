// How long does one vector-memory instruction occupy a CU's address/L1 path, by active lane count and address pattern?
// 8 single-wave workgroups per CU issue back-to-back INDEPENDENT loads from a small (L2/L1-resident) table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int MODE>
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ tab, unsigned long long* out, int iters, int lanes) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    uint32_t x = lane * 2654435761u + wg * 40503u + 1, acc = 0;
    const uint32_t* t = tab + (wg & 255) * 4096;       // 16 KiB per wave slot
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            x = x * 1664525u + 1013904223u;
            uint32_t idx;
            if (MODE == 0) idx = (x >> 10) & 4095;                    // random dword per lane
            else if (MODE == 1) idx = ((x >> 10) & 4032) + lane;      // coalesced row
            else idx = (__builtin_amdgcn_readfirstlane(x) >> 10) & 4095;   // all lanes the same address
            if ((int)lane < lanes) acc += t[idx];
        }
    }
    unsigned long long t1 = clock64();
    if (lane == 0) out[wg] = (t1 - t0) * 100 / iters;
    if (acc == 0x12345677u) out[wg] = 1;
}
int main() {
    const int nwg = 2048, iters = 4000;
    uint32_t* tab; unsigned long long* out;
    CHK(hipMalloc(&tab, 256 * 4096 * 4)); CHK(hipMemset(tab, 1, 256 * 4096 * 4)); CHK(hipMalloc(&out, nwg * 8));
    const char* names[3] = {"random dword/lane", "coalesced row", "same address"};
    for (int mode = 0; mode < 3; mode++)
        for (int lanes : {1, 4, 9, 16, 32, 64}) {
            for (int rep = 0; rep < 2; rep++) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(64), 0, 0, tab, out, iters, lanes);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(64), 0, 0, tab, out, iters, lanes);
                else hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(64), 0, 0, tab, out, iters, lanes);
                CHK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> h(nwg); CHK(hipMemcpy(h.data(), out, nwg * 8, hipMemcpyDeviceToHost));
            double m = 0; for (auto v : h) m += v; m /= nwg * 100.0;
            printf("%-18s lanes %2d: %6.1f cycles per load per wave (8 waves/CU) -> %.1f cycles of CU time per instruction\n", names[mode], lanes, m, m / 8);
        }
    return 0;
}
