// Temporary: Zstd stages not built yet — every chunk reports TSX_E_UNSUPPORTED.
#include "zstd_gpu.h"
struct tsx_zstd_consts { uint32_t dummy[4]; };
size_t tsx_zstd_consts_bytes(void) { return sizeof(tsx_zstd_consts); }
void tsx_zstd_build_consts(tsx_zstd_consts* h) { h->dummy[0] = 0; }
size_t tsx_zstd_workspace_bytes(uint32_t, uint32_t) { return 256; }
__global__ void zstd_unsupported_kernel(int32_t* status, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) status[i] = TSX_E_UNSUPPORTED;
}
uint32_t tsx_launch_zstd_compress(hipStream_t st, const tsx_zstd_consts*, const uint8_t*, const tsx_chunk_desc*, uint32_t n, uint32_t,
                                  uint8_t*, size_t, uint32_t*, int32_t* d_status, void*, uint32_t) {
    hipLaunchKernelGGL(zstd_unsupported_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_status, n);
    return 1;
}
uint32_t tsx_launch_zstd_decompress(hipStream_t st, const tsx_zstd_consts*, const uint8_t*, int, uint64_t, tsx_chunk_desc*, uint32_t n,
                                    uint8_t*, int32_t* d_status, void*) {
    hipLaunchKernelGGL(zstd_unsupported_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_status, n);
    return 1;
}
