// Zstandard stages (level-3 frame compressor, frame decoder) — declarations for the C-ABI front end.
#pragma once
#include "tsx_internal.h"

struct tsx_zstd_consts;   // device-resident constant tables (defined in zstd_common.h)
size_t tsx_zstd_consts_bytes(void);
void tsx_zstd_build_consts(tsx_zstd_consts* host_out);
// Device workspace needed for a batch of n chunks of at most max_len bytes (compress or decompress).
size_t tsx_zstd_workspace_bytes(uint32_t n, uint32_t max_len);
// One Zstd frame per chunk (CompressionChunkEnumeration.java:50-63) is the work of the device's compressor service
// (tsx_internal.h: tsx_zseg, tsx_launch_zstd_service): chunk i of a member = src_base + descs[i].src_off, frame i written at
// mid + i * mid_stride, its size to zlen[i].  With fuse.crc set the wave first stores the CRC32C of its source chunk in
// descs[i].crc32c; with fuse.key set, the wave that wrote frame i also encrypts it (EncryptionChunkEnumeration.java:66-84) to
// fuse.out + descs[i].dst_off and sets descs[i].dst_len (0 and TSX_E_DST_TOO_SMALL when the slot is too small); with fuse.out but
// no key the frame itself goes to the slot.
// Inverse (DecompressionChunkEnumeration.java:39-46): frame i at (from_mid ? frames + i*mid_stride :
// frames + descs[i].src_off), length descs[i].src_len - (from_mid ? 28 : 0); output to dst + descs[i].dst_off,
// descs[i].dst_len set; status TSX_E_BAD_SIZE / TSX_E_BAD_FRAME / TSX_E_DST_TOO_SMALL on failure.
// skip (may be NULL): chunk i is left alone when skip[i * skip_stride] == 1 - it was decoded by the block-parallel form below.
// no_scratch (or a skip list): the build of the kernel that keeps everything in registers (4 waves per SIMD instead of 6) - the only one that
// may be launched while the compressor service's kernel is alive (zstd_dec.hip).
uint32_t tsx_launch_zstd_decompress(hipStream_t st, const tsx_zstd_consts* d_zc, const uint8_t* frames, int from_mid, uint64_t mid_stride,
                                    tsx_chunk_desc* d_descs, uint32_t n, uint8_t* dst, int32_t* d_status, void* d_work,
                                    const uint32_t* skip, uint32_t skip_stride, bool no_scratch);
// The same, one workgroup per BLOCK (zstd_dec_blocks.hip): for small batches, where a chunk's 32 blocks one after the other are all
// latency.  A fast path with a fallback: chunks it does not take (or gives up on) keep their skip word at 0 and must be decoded by
// tsx_launch_zstd_decompress behind it, with tsx_zstd_blockmode_skip() as the skip list.  bwork: tsx_zstd_blockmode_bytes(n, max_out).
size_t tsx_zstd_blockmode_bytes(uint32_t n, uint32_t max_out);
bool tsx_zstd_blockmode_takes(uint32_t max_out);
const uint32_t* tsx_zstd_blockmode_skip(const void* bwork, uint32_t* stride_words);
uint32_t tsx_launch_zstd_decompress_blocks(hipStream_t st, const uint8_t* frames, int from_mid, uint64_t mid_stride, tsx_chunk_desc* d_descs, uint32_t n,
                                           uint32_t max_out, uint8_t* dst, int32_t* d_status, void* bwork);
