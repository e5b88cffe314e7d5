// Shared definitions of the Zstandard stages (encoder + decoder): format constants and the per-chunk
// device workspace layout.
#pragma once
#include "tsx_internal.h"
#include "zstd_gpu.h"

#define ZS_BLOCK_MAX (128u << 10)
#define ZS_MAX_SEQ 32768u            /* 128 KiB / minMatch 4 */
#define ZS_MaxLL 35
#define ZS_MaxML 52
#define ZS_MaxOff 31
#define ZS_DefaultMaxOff 28
#define ZS_LLFSELog 9
#define ZS_MLFSELog 9
#define ZS_OffFSELog 8
#define ZS_LitHufLog 11
#define ZS_HUF_TABLELOG_MAX 12

struct tsx_zstd_consts { uint32_t abi; uint32_t pad[3]; };

struct zs_seq { uint32_t offBase, litLength, mlBase, litPos; };   /* litPos: chunk offset of the literal run (encoder) */

// ---- per-chunk workspace (global memory) --------------------------------------------------------------
#define ZS_WS_HASHLONG 0u                                             /* u32[1 << 17]                       */
#define ZS_WS_HASHSMALL (ZS_WS_HASHLONG + (4u << 17))                 /* u32[1 << 16]                       */
#define ZS_WS_SEQS (ZS_WS_HASHSMALL + (4u << 16))                     /* zs_seq[ZS_MAX_SEQ + 64]            */
#define ZS_WS_LIT (ZS_WS_SEQS + 16u * (ZS_MAX_SEQ + 64))              /* literals of the current block      */
#define ZS_WS_CODES (ZS_WS_LIT + ZS_BLOCK_MAX + 256)                  /* llCode | ofCode | mlCode           */
#define ZS_WS_CODE_STRIDE (ZS_MAX_SEQ + 64)
#define ZS_WS_STBITS (ZS_WS_CODES + 3u * ZS_WS_CODE_STRIDE)           /* u16[3][stride]: FSE state bits per sequence (LL|OF|ML) */
#define ZS_WS_BLOCKOUT (ZS_WS_STBITS + 6u * ZS_WS_CODE_STRIDE)        /* compressed block being built       */
#define ZS_BLOCKOUT_CAP (384u << 10)
#define ZS_WS_HUFSAVE (ZS_WS_BLOCKOUT + ZS_BLOCKOUT_CAP + 256)        /* the two Huffman tables of the literal stage, between blocks */
#define ZS_WS_KEYCOPY (ZS_WS_HUFSAVE + 2048)                           /* the wave's own copy of the batch key schedule (tsx_gcm_key), wiped by the wave */
#define ZS_WS_KEYCOPY_BYTES 21504u
#define ZS_WS_BYTES ((size_t)(ZS_WS_KEYCOPY + ZS_WS_KEYCOPY_BYTES))
#define ZS_WS_HASH_BYTES (ZS_WS_SEQS)                                 /* prefix that must be zero at start  */

struct zs_cparams { uint32_t windowLog, chainLog, hashLog, minMatch; };

// ZSTD_defaultCParameters[*][3] + ZSTD_adjustCParams_internal (libzstd 1.5.x), level 3 = dfast everywhere.
__host__ __device__ static inline zs_cparams zs_level3_cparams(uint32_t srcSize) {
    zs_cparams c;
    if (srcSize <= (16u << 10)) { c.windowLog = 14; c.chainLog = 14; c.hashLog = 15; c.minMatch = 4; }
    else if (srcSize <= (128u << 10)) { c.windowLog = 17; c.chainLog = 15; c.hashLog = 16; c.minMatch = 5; }
    else if (srcSize <= (256u << 10)) { c.windowLog = 18; c.chainLog = 16; c.hashLog = 16; c.minMatch = 4; }
    else { c.windowLog = 21; c.chainLog = 16; c.hashLog = 17; c.minMatch = 5; }
    uint32_t srcLog = 6;
    if (srcSize >= 64) { uint32_t v = srcSize - 1; srcLog = 0; while (v) { srcLog++; v >>= 1; } }
    if (c.windowLog > srcLog) c.windowLog = srcLog;
    if (c.hashLog > c.windowLog + 1) c.hashLog = c.windowLog + 1;
    if (c.chainLog > c.windowLog) c.chainLog = c.windowLog;
    if (c.windowLog < 10) c.windowLog = 10;
    return c;
}
