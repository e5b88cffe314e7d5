// Per-chunk bookkeeping of the block-parallel frame decoder (zstd_dec_blocks.hip): written by its index kernel, completed by its
// decode kernel, read by its execute kernel.  Lives in the context's block-mode workspace, one header per chunk, zeroed per call.
#pragma once
#include <stdint.h>

#define ZB_MAX_BLOCKS 264u                       /* blocks of one chunk this form takes: 2 x (16 MiB / 128 KiB) + slack; more -> fallback */
#define ZB_MAX_CHUNK (16u << 20)                 /* largest chunk (dst_cap) this form takes */

struct ZbBlock {
    uint32_t off, bsize;                         // block content in the frame: offset, Block_Size
    uint8_t btype, last, ltype, modes;           // block type, last-block bit, literals type, Symbol_Compression_Modes
    uint32_t litSize, q;                         // regenerated size of the literals, offset of the sequences section in the block
    uint32_t nbSeq;
    uint32_t litAt, seqAt;                       // this block's place in the literal arena (bytes) / the sequence arenas (entries)
    uint16_t hufSrc, tblSrc[3];                  // block whose bytes hold the Huffman tree / the LL, OF, ML table in force here
    uint32_t tOff[3], streamOff;                 // offsets in the block: description of each table defined here, start of the bit stream
    uint32_t regen;                              // bytes this block regenerates (raw / RLE: Block_Size; compressed: after decoding)
    uint32_t endHist[3];                         // repeat-offset history behind the block: offsets, or references into the incoming history
    uint32_t ok;                                 // decode kernel: sequences decoded and consistent
};
struct ZbChunk {
    uint32_t mode;                               // 1: the block form is decoding this chunk; 0: left to (or handed back to) zstd_decompress_kernel
    uint32_t nblocks, contentSize, pad;
    uint32_t live[32];                           // jump round r left unresolved words in this chunk (round r + 1 returns at once when not)
    ZbBlock blk[ZB_MAX_BLOCKS];
};
#define ZB_CHUNK_HDR_BYTES ((sizeof(ZbChunk) + 255u) & ~(size_t)255u)
