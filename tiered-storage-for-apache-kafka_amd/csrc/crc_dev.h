// GF(2)[x]/P building blocks of CRC32C shared by the batch CRC kernels (crc32c.hip) and the one-wave-per-chunk CRC that the
// Zstd compressor's wave runs over its own source chunk (zstd_enc.hip) — gfx950.  Internal: not part of the C ABI.
#pragma once
#include "tsx_internal.h"

#define POLY 0x82F63B78u

// ---- GF(2)[x]/P helpers (reflected representation: bit 31 = x^0, shifting right multiplies by x) ----
__host__ __device__ static inline uint32_t crc_mulx(uint32_t v) { return (v >> 1) ^ (POLY & (0u - (v & 1u))); }

__host__ __device__ static inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t acc = 0;
    for (int i = 0; i < 32; i++) {
        acc ^= b & (0u - ((a >> (31 - i)) & 1u));
        b = crc_mulx(b);
    }
    return acc;
}

// x^(128*e) mod P by square-and-multiply over the pow2 table.
__device__ static inline uint32_t crc_pow_pieces(const tsx_crc_tables* tab, uint32_t e) {
    uint32_t r = 0x80000000u;
    for (int k = 0; e; k++, e >>= 1)
        if (e & 1u) r = crc_mulmod(r, tab->pow2[k]);
    return r;
}


// ---------------------------------------------------------------------------------------------------
// CRC32C of ONE buffer by ONE wave (java.util.zip.CRC32C semantics: init and xorout 0xFFFFFFFF).  Used where a wave already
// owns the buffer - the Zstd compressor checksums the chunk it is about to parse, so no separate launch has to find 20 KiB of
// LDS per workgroup on a chip full of compressor waves.  The whole 16-byte pieces are cut into 64 contiguous stripes, one per
// lane (each lane consumes whole 64-byte lines: four back-to-back 16-byte loads), reduced with slicing-by-4 (tables 0..3 of
// tsx_crc_tables, 4 KiB staged in `ldsTab`); every lane's remainder is then moved to the end of the last piece
// (x^(128 * pieces after the stripe), square-and-multiply), the wave XOR-reduces, lane 0 adds the < 16-byte tail.
// All 64 lanes must call (workgroup barrier inside); the result is returned in every lane.
// ---------------------------------------------------------------------------------------------------
__device__ static inline uint32_t crc32c_wave(const tsx_crc_tables* __restrict__ tab, const uint8_t* __restrict__ buf, uint32_t len,
                                              uint32_t* ldsTab, uint32_t lane) {
    {
        const uint4* g = reinterpret_cast<const uint4*>(&tab->slice[0][0]);
        uint4* l = reinterpret_cast<uint4*>(ldsTab);
        for (uint32_t i = lane; i < 4 * 256 / 4; i += 64) l[i] = g[i];
    }
    __syncthreads();
    const uint32_t q = len >> 4;
    const uint32_t per = (((q + 63) >> 6) + 3) & ~3u;                     // stripe = whole 64-byte lines
    const uint32_t p0 = min(lane * per, q), p1 = min(p0 + per, q);
    const uint4* in = reinterpret_cast<const uint4*>(buf);
    uint32_t s = 0;
    #define CRC_W(x) { s ^= (x); s = ldsTab[3 * 256 + (s & 0xFF)] ^ ldsTab[2 * 256 + ((s >> 8) & 0xFF)] ^ ldsTab[256 + ((s >> 16) & 0xFF)] ^ ldsTab[s >> 24]; }
    uint32_t p = p0;
    if (p < p1 && p == 0) {                                             // CRC init folded into the first four bytes
        const uint4 w = in[0];
        CRC_W(w.x ^ 0xFFFFFFFFu) CRC_W(w.y) CRC_W(w.z) CRC_W(w.w)
        p = 1;
    }
    for (; p + 4 <= p1; p += 4) {
        const uint4 w0 = in[p], w1 = in[p + 1], w2 = in[p + 2], w3 = in[p + 3];
        CRC_W(w0.x) CRC_W(w0.y) CRC_W(w0.z) CRC_W(w0.w)
        CRC_W(w1.x) CRC_W(w1.y) CRC_W(w1.z) CRC_W(w1.w)
        CRC_W(w2.x) CRC_W(w2.y) CRC_W(w2.z) CRC_W(w2.w)
        CRC_W(w3.x) CRC_W(w3.y) CRC_W(w3.z) CRC_W(w3.w)
    }
    for (; p < p1; p++) {
        const uint4 w = in[p];
        CRC_W(w.x) CRC_W(w.y) CRC_W(w.z) CRC_W(w.w)
    }
    #undef CRC_W
    uint32_t v = p1 > p0 ? crc_mulmod(s, crc_pow_pieces(tab, q - p1)) : 0u;
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    if (q == 0) v = 0xFFFFFFFFu;                                        // no piece carried the init
    const uint8_t* t = buf + ((size_t)q << 4);
    for (uint32_t i = 0; i < (len & 15u); i++) {                        // uniform: every lane repeats the short tail
        v ^= t[i];
        for (int k = 0; k < 8; k++) v = crc_mulx(v);
    }
    __syncthreads();                                                    // ldsTab may be reused by the caller
    return ~v;
}
