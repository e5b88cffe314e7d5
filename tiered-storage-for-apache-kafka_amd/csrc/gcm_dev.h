// Device-side AES-256 / GF(2^128) building blocks shared by the batch GCM kernels (gcm.hip) and the GCM tail that the
// Zstd compressor's wave runs over its own frame (zstd_enc.hip) — gfx950.  Internal: not part of the C ABI.
#pragma once
#include "tsx_internal.h"

// ---------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------
__device__ static inline uint32_t bswap32(uint32_t v) { return __byte_perm(v, 0, 0x0123); }
__device__ static inline uint32_t rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }

// AES-256 encryption of one block given as four little-endian column words.  `T0(x)` returns T0[x].
template <class Lookup>
__device__ static inline void aes256_encrypt(const uint32_t* __restrict__ rk, Lookup T0, uint32_t& w0, uint32_t& w1,
                                             uint32_t& w2, uint32_t& w3) {
    uint32_t s0 = w0 ^ rk[0], s1 = w1 ^ rk[1], s2 = w2 ^ rk[2], s3 = w3 ^ rk[3];
#pragma unroll
    for (int r = 1; r < 14; r++) {
        uint32_t t0 = T0(s0 & 0xFF) ^ rotl32(T0((s1 >> 8) & 0xFF), 8) ^ rotl32(T0((s2 >> 16) & 0xFF), 16) ^ rotl32(T0(s3 >> 24), 24) ^ rk[4 * r + 0];
        uint32_t t1 = T0(s1 & 0xFF) ^ rotl32(T0((s2 >> 8) & 0xFF), 8) ^ rotl32(T0((s3 >> 16) & 0xFF), 16) ^ rotl32(T0(s0 >> 24), 24) ^ rk[4 * r + 1];
        uint32_t t2 = T0(s2 & 0xFF) ^ rotl32(T0((s3 >> 8) & 0xFF), 8) ^ rotl32(T0((s0 >> 16) & 0xFF), 16) ^ rotl32(T0(s1 >> 24), 24) ^ rk[4 * r + 2];
        uint32_t t3 = T0(s3 & 0xFF) ^ rotl32(T0((s0 >> 8) & 0xFF), 8) ^ rotl32(T0((s1 >> 16) & 0xFF), 16) ^ rotl32(T0(s2 >> 24), 24) ^ rk[4 * r + 3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    // final round: SubBytes + ShiftRows only; S(x) is byte 1 of T0[x]
    #define SB(x) ((T0(x) >> 8) & 0xFFu)
    w0 = (SB(s0 & 0xFF) | (SB((s1 >> 8) & 0xFF) << 8) | (SB((s2 >> 16) & 0xFF) << 16) | (SB(s3 >> 24) << 24)) ^ rk[56];
    w1 = (SB(s1 & 0xFF) | (SB((s2 >> 8) & 0xFF) << 8) | (SB((s3 >> 16) & 0xFF) << 16) | (SB(s0 >> 24) << 24)) ^ rk[57];
    w2 = (SB(s2 & 0xFF) | (SB((s3 >> 8) & 0xFF) << 8) | (SB((s0 >> 16) & 0xFF) << 16) | (SB(s1 >> 24) << 24)) ^ rk[58];
    w3 = (SB(s3 & 0xFF) | (SB((s0 >> 8) & 0xFF) << 8) | (SB((s1 >> 16) & 0xFF) << 16) | (SB(s2 >> 24) << 24)) ^ rk[59];
    #undef SB
}

__device__ static inline tsx_gf128 gf_from_le_words(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    tsx_gf128 r;
    r.hi = ((uint64_t)bswap32(w0) << 32) | bswap32(w1);
    r.lo = ((uint64_t)bswap32(w2) << 32) | bswap32(w3);
    return r;
}
__device__ static inline void gf_to_le_words(const tsx_gf128& g, uint32_t w[4]) {
    w[0] = bswap32((uint32_t)(g.hi >> 32)); w[1] = bswap32((uint32_t)g.hi);
    w[2] = bswap32((uint32_t)(g.lo >> 32)); w[3] = bswap32((uint32_t)g.lo);
}
__device__ static inline tsx_gf128 gf_from_bytes(const uint8_t* p, uint32_t n) {   // zero padded
    uint8_t b[16];
    for (uint32_t i = 0; i < 16; i++) b[i] = i < n ? p[i] : 0;
    tsx_gf128 r; r.hi = 0; r.lo = 0;
    for (int i = 0; i < 8; i++) { r.hi = (r.hi << 8) | b[i]; r.lo = (r.lo << 8) | b[8 + i]; }
    return r;
}
// multiply by x: one step to the right in GCM bit order, reduction by R = 0xE1 || 0^120
__device__ static inline void gf_mulx(tsx_gf128& v) {
    uint64_t carry = v.lo & 1u;
    v.lo = (v.lo >> 1) | (v.hi << 63);
    v.hi = (v.hi >> 1) ^ (0xE100000000000000ull & (0ull - carry));
}
// generic bit-serial product (SP 800-38D Algorithm 1)
__device__ static tsx_gf128 gf_mul(const tsx_gf128& x, tsx_gf128 v) {
    tsx_gf128 z; z.hi = 0; z.lo = 0;
    for (int i = 0; i < 64; i++) {
        uint64_t m = 0ull - ((x.hi >> (63 - i)) & 1u);
        z.hi ^= v.hi & m; z.lo ^= v.lo & m;
        gf_mulx(v);
    }
    for (int i = 0; i < 64; i++) {
        uint64_t m = 0ull - ((x.lo >> (63 - i)) & 1u);
        z.hi ^= v.hi & m; z.lo ^= v.lo & m;
        gf_mulx(v);
    }
    return z;
}
__device__ static tsx_gf128 gf_pow_h(const tsx_gcm_key* key, uint32_t e) {
    tsx_gf128 r; r.hi = 0x8000000000000000ull; r.lo = 0;
    bool first = true;
    for (int k = 0; e; k++, e >>= 1) {
        if (!(e & 1u)) continue;
        if (first) { r = key->hpow2[k]; first = false; }
        else r = gf_mul(r, key->hpow2[k]);
    }
    return r;
}


// ---------------------------------------------------------------------------------------------------
// GCM-AE of ONE message by ONE wave (SP 800-38D 7.1, 96-bit IV): out = IV(12) || C(n) || TAG(16), the layout
// EncryptionChunkEnumeration.java:66-84 produces.  Used where a wave already owns the message - the Zstd compressor
// encrypts the frame it has just written, so no separate launch has to find free LDS on a chip full of compressor waves.
// Lane l owns blocks l, l + 64, ...; its GHASH accumulator advances by H^64 per block through 2-bit tables
// (key->h64_tab, 4 KiB, staged in LDS next to the 1 KiB T0 table); lanes are aligned with H^(distance to the end) and
// XOR-reduced; lane 0 adds AAD and length blocks and the tag.  `ldsT0`: 256 words, `ldsTab`: 256 entries; all 64 lanes
// must call (workgroup barrier inside).
// ---------------------------------------------------------------------------------------------------
__device__ static inline void gcm_encrypt_wave(const tsx_aes_tables* __restrict__ aes, const tsx_gcm_key* __restrict__ key,
                                               const uint8_t* __restrict__ ivp, const uint8_t* __restrict__ src, uint32_t n,
                                               uint8_t* __restrict__ out, uint32_t* ldsT0, tsx_gf128* ldsTab, uint32_t lane) {
    for (uint32_t i = lane; i < 256; i += 64) { ldsT0[i] = aes->te0[i]; ldsTab[i] = (&key->h64_tab[0][0])[i]; }
    __syncthreads();
    auto T0 = [&](uint32_t x) { return ldsT0[x]; };
    const uint32_t iv0 = (uint32_t)ivp[0] | ((uint32_t)ivp[1] << 8) | ((uint32_t)ivp[2] << 16) | ((uint32_t)ivp[3] << 24);
    const uint32_t iv1 = (uint32_t)ivp[4] | ((uint32_t)ivp[5] << 8) | ((uint32_t)ivp[6] << 16) | ((uint32_t)ivp[7] << 24);
    const uint32_t iv2 = (uint32_t)ivp[8] | ((uint32_t)ivp[9] << 8) | ((uint32_t)ivp[10] << 16) | ((uint32_t)ivp[11] << 24);
    const uint32_t nb = (n + 15) >> 4;
    uint8_t* dst = out + 12;
    tsx_gf128 y; y.hi = 0; y.lo = 0;
    uint32_t last = 0;
    bool any = false;
    for (uint32_t j = lane; j < nb; j += 64) {
        uint32_t k0 = iv0, k1 = iv1, k2 = iv2, k3 = bswap32(2u + j);
        aes256_encrypt(key->rk, T0, k0, k1, k2, k3);
        const uint32_t m = min(16u, n - (j << 4));
        uint32_t c[4];
        if (m == 16) {
            tsx_u128a4 v = *reinterpret_cast<const tsx_u128a4*>(src + ((size_t)j << 4));
            v.v[0] ^= k0; v.v[1] ^= k1; v.v[2] ^= k2; v.v[3] ^= k3;
            *reinterpret_cast<tsx_u128a4*>(dst + ((size_t)j << 4)) = v;
            c[0] = v.v[0]; c[1] = v.v[1]; c[2] = v.v[2]; c[3] = v.v[3];
        } else {
            c[0] = c[1] = c[2] = c[3] = 0;
            for (uint32_t b = 0; b < m; b++) c[b >> 2] |= (uint32_t)src[((size_t)j << 4) + b] << (8 * (b & 3));
            c[0] ^= k0; c[1] ^= k1; c[2] ^= k2; c[3] ^= k3;
            for (uint32_t b = 0; b < m; b++) dst[((size_t)j << 4) + b] = (uint8_t)(c[b >> 2] >> (8 * (b & 3)));
            for (uint32_t b = m; b < 16; b++) c[b >> 2] &= ~(0xFFu << (8 * (b & 3)));   // GHASH sees zero padding
        }
        tsx_gf128 z; z.hi = 0; z.lo = 0;                                // Y <- Y * H^64 xor C_j
#pragma unroll 8
        for (int q = 0; q < 32; q++) {
            const tsx_gf128 e0 = ldsTab[q * 4 + ((y.hi >> (62 - 2 * q)) & 3u)];
            const tsx_gf128 e1 = ldsTab[(32 + q) * 4 + ((y.lo >> (62 - 2 * q)) & 3u)];
            z.hi ^= e0.hi ^ e1.hi; z.lo ^= e0.lo ^ e1.lo;
        }
        const tsx_gf128 x = gf_from_le_words(c[0], c[1], c[2], c[3]);
        y.hi = z.hi ^ x.hi; y.lo = z.lo ^ x.lo;
        last = j; any = true;
    }
    tsx_gf128 acc; acc.hi = 0; acc.lo = 0;
    if (any) acc = gf_mul(y, key->hpow[nb - 1 - last + 2]);             // block j carries H^(nb-j+1): the length block follows
    for (int o = 32; o; o >>= 1) { acc.hi ^= __shfl_xor(acc.hi, o); acc.lo ^= __shfl_xor(acc.lo, o); }
    if (lane == 0) {
        const uint32_t alen = key->aad_len;
        if (alen) {                                                     // AAD blocks: Horner with H, then past C and the length block
            tsx_gf128 a; a.hi = 0; a.lo = 0;
            for (uint32_t o = 0; o < alen; o += 16) {
                const tsx_gf128 x = gf_from_bytes(key->aad + o, min(16u, alen - o));
                a.hi ^= x.hi; a.lo ^= x.lo;
                a = gf_mul(a, key->h);
            }
            const tsx_gf128 r = gf_mul(a, gf_pow_h(key, nb + 1));
            acc.hi ^= r.hi; acc.lo ^= r.lo;
        }
        tsx_gf128 l; l.hi = (uint64_t)alen * 8; l.lo = (uint64_t)n * 8; // [len(A)]64 || [len(C)]64 in bits, times H
        const tsx_gf128 r = gf_mul(l, key->h);
        acc.hi ^= r.hi; acc.lo ^= r.lo;
        uint32_t k0 = iv0, k1 = iv1, k2 = iv2, k3 = 0x01000000u;       // J0 = IV || 0^31 || 1
        aes256_encrypt(key->rk, T0, k0, k1, k2, k3);
        uint32_t tagw[4];
        gf_to_le_words(acc, tagw);
        tagw[0] ^= k0; tagw[1] ^= k1; tagw[2] ^= k2; tagw[3] ^= k3;
        for (int i = 0; i < 12; i++) out[i] = ivp[i];
        for (int i = 0; i < 16; i++) out[12 + n + i] = (uint8_t)(tagw[i >> 2] >> (8 * (i & 3)));
    }
}
