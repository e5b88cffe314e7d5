// AES-256-GCM over every chunk of a batch — gfx950.
//
// Replaces the JCE calls of
//   core/src/main/java/io/aiven/kafka/tieredstorage/transform/EncryptionChunkEnumeration.java:66-84
//       cipher.doFinal(chunk, 0, n, out, ivLen)  ->  out = IV(12) || C(n) || TAG(16)
//   core/.../transform/DecryptionChunkEnumeration.java:54-62
//       cipher.doFinal(chunk, ivSize, len - ivSize) (verify TAG, return plaintext)
// with the cipher parameters of core/.../security/AesEncryptionProvider.java:36-39,60-84
// ("AES/GCM/NoPadding", 256-bit key, 128-bit tag, 12-byte IV, AAD via updateAAD).
//
// Layout of the work.  A chunk is cut into 64 KiB sub-blocks (4096 AES blocks), one per 256-thread
// workgroup.  Thread t owns blocks t, t+256, … so that plaintext loads and ciphertext stores of a wave
// are contiguous 16 B per lane.  For each block the thread
//   * encrypts the counter block IV || BE32(2 + j) with T-table AES-256 (the T0 table is replicated
//     once per LDS bank — 32 copies, 32 KiB — so the 224 data-dependent lookups per block are
//     bank-conflict free; T1..T3 are rotations of T0; round keys sit in SGPRs),
//   * XORs, stores, and folds the ciphertext block into a private GHASH accumulator with stride 256:
//         Y <- Y * H^256  xor  C_j            (4-bit Shoup tables of H^256 in LDS, 8 KiB,
//                                              one conflict-free ds_read_b128 per nibble).
// After its last block the thread scales Y by H^(blocks to the end of the sub-block) (bit-serial
// multiply, once), the workgroup XOR-reduces, and a second small kernel (one wave per chunk) applies
// H^(distance to the end) to every sub-block, adds the AAD and length blocks, encrypts J0 and writes /
// verifies the tag.  GF(2^128) has no carry-less multiply on CDNA4: everything is shifts, XORs and
// table lookups.  Algorithmic traffic: n bytes read + n + 28 bytes written per chunk.
#include <mutex>
#include <string.h>
#include "tsx_internal.h"

// ---------------------------------------------------------------------------------------------------
// host: AES tables (FIPS-197 5.1.1: S(x) = affine(x^-1)); built with log/antilog tables over generator 3
// ---------------------------------------------------------------------------------------------------
void tsx_aes_build_tables(tsx_aes_tables* t) {
    uint8_t exp3[256], log3[256];
    uint8_t v = 1;
    for (int i = 0; i < 255; i++) {
        exp3[i] = v; log3[v] = (uint8_t)i;
        uint8_t v2 = (uint8_t)((v << 1) ^ ((v & 0x80) ? 0x1B : 0));
        v = (uint8_t)(v2 ^ v);                              // v * 3
    }
    exp3[255] = exp3[0];
    for (int x = 0; x < 256; x++) {
        uint8_t inv = x ? exp3[(255 - log3[x]) % 255] : 0;
        uint8_t s = inv;
        for (int k = 1; k <= 4; k++) s ^= (uint8_t)((inv << k) | (inv >> (8 - k)));
        s ^= 0x63;
        uint8_t s2 = (uint8_t)((s << 1) ^ ((s & 0x80) ? 0x1B : 0));
        uint8_t s3 = (uint8_t)(s2 ^ s);
        t->te0[x] = (uint32_t)s2 | ((uint32_t)s << 8) | ((uint32_t)s << 16) | ((uint32_t)s3 << 24);
    }
}

// ---------------------------------------------------------------------------------------------------
// per-key setup on the HOST: the arithmetic of gcm_setup_kernel below, word for word, in plain C++ (round keys, H = E_K(0),
// powers of H, the 4-bit tables of H^256 and the 2-bit tables of H^64).  ~0.1 ms of one core per batch key with the bit loop, ~10 us
// with the host's carry-less multiplier.
// ---------------------------------------------------------------------------------------------------
namespace {
struct HostAes { uint32_t te0[256]; std::once_flag once; };
HostAes g_host_aes;
inline uint32_t h_rotl(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
inline uint32_t h_sbx(uint32_t x) { return (g_host_aes.te0[x & 0xFF] >> 8) & 0xFFu; }
inline uint32_t h_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }
inline void h_mulx(tsx_gf128& v) {                                   // v * x in GF(2^128), bit 63 of hi = x^0
    const uint64_t carry = v.lo & 1;
    v.lo = (v.lo >> 1) | (v.hi << 63);
    v.hi >>= 1;
    if (carry) v.hi ^= 0xE100000000000000ull;
}
inline tsx_gf128 h_mul(const tsx_gf128& x, tsx_gf128 v) {
    tsx_gf128 z; z.hi = 0; z.lo = 0;
    for (int i = 0; i < 64; i++) { if ((x.hi >> (63 - i)) & 1) { z.hi ^= v.hi; z.lo ^= v.lo; } h_mulx(v); }
    for (int i = 0; i < 64; i++) { if ((x.lo >> (63 - i)) & 1) { z.hi ^= v.hi; z.lo ^= v.lo; } h_mulx(v); }
    return z;
}
// The same product with the host's carry-less multiplier (x86-64 PCLMULQDQ: every host a broker runs on has it; the bit loop above stays
// as the fallback and as the reference the self-test compares with).  With integer bit 127 - i holding the coefficient of x^i, the
// carry-less product C of the two integers is the bit-reversed polynomial product one place to the right: C << 1 = rev256(P).  Its high
// half H is rev(P mod x^128) and its low half L is rev(P div x^128), which folds in through x^128 = x^7 + x^2 + x + 1 (a right shift in
// this bit order) - the bits those right shifts drop are the second fold W.
#if defined(__x86_64__)
typedef long long tsx_v2di __attribute__((vector_size(16)));
__attribute__((target("pclmul,sse2"))) inline tsx_gf128 h_mul_clmul(const tsx_gf128& x, const tsx_gf128& y) {
    typedef unsigned __int128 u128;
    const tsx_v2di a = {(long long)x.lo, (long long)x.hi}, b = {(long long)y.lo, (long long)y.hi};
    auto prod = [](tsx_v2di v) { return ((u128)(uint64_t)v[1] << 64) | (uint64_t)v[0]; };
    const u128 p00 = prod(__builtin_ia32_pclmulqdq128(a, b, 0x00)), p11 = prod(__builtin_ia32_pclmulqdq128(a, b, 0x11));
    const u128 mid = prod(__builtin_ia32_pclmulqdq128(a, b, 0x10)) ^ prod(__builtin_ia32_pclmulqdq128(a, b, 0x01));
    u128 lo = p00 ^ (mid << 64), hi = p11 ^ (mid >> 64);              // C = hi : lo
    hi = (hi << 1) | (lo >> 127); lo <<= 1;                            // C << 1
    const u128 m = lo ^ (lo << 127) ^ (lo << 126) ^ (lo << 121);
    const u128 r = hi ^ m ^ (m >> 1) ^ (m >> 2) ^ (m >> 7);
    tsx_gf128 z; z.hi = (uint64_t)(r >> 64); z.lo = (uint64_t)r;
    return z;
}
inline bool h_have_clmul() { static const bool v = __builtin_cpu_supports("pclmul") && !getenv("TSX_NO_PCLMUL"); return v; }
#else
inline tsx_gf128 h_mul_clmul(const tsx_gf128& x, const tsx_gf128& y) { return h_mul(x, y); }
inline bool h_have_clmul() { return false; }
#endif
inline tsx_gf128 h_mulf(const tsx_gf128& x, const tsx_gf128& y) { return h_have_clmul() ? h_mul_clmul(x, y) : h_mul(x, y); }
// one AES-256 block through the Te0 table (little-endian state words, as aes256_encrypt in gcm_dev.h)
inline void h_aes256(const uint32_t* rk, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3) {
    const uint32_t* T = g_host_aes.te0;
    uint32_t s0 = w0 ^ rk[0], s1 = w1 ^ rk[1], s2 = w2 ^ rk[2], s3 = w3 ^ rk[3];
    auto col = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        return T[a & 0xFF] ^ h_rotl(T[(b >> 8) & 0xFF], 8) ^ h_rotl(T[(c >> 16) & 0xFF], 16) ^ h_rotl(T[(d >> 24) & 0xFF], 24);
    };
    for (int r = 1; r < 14; r++) {
        const uint32_t t0 = col(s0, s1, s2, s3) ^ rk[4 * r], t1 = col(s1, s2, s3, s0) ^ rk[4 * r + 1], t2 = col(s2, s3, s0, s1) ^ rk[4 * r + 2],
                       t3 = col(s3, s0, s1, s2) ^ rk[4 * r + 3];
        s0 = t0; s1 = t1; s2 = t2; s3 = t3;
    }
    auto last = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        return h_sbx(a) | (h_sbx(b >> 8) << 8) | (h_sbx(c >> 16) << 16) | (h_sbx(d >> 24) << 24);
    };
    w0 = last(s0, s1, s2, s3) ^ rk[56]; w1 = last(s1, s2, s3, s0) ^ rk[57]; w2 = last(s2, s3, s0, s1) ^ rk[58]; w3 = last(s3, s0, s1, s2) ^ rk[59];
}
}  // namespace

void tsx_gcm_key_build_host(const uint8_t key32[32], const uint8_t* aad, uint32_t aad_len, tsx_gcm_key* out) {
    std::call_once(g_host_aes.once, [] { tsx_aes_tables t; tsx_aes_build_tables(&t); memcpy(g_host_aes.te0, t.te0, sizeof g_host_aes.te0); });
    memset(out, 0, sizeof *out);
    uint32_t* rk = out->rk;
    for (int i = 0; i < 8; i++) rk[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
    uint32_t rcon = 1;
    for (int i = 8; i < 60; i++) {
        uint32_t tmp = rk[i - 1];
        if ((i & 7) == 0) {
            tmp = (tmp >> 8) | (tmp << 24);
            tmp = h_sbx(tmp) | (h_sbx(tmp >> 8) << 8) | (h_sbx(tmp >> 16) << 16) | (h_sbx(tmp >> 24) << 24);
            tmp ^= rcon;
            rcon = ((rcon << 1) ^ ((rcon & 0x80) ? 0x1B : 0)) & 0xFF;
        } else if ((i & 7) == 4) {
            tmp = h_sbx(tmp) | (h_sbx(tmp >> 8) << 8) | (h_sbx(tmp >> 16) << 16) | (h_sbx(tmp >> 24) << 24);
        }
        rk[i] = rk[i - 8] ^ tmp;
    }
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    h_aes256(rk, w0, w1, w2, w3);
    tsx_gf128 h; h.hi = ((uint64_t)h_bswap(w0) << 32) | h_bswap(w1); h.lo = ((uint64_t)h_bswap(w2) << 32) | h_bswap(w3);
    out->h = h; out->aad_len = aad_len;
    for (uint32_t t = 0; t < 64; t++) out->aad[t] = t < aad_len ? aad[t] : 0;
    out->hpow2[0] = h;
    for (int k = 1; k < 32; k++) out->hpow2[k] = h_mulf(out->hpow2[k - 1], out->hpow2[k - 1]);
    tsx_gf128 one; one.hi = 0x8000000000000000ull; one.lo = 0;
    out->hpow[0] = one; out->hpow[1] = h;
    for (int k = 1; k < 9; k++) { const uint32_t half = 1u << k; for (uint32_t d = 0; d < half; d++) out->hpow[half + d] = h_mulf(out->hpow[d], out->hpow2[k]); }
    tsx_gf128 sv[128];
    { tsx_gf128 v = out->hpow2[8]; for (int i = 0; i < 128; i++) { sv[i] = v; h_mulx(v); } }
    for (uint32_t e = 0; e < 512; e++) {                               // 4-bit (Shoup) tables of H^256
        const uint32_t p = e >> 4, val = e & 15;
        tsx_gf128 r; r.hi = 0; r.lo = 0;
        for (int b = 0; b < 4; b++) if ((val >> (3 - b)) & 1u) { r.hi ^= sv[4 * p + b].hi; r.lo ^= sv[4 * p + b].lo; }
        out->hstride_tab[p][val] = r;
    }
    {   // 2-bit tables of H^64: entry (p, val) = val.bit1 * H^64 x^(2p)  xor  val.bit0 * H^64 x^(2p+1)
        tsx_gf128 v = out->hpow[64];
        for (uint32_t p = 0; p < 64; p++) {
            tsx_gf128 a = v; h_mulx(v); tsx_gf128 b = v; h_mulx(v);
            for (uint32_t val = 0; val < 4; val++) {
                tsx_gf128 r; r.hi = 0; r.lo = 0;
                if (val & 2u) { r.hi ^= a.hi; r.lo ^= a.lo; }
                if (val & 1u) { r.hi ^= b.hi; r.lo ^= b.lo; }
                out->h64_tab[p][val] = r;
            }
        }
    }
}

// test hook: n pseudo-random products through the carry-less multiplier and through the bit loop; returns the number that differ
// (-1: this host has no carry-less multiplier, nothing compared)
extern "C" int tsx_debug_hmul_selftest(uint32_t n) {
    if (!h_have_clmul()) return -1;
    uint64_t s_ = 0x9E3779B97F4A7C15ull; int bad = 0;
    auto next = [&] { s_ ^= s_ << 13; s_ ^= s_ >> 7; s_ ^= s_ << 17; return s_; };
    for (uint32_t i = 0; i < n; i++) {
        tsx_gf128 a, b; a.hi = next(); a.lo = next(); b.hi = next(); b.lo = next();
        if (i % 7 == 0) { a.lo = 0; } if (i % 11 == 0) { b.hi = 0x8000000000000000ull; b.lo = 0; } if (i % 13 == 0) { a.hi = 0; a.lo = 1; }
        const tsx_gf128 p = h_mul(a, b), q = h_mul_clmul(a, b);
        if (p.hi != q.hi || p.lo != q.lo) bad++;
    }
    return bad;
}

#include "gcm_dev.h"

// ---------------------------------------------------------------------------------------------------
// per-key setup: round keys, H, powers of H, Shoup tables of H^256.  One 256-thread workgroup.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcm_setup_kernel(const tsx_aes_tables* __restrict__ aes, const uint8_t* __restrict__ key32,
                                                        const uint8_t* __restrict__ aad, uint32_t aad_len,
                                                        tsx_gcm_key* __restrict__ out) {
    __shared__ uint32_t s_rk[60];
    __shared__ tsx_gf128 s_v[128];
    __shared__ tsx_gf128 s_pow[512];
    __shared__ tsx_gf128 s_pow2[32];
    const uint32_t t = threadIdx.x;
    auto T0 = [&](uint32_t x) { return aes->te0[x]; };
    if (t == 0) {
        // FIPS-197 5.2 key expansion, Nk = 8
        for (int i = 0; i < 8; i++)
            s_rk[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
        uint32_t rcon = 1;
        for (int i = 8; i < 60; i++) {
            uint32_t tmp = s_rk[i - 1];
            #define SBX(x) ((aes->te0[(x) & 0xFF] >> 8) & 0xFFu)
            if ((i & 7) == 0) {
                tmp = (tmp >> 8) | (tmp << 24);                         // RotWord on little-endian packing
                tmp = SBX(tmp) | (SBX(tmp >> 8) << 8) | (SBX(tmp >> 16) << 16) | (SBX(tmp >> 24) << 24);
                tmp ^= rcon;
                rcon = ((rcon << 1) ^ ((rcon & 0x80) ? 0x1B : 0)) & 0xFF;
            } else if ((i & 7) == 4) {
                tmp = SBX(tmp) | (SBX(tmp >> 8) << 8) | (SBX(tmp >> 16) << 16) | (SBX(tmp >> 24) << 24);
            }
            #undef SBX
            s_rk[i] = s_rk[i - 8] ^ tmp;
        }
        uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
        aes256_encrypt(s_rk, T0, w0, w1, w2, w3);
        tsx_gf128 h = gf_from_le_words(w0, w1, w2, w3);
        s_pow2[0] = h;
        for (int k = 1; k < 32; k++) s_pow2[k] = gf_mul(s_pow2[k - 1], s_pow2[k - 1]);
        tsx_gf128 one; one.hi = 0x8000000000000000ull; one.lo = 0;
        s_pow[0] = one; s_pow[1] = h;
        tsx_gf128 v = s_pow2[8];                                        // H^256 * x^i, i = 0..127
        for (int i = 0; i < 128; i++) { s_v[i] = v; gf_mulx(v); }
        out->h = h;
        out->aad_len = aad_len;
    }
    __syncthreads();
    for (int k = 1; k < 9; k++) {                                       // hpow[2^k + d] = hpow[d] * H^(2^k)
        uint32_t half = 1u << k;
        for (uint32_t d = t; d < half; d += 256) s_pow[half + d] = gf_mul(s_pow[d], s_pow2[k]);
        __syncthreads();
    }
    if (t < 60) out->rk[t] = s_rk[t];
    if (t < 64) out->aad[t] = t < aad_len ? aad[t] : 0;
    if (t < 32) out->hpow2[t] = s_pow2[t];
    for (uint32_t d = t; d < 512; d += 256) out->hpow[d] = s_pow[d];
    {   // 2-bit tables of H^64: entry (p, val) = val.bit1 * H^64 x^(2p)  xor  val.bit0 * H^64 x^(2p+1)
        const uint32_t p = t >> 2, val = t & 3;
        tsx_gf128 v = s_pow[64];
        for (uint32_t i = 0; i < 2 * p; i++) gf_mulx(v);
        tsx_gf128 r; r.hi = 0; r.lo = 0;
        if (val & 2u) { r.hi ^= v.hi; r.lo ^= v.lo; }
        gf_mulx(v);
        if (val & 1u) { r.hi ^= v.hi; r.lo ^= v.lo; }
        out->h64_tab[p][val] = r;
    }
    for (uint32_t e = t; e < 512; e += 256) {                           // Shoup tables
        uint32_t p = e >> 4, val = e & 15;
        tsx_gf128 r; r.hi = 0; r.lo = 0;
        for (int b = 0; b < 4; b++)
            if ((val >> (3 - b)) & 1u) { r.hi ^= s_v[4 * p + b].hi; r.lo ^= s_v[4 * p + b].lo; }
        out->hstride_tab[p][val] = r;
    }
}

// ---------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TSX_GCM_THREADS) void gcm_ctr_ghash_kernel(
        const tsx_aes_tables* __restrict__ aes, const tsx_gcm_key* __restrict__ key, const tsx_gcm_chunk* __restrict__ chunks,
        uint32_t max_sub, const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t* __restrict__ partials, int decrypt) {
    __shared__ uint32_t lds_t0[256 * 32];          // T0 replicated per bank: entry x, copy b at [x*32 + b]
    __shared__ tsx_gf128 lds_hs[32 * 16];          // Shoup tables of H^256
    // 40 KiB = exactly 32 of the 1280-byte granules LDS is allocated in: 4 workgroups per CU (the registers allow 4 waves per SIMD).  The
    // fold's four values reuse the Shoup tables' first entries once every lane is done with them: 64 bytes more cost a granule and the
    // fourth workgroup.
    static_assert(sizeof(lds_t0) + sizeof(lds_hs) == 32 * 1280, "LDS granules");
    const uint32_t t = threadIdx.x;
    const uint32_t ci = blockIdx.x / max_sub, sub = blockIdx.x % max_sub;
    const tsx_gcm_chunk ch = chunks[ci];
    const uint32_t n = ch.len;
    const uint32_t nb = (n + 15) >> 4;                                  // ciphertext blocks incl. a partial one
    const uint32_t j0 = sub * TSX_GCM_SUB_BLOCKS;
    uint32_t* my_partial = partials + ((size_t)ci * max_sub + sub) * 4;
    if (ch.skip || j0 >= nb) {                                          // uniform exit
        if (t < 4) my_partial[t] = 0;
        return;
    }
    const uint32_t j1 = min(j0 + (uint32_t)TSX_GCM_SUB_BLOCKS, nb);
    for (uint32_t i = t; i < 256 * 32; i += TSX_GCM_THREADS) lds_t0[i] = aes->te0[i >> 5];
    for (uint32_t i = t; i < 512; i += TSX_GCM_THREADS) lds_hs[i] = (&key->hstride_tab[0][0])[i];
    __syncthreads();
    const uint32_t bank = t & 31;
    auto T0 = [&](uint32_t x) { return lds_t0[(x << 5) | bank]; };
    // encrypt: in = plaintext, out = IV||C||TAG.  decrypt: in = IV||C||TAG, out = plaintext.
    const uint8_t* src = in + ch.in_off + (decrypt ? 12 : 0);
    uint8_t* dst = out + ch.out_off + (decrypt ? 0 : 12);
    const uint8_t* ivp = decrypt ? in + ch.in_off : ch.iv;
    const uint32_t iv0 = (uint32_t)ivp[0] | ((uint32_t)ivp[1] << 8) | ((uint32_t)ivp[2] << 16) | ((uint32_t)ivp[3] << 24);
    const uint32_t iv1 = (uint32_t)ivp[4] | ((uint32_t)ivp[5] << 8) | ((uint32_t)ivp[6] << 16) | ((uint32_t)ivp[7] << 24);
    const uint32_t iv2 = (uint32_t)ivp[8] | ((uint32_t)ivp[9] << 8) | ((uint32_t)ivp[10] << 16) | ((uint32_t)ivp[11] << 24);
    tsx_gf128 y; y.hi = 0; y.lo = 0;
    uint32_t last = 0;
    bool any = false;
    for (uint32_t j = j0 + t; j < j1; j += TSX_GCM_THREADS) {
        uint32_t k0 = iv0, k1 = iv1, k2 = iv2, k3 = bswap32(2u + j);    // GCTR: first block uses inc32(J0) = IV||2
        aes256_encrypt(key->rk, T0, k0, k1, k2, k3);
        const uint32_t m = min(16u, n - (j << 4));
        uint32_t p[4];
        if (m == 16) {
            tsx_u128a4 v = *reinterpret_cast<const tsx_u128a4*>(src + ((size_t)j << 4));
            p[0] = v.v[0]; p[1] = v.v[1]; p[2] = v.v[2]; p[3] = v.v[3];
        } else {
            p[0] = p[1] = p[2] = p[3] = 0;
            for (uint32_t b = 0; b < m; b++) p[b >> 2] |= (uint32_t)src[((size_t)j << 4) + b] << (8 * (b & 3));
        }
        uint32_t c[4] = {p[0] ^ k0, p[1] ^ k1, p[2] ^ k2, p[3] ^ k3};
        if (m == 16) {
            tsx_u128a4 v; v.v[0] = c[0]; v.v[1] = c[1]; v.v[2] = c[2]; v.v[3] = c[3];
            *reinterpret_cast<tsx_u128a4*>(dst + ((size_t)j << 4)) = v;
        } else {
            for (uint32_t b = 0; b < m; b++) dst[((size_t)j << 4) + b] = (uint8_t)(c[b >> 2] >> (8 * (b & 3)));
            for (uint32_t b = m; b < 16; b++) c[b >> 2] &= ~(0xFFu << (8 * (b & 3)));   // GHASH sees zero padding
        }
        const uint32_t* gx = decrypt ? p : c;                           // GHASH always runs over the ciphertext
        // Y <- Y * H^256 xor X
        tsx_gf128 z; z.hi = 0; z.lo = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            tsx_gf128 e0 = lds_hs[q * 16 + ((y.hi >> (60 - 4 * q)) & 15u)];
            tsx_gf128 e1 = lds_hs[(16 + q) * 16 + ((y.lo >> (60 - 4 * q)) & 15u)];
            z.hi ^= e0.hi ^ e1.hi; z.lo ^= e0.lo ^ e1.lo;
        }
        tsx_gf128 x = gf_from_le_words(gx[0], gx[1], gx[2], gx[3]);
        y.hi = z.hi ^ x.hi; y.lo = z.lo ^ x.lo;
        last = j; any = true;
    }
    tsx_gf128 r; r.hi = 0; r.lo = 0;
    if (any) r = gf_mul(y, key->hpow[j1 - 1 - last]);
    for (int o = 32; o; o >>= 1) { r.hi ^= __shfl_xor(r.hi, o); r.lo ^= __shfl_xor(r.lo, o); }
    __syncthreads();                                                    // every lane has made its last lookup in lds_hs (gf_mul reads no table)
    if ((t & 63) == 0) lds_hs[t >> 6] = r;
    __syncthreads();
    if (t == 0) {
        tsx_gf128 s; s.hi = 0; s.lo = 0;
        for (int w = 0; w < TSX_GCM_THREADS / 64; w++) { s.hi ^= lds_hs[w].hi; s.lo ^= lds_hs[w].lo; }
        my_partial[0] = (uint32_t)s.hi; my_partial[1] = (uint32_t)(s.hi >> 32);
        my_partial[2] = (uint32_t)s.lo; my_partial[3] = (uint32_t)(s.lo >> 32);
    }
}

// H^e out of the key's tables: H^(e mod 512) is an entry of hpow[], every set bit above costs one multiplication by H^(2^k) - at most
// nine for a 4 MiB chunk (gf_pow_h of gcm_dev.h multiplies once per set bit of e: up to 17, each a 128-step bit loop).
__device__ static tsx_gf128 gf_pow_h_tab(const tsx_gcm_key* key, uint32_t e) {
    tsx_gf128 r = key->hpow[e & 511u];
    e >>= 9;
    for (int k = 9; e; k++, e >>= 1)
        if (e & 1u) r = gf_mul(r, key->hpow2[k]);
    return r;
}

// One wave per chunk: stitch the sub-block GHASH values, add AAD and length blocks, E_K(J0), tag.
// The terms of the sum are independent - sub-block s carries H^(blocks behind it + 2), the AAD's Horner value H^(nb + 1), the length block
// H - so the AAD and length terms are two more items of the same loop, on lanes of their own, instead of ~20 multiplications one after
// the other on lane 0 behind the reduction (a single-chunk fetch spent 93 us here: profiles/r06_dec_single_chunk_kernel_stats.txt).
__global__ __launch_bounds__(64) void gcm_final_kernel(const tsx_aes_tables* __restrict__ aes, const tsx_gcm_key* __restrict__ key,
                                                       const tsx_gcm_chunk* __restrict__ chunks, uint32_t max_sub,
                                                       const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       const uint32_t* __restrict__ partials, int32_t* __restrict__ status, int decrypt) {
    const uint32_t ci = blockIdx.x, lane = threadIdx.x;
    const tsx_gcm_chunk ch = chunks[ci];
    if (ch.skip) return;
    const uint32_t n = ch.len, nb = (n + 15) >> 4;
    const uint32_t nsub = (nb + (uint32_t)TSX_GCM_SUB_BLOCKS - 1) / (uint32_t)TSX_GCM_SUB_BLOCKS;
    // AAD blocks: Horner with H (every lane computes the same value: as long as one lane doing it), scaled below past the ciphertext and the length block
    const uint32_t alen = key->aad_len;
    tsx_gf128 a; a.hi = 0; a.lo = 0;
    for (uint32_t o = 0; o < alen; o += 16) {
        tsx_gf128 x = gf_from_bytes(key->aad + o, min(16u, alen - o));
        a.hi ^= x.hi; a.lo ^= x.lo;
        a = gf_mul(a, key->h);
    }
    tsx_gf128 acc; acc.hi = 0; acc.lo = 0;
    for (uint32_t it = lane; it < nsub + 2; it += 64) {
        tsx_gf128 z; uint32_t e;
        if (it < nsub) {
            const uint32_t* pp = partials + ((size_t)ci * max_sub + it) * 4;
            z.hi = ((uint64_t)pp[1] << 32) | pp[0]; z.lo = ((uint64_t)pp[3] << 32) | pp[2];
            const uint32_t jend = min((it + 1) * (uint32_t)TSX_GCM_SUB_BLOCKS, nb);
            e = nb - jend + 2;                                          // block j carries H^(nb-j+1): length block follows
        } else if (it == nsub) {
            z = a; e = nb + 1;                                          // (zero without AAD)
        } else {
            z.hi = (uint64_t)alen * 8; z.lo = (uint64_t)n * 8; e = 1;   // length block [len(A)]64 || [len(C)]64 in bits, times H
        }
        const tsx_gf128 r = gf_mul(z, gf_pow_h_tab(key, e));
        acc.hi ^= r.hi; acc.lo ^= r.lo;
    }
    for (int o = 32; o; o >>= 1) { acc.hi ^= __shfl_xor(acc.hi, o); acc.lo ^= __shfl_xor(acc.lo, o); }
    if (lane != 0) return;
    const uint8_t* ivp = decrypt ? in + ch.in_off : ch.iv;
    uint32_t k0 = (uint32_t)ivp[0] | ((uint32_t)ivp[1] << 8) | ((uint32_t)ivp[2] << 16) | ((uint32_t)ivp[3] << 24);
    uint32_t k1 = (uint32_t)ivp[4] | ((uint32_t)ivp[5] << 8) | ((uint32_t)ivp[6] << 16) | ((uint32_t)ivp[7] << 24);
    uint32_t k2 = (uint32_t)ivp[8] | ((uint32_t)ivp[9] << 8) | ((uint32_t)ivp[10] << 16) | ((uint32_t)ivp[11] << 24);
    uint32_t k3 = 0x01000000u;                                          // J0 = IV || 0^31 || 1
    auto T0 = [&](uint32_t x) { return aes->te0[x]; };
    aes256_encrypt(key->rk, T0, k0, k1, k2, k3);
    uint32_t tagw[4];
    gf_to_le_words(acc, tagw);
    tagw[0] ^= k0; tagw[1] ^= k1; tagw[2] ^= k2; tagw[3] ^= k3;
    if (!decrypt) {
        uint8_t* o = out + ch.out_off;
        for (int i = 0; i < 12; i++) o[i] = ch.iv[i];
        for (int i = 0; i < 16; i++) o[12 + n + i] = (uint8_t)(tagw[i >> 2] >> (8 * (i & 3)));
    } else {
        const uint8_t* tg = in + ch.in_off + 12 + n;
        uint32_t diff = 0;
        for (int i = 0; i < 16; i++) diff |= (uint32_t)(tg[i] ^ (uint8_t)(tagw[i >> 2] >> (8 * (i & 3))));
        if (diff) status[ci] = TSX_E_TAG_MISMATCH;
    }
}

void tsx_launch_gcm_setup(hipStream_t st, const tsx_aes_tables* d_aes, const uint8_t* d_key32, const uint8_t* d_aad,
                          uint32_t aad_len, tsx_gcm_key* d_key) {
    hipLaunchKernelGGL(gcm_setup_kernel, dim3(1), dim3(256), 0, st, d_aes, d_key32, d_aad, aad_len, d_key);
}

void tsx_launch_gcm(hipStream_t st, const tsx_aes_tables* d_aes, const tsx_gcm_key* d_key, const tsx_gcm_chunk* d_chunks,
                    uint32_t n, uint32_t max_len, const uint8_t* in, uint8_t* out, uint32_t* d_partials, int32_t* d_status,
                    int decrypt) {
    if (!n) return;
    uint32_t max_sub = (max_len + TSX_GCM_SUB_BYTES - 1) / TSX_GCM_SUB_BYTES;
    if (max_sub == 0) max_sub = 1;
    hipLaunchKernelGGL(gcm_ctr_ghash_kernel, dim3(n * max_sub), dim3(TSX_GCM_THREADS), 0, st, d_aes, d_key, d_chunks, max_sub,
                       in, out, d_partials, decrypt);
    hipLaunchKernelGGL(gcm_final_kernel, dim3(n), dim3(64), 0, st, d_aes, d_key, d_chunks, max_sub, in, out,
                       (const uint32_t*)d_partials, d_status, decrypt);
}
