/*
 * ORACLE (test infrastructure only — never linked into the product library).
 *
 * AES-256-GCM exactly as the reference configures JCE:
 *   core/src/main/java/io/aiven/kafka/tieredstorage/security/AesEncryptionProvider.java:36-39
 *     KEY_SIZE = 256, "AES/GCM/NoPadding", GCM_TAG_LENGTH = 128
 *   AesEncryptionProvider.java:60-75   encryptionCipher(): init(ENCRYPT_MODE, key) + updateAAD(aad)
 *   AesEncryptionProvider.java:77-84   decryptionCipher(): GCMParameterSpec(128, chunk, 0, ivSize) + updateAAD
 *   core/.../manifest/SegmentEncryptionMetadataV1.java:30   IV_SIZE = 12
 * and the chunk layout  IV(12) || CIPHERTEXT(n) || TAG(16):
 *   core/.../transform/EncryptionChunkEnumeration.java:66-84
 *   core/.../transform/DecryptionChunkEnumeration.java:54-62
 *
 * The arithmetic itself lives in the JDK (SunJCE GaloisCounterMode), which is not under
 * /root/reference; this file restates NIST SP 800-38D (GCM) over FIPS-197 (AES).
 * Pinned by: FIPS-197 C.3 AES-256 KAT, the GCM spec's AES-256 test cases 13-16, and a
 * cross-check against OpenSSL 3.0 EVP aes-256-gcm (tests/test_oracle_aes_gcm.py).
 * The reference's own tests hold no AES known-answer vectors (IV/key/AAD random,
 * CT/transform/EncryptionChunkEnumerationTest.java:100-110 is a round trip).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

static uint8_t SBOX[256];
static int g_init;

static uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1B)); }

static uint8_t gmul(uint8_t a, uint8_t b) {
    uint8_t r = 0;
    while (b) { if (b & 1) r ^= a; a = xtime(a); b >>= 1; }
    return r;
}

static void init_sbox(void) {
    /* S(x) = affine(inverse(x)) over GF(2^8), FIPS-197 5.1.1 */
    for (int x = 0; x < 256; x++) {
        uint8_t inv = 0;
        if (x) for (int y = 1; y < 256; y++) if (gmul((uint8_t)x, (uint8_t)y) == 1) { inv = (uint8_t)y; break; }
        uint8_t s = inv, r = inv;
        for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        SBOX[x] = s ^ 0x63;
    }
    g_init = 1;
}

/* 15 round keys x 16 bytes, FIPS-197 5.2 (Nk = 8, Nr = 14). */
void orc_aes256_expand_key(const uint8_t key[32], uint8_t rk[240]) {
    if (!g_init) init_sbox();
    memcpy(rk, key, 32);
    uint8_t rcon = 1;
    for (int i = 8; i < 60; i++) {
        uint8_t t[4];
        memcpy(t, rk + 4 * (i - 1), 4);
        if (i % 8 == 0) {
            uint8_t t0 = t[0];
            t[0] = SBOX[t[1]] ^ rcon; t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[t0];
            rcon = xtime(rcon);
        } else if (i % 8 == 4) {
            for (int k = 0; k < 4; k++) t[k] = SBOX[t[k]];
        }
        for (int k = 0; k < 4; k++) rk[4 * i + k] = rk[4 * (i - 8) + k] ^ t[k];
    }
}

void orc_aes256_encrypt_block(const uint8_t rk[240], const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= 14; r++) {
        /* SubBytes + ShiftRows: state is column-major, s[4*c + row] */
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++)
                t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];
        if (r < 14) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c + 0] = (uint8_t)(xtime(a0) ^ (xtime(a1) ^ a1) ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ xtime(a1) ^ (xtime(a2) ^ a2) ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xtime(a2) ^ (xtime(a3) ^ a3));
                s[4 * c + 3] = (uint8_t)((xtime(a0) ^ a0) ^ a1 ^ a2 ^ xtime(a3));
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

/* GF(2^128) multiply, SP 800-38D 6.3 Algorithm 1 (bit 0 = MSB of byte 0, R = 0xE1 || 0^120). */
static void gf128_mul(uint8_t x[16], const uint8_t y[16]) {
    uint8_t z[16] = {0}, v[16];
    memcpy(v, y, 16);
    for (int i = 0; i < 128; i++) {
        if ((x[i >> 3] >> (7 - (i & 7))) & 1) for (int k = 0; k < 16; k++) z[k] ^= v[k];
        int lsb = v[15] & 1;
        for (int k = 15; k > 0; k--) v[k] = (uint8_t)((v[k] >> 1) | (v[k - 1] << 7));
        v[0] >>= 1;
        if (lsb) v[0] ^= 0xE1;
    }
    memcpy(x, z, 16);
}

/* 4-bit table version of "multiply by H" so 4 MiB chunks run in well under a second.
 * M[i] = (i as a 4-bit polynomial, MSB-first) * H; the test suite checks it against gf128_mul. */
typedef struct { uint8_t m[16][16]; } ghash_tab;

static void shift_right1(uint8_t v[16]) {
    int lsb = v[15] & 1;
    for (int k = 15; k > 0; k--) v[k] = (uint8_t)((v[k] >> 1) | (v[k - 1] << 7));
    v[0] >>= 1;
    if (lsb) v[0] ^= 0xE1;
}

static void ghash_tab_init(ghash_tab* t, const uint8_t h[16]) {
    memset(t, 0, sizeof *t);
    memcpy(t->m[8], h, 16);                              /* 1000b = x^0 * H */
    for (int i = 4; i >= 1; i >>= 1) { memcpy(t->m[i], t->m[i * 2], 16); shift_right1(t->m[i]); }
    for (int i = 2; i < 16; i <<= 1)
        for (int j = 1; j < i; j++)
            for (int k = 0; k < 16; k++) t->m[i + j][k] = t->m[i][k] ^ t->m[j][k];
}

static void ghash_tab_mul(const ghash_tab* t, uint8_t x[16]) {
    uint8_t z[16] = {0};
    for (int i = 15; i >= 0; i--) {
        for (int half = 0; half < 2; half++) {
            int nib = half == 0 ? (x[i] & 0xF) : (x[i] >> 4);
            if (!(i == 15 && half == 0)) {
                /* z = z * x^4 : four single-bit shifts with reduction */
                for (int s = 0; s < 4; s++) shift_right1(z);
            }
            for (int k = 0; k < 16; k++) z[k] ^= t->m[nib][k];
        }
    }
    memcpy(x, z, 16);
}

static void ghash_update(const ghash_tab* t, uint8_t y[16], const uint8_t* p, size_t n) {
    while (n >= 16) { for (int k = 0; k < 16; k++) y[k] ^= p[k]; ghash_tab_mul(t, y); p += 16; n -= 16; }
    if (n) { for (size_t k = 0; k < n; k++) y[k] ^= p[k]; ghash_tab_mul(t, y); }
}

void orc_gf128_mul(uint8_t x[16], const uint8_t y[16]) { gf128_mul(x, y); }
void orc_gf128_mul_tab(uint8_t x[16], const uint8_t h[16]) { ghash_tab t; ghash_tab_init(&t, h); ghash_tab_mul(&t, x); }

static void inc32(uint8_t cb[16]) {
    for (int k = 15; k >= 12; k--) if (++cb[k]) break;
}

/* Core: GCTR over `in` with ICB = inc32(J0), GHASH over aad || C, tag = GHASH ^ E(J0). */
static void gcm_core(const uint8_t key[32], const uint8_t iv[12], const uint8_t* aad, size_t aad_len,
                     const uint8_t* in, size_t n, uint8_t* out, int decrypt, uint8_t tag[16]) {
    uint8_t rk[240], h[16] = {0}, j0[16], cb[16], ks[16], y[16] = {0}, lenblk[16], ej0[16];
    ghash_tab t;
    orc_aes256_expand_key(key, rk);
    orc_aes256_encrypt_block(rk, h, h);
    ghash_tab_init(&t, h);
    memcpy(j0, iv, 12); j0[12] = 0; j0[13] = 0; j0[14] = 0; j0[15] = 1;   /* 96-bit IV: J0 = IV || 0^31 || 1 */
    memcpy(cb, j0, 16);
    ghash_update(&t, y, aad, aad_len);
    if (decrypt) ghash_update(&t, y, in, n);
    for (size_t off = 0; off < n; off += 16) {
        inc32(cb);
        orc_aes256_encrypt_block(rk, cb, ks);
        size_t m = n - off < 16 ? n - off : 16;
        for (size_t k = 0; k < m; k++) out[off + k] = in[off + k] ^ ks[k];
    }
    if (!decrypt) ghash_update(&t, y, out, n);
    uint64_t abits = (uint64_t)aad_len * 8, cbits = (uint64_t)n * 8;
    for (int k = 0; k < 8; k++) { lenblk[k] = (uint8_t)(abits >> (56 - 8 * k)); lenblk[8 + k] = (uint8_t)(cbits >> (56 - 8 * k)); }
    ghash_update(&t, y, lenblk, 16);
    orc_aes256_encrypt_block(rk, j0, ej0);
    for (int k = 0; k < 16; k++) tag[k] = y[k] ^ ej0[k];
}

/* EncryptionChunkEnumeration.nextElement (:66-80): out = IV || C || TAG, returns n + 28. */
size_t orc_gcm_encrypt_chunk(const uint8_t key[32], const uint8_t iv[12], const uint8_t* aad, size_t aad_len,
                             const uint8_t* pt, size_t n, uint8_t* out) {
    memcpy(out, iv, 12);
    gcm_core(key, iv, aad, aad_len, pt, n, out + 12, 0, out + 12 + n);
    return n + 28;
}

/* DecryptionChunkEnumeration.nextElement (:54-62): IV = first 12 bytes; returns plaintext length,
 * or -1 on tag mismatch (JCE: AEADBadTagException -> RuntimeException), -2 if the chunk is shorter
 * than IV + TAG. Plaintext is only released when the tag verifies, as JCE does. */
long orc_gcm_decrypt_chunk(const uint8_t key[32], const uint8_t* aad, size_t aad_len,
                           const uint8_t* chunk, size_t len, uint8_t* out) {
    if (len < 28) return -2;
    size_t n = len - 28;
    uint8_t tag[16];
    gcm_core(key, chunk, aad, aad_len, chunk + 12, n, out, 1, tag);
    uint8_t diff = 0;
    for (int k = 0; k < 16; k++) diff |= (uint8_t)(tag[k] ^ chunk[12 + n + k]);
    if (diff) { memset(out, 0, n); return -1; }
    return (long)n;
}

/* Raw CTR keystream XOR starting at counter block IV || BE32(ctr0): the "AES-256-CTR" of
 * BASELINE.json configs[2] is the GCTR half of GCM, i.e. ctr0 = 2. */
void orc_aes256_ctr(const uint8_t key[32], const uint8_t iv[12], uint32_t ctr0,
                    const uint8_t* in, size_t n, uint8_t* out) {
    uint8_t rk[240], cb[16], ks[16];
    orc_aes256_expand_key(key, rk);
    memcpy(cb, iv, 12);
    for (size_t off = 0; off < n; off += 16, ctr0++) {
        cb[12] = (uint8_t)(ctr0 >> 24); cb[13] = (uint8_t)(ctr0 >> 16); cb[14] = (uint8_t)(ctr0 >> 8); cb[15] = (uint8_t)ctr0;
        orc_aes256_encrypt_block(rk, cb, ks);
        size_t m = n - off < 16 ? n - off : 16;
        for (size_t k = 0; k < m; k++) out[off + k] = in[off + k] ^ ks[k];
    }
}
