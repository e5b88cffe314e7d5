/*
 * ORACLE (test infrastructure only — never linked into the product library).
 *
 * CRC-32C (Castagnoli), the checksum java.util.zip.CRC32C computes: reflected
 * polynomial 0x82F63B78, init 0xFFFFFFFF, xorout 0xFFFFFFFF.
 *
 * The reference has no checksum stage in core/.../transform/ (SURVEY.md §0); the
 * only CRC32C on the path is kafka-clients' RecordBatch.ensureValid(), reached from
 *   core/src/main/java/io/aiven/kafka/tieredstorage/SegmentCompressionChecker.java:37-53
 * so the oracle for the additive per-chunk CRC stage is the
 * java.util.zip.CRC32C definition itself.  Pinned by the iSCSI KATs in
 * tests/test_oracle_crc32c.py: "123456789" -> E3069283, 32x00 -> 8A9136AA,
 * 32xFF -> 62A8AB43.
 */
#include <stddef.h>
#include <stdint.h>

static uint32_t g_tab[8][256];
static int g_init;

static void init_tables(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        g_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            g_tab[t][i] = (g_tab[t - 1][i] >> 8) ^ g_tab[0][g_tab[t - 1][i] & 0xFF];
    g_init = 1;
}

/* Bit-at-a-time definition; used by the tests to pin the table version. */
uint32_t orc_crc32c_bitwise(const uint8_t* p, size_t n) {
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
    }
    return ~c;
}

/* Streaming form: crc = orc_crc32c_update(0, p, n) == java CRC32C.update(p); getValue(). */
uint32_t orc_crc32c_update(uint32_t crc, const uint8_t* p, size_t n) {
    if (!g_init) init_tables();
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = (c >> 8) ^ g_tab[0][(c ^ *p++) & 0xFF]; n--; }
    while (n >= 8) {
        uint32_t lo = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        uint32_t hi = (uint32_t)p[4] | ((uint32_t)p[5] << 8) | ((uint32_t)p[6] << 16) | ((uint32_t)p[7] << 24);
        lo ^= c;
        c = g_tab[7][lo & 0xFF] ^ g_tab[6][(lo >> 8) & 0xFF] ^ g_tab[5][(lo >> 16) & 0xFF] ^ g_tab[4][lo >> 24] ^
            g_tab[3][hi & 0xFF] ^ g_tab[2][(hi >> 8) & 0xFF] ^ g_tab[1][(hi >> 16) & 0xFF] ^ g_tab[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ g_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

uint32_t orc_crc32c(const uint8_t* p, size_t n) { return orc_crc32c_update(0, p, n); }
