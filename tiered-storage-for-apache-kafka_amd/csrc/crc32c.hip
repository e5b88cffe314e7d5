// CRC32C (Castagnoli) of every chunk of a batch — gfx950.
//
// Replaces (adds) the per-chunk checksum stage of SURVEY.md §8 a15; result must equal
// java.util.zip.CRC32C over the chunk (reflected poly 0x82F63B78, init/xorout 0xFFFFFFFF).
//
// HBM-bound design.  A chunk is cut into 256 KiB sub-blocks, one per 256-thread workgroup (4 MiB chunk
// = 16 workgroups, a 256-chunk segment = 4096 workgroups >> 256 CUs).  Inside a sub-block the bytes are
// read as ROWS of 256 x 16 B so that every wave issues fully coalesced global_load_dwordx4 (1 KiB per
// wave-instruction); thread t therefore owns the 16-byte pieces t, t+256, t+512, … and keeps a private
// 32-bit remainder that is advanced by one row (4096 bytes) per step:
//        A <- A * x^(8*4096)  xor  (piece * x^32)        (mod P, GF(2))
// The multiply-by-constant and the 16-byte reduction are slicing tables staged in LDS (20 KiB: 16 data
// tables + 4 row-stride tables).  After the last row each thread scales its remainder by
// x^(128 * pieces-to-the-end-of-the-sub-block) (bit-serial GF(2) multiply, once per sub-block), the
// workgroup XOR-reduces with wave shuffles, and a tiny second kernel stitches the sub-blocks and the
// <16-byte tail together.  Algorithmic traffic: N bytes read + 4 bytes written per chunk.
#include "crc_dev.h"

void tsx_crc_build_tables(tsx_crc_tables* t) {
    static uint32_t all[TSX_CRC_ROW_BYTES][256];   // slicing tables 0..4095 (4 MiB scratch, built once)
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = crc_mulx(c);
        all[0][i] = c;
    }
    for (int d = 1; d < TSX_CRC_ROW_BYTES; d++)
        for (uint32_t i = 0; i < 256; i++) all[d][i] = (all[d - 1][i] >> 8) ^ all[0][all[d - 1][i] & 0xFF];
    for (int d = 0; d < 16; d++)
        for (int i = 0; i < 256; i++) t->slice[d][i] = all[d][i];
    // state byte k (k = 0 is the low byte = first in stream order) followed by a whole row:
    // equivalent to a data byte with (ROW_BYTES - 1 - k) bytes after it.
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 256; i++) t->stride[k][i] = all[TSX_CRC_ROW_BYTES - 1 - k][i];
    uint32_t x128 = 0x80000000u;
    for (int i = 0; i < 128; i++) x128 = crc_mulx(x128);
    t->piece_pow[0] = 0x80000000u;
    for (int d = 1; d < 512; d++) t->piece_pow[d] = crc_mulmod(t->piece_pow[d - 1], x128);
    t->pow2[0] = x128;
    for (int k = 1; k < 32; k++) t->pow2[k] = crc_mulmod(t->pow2[k - 1], t->pow2[k - 1]);
}

__device__ static inline void crc_chunk_span(const tsx_chunk_desc* d, int use_dst_side, uint64_t* off, uint32_t* len) {
    if (use_dst_side) { *off = d->dst_off; *len = d->status == TSX_OK ? d->dst_len : 0; }
    else { *off = d->src_off; *len = d->src_len; }
}

__global__ __launch_bounds__(TSX_CRC_THREADS) void crc32c_partial_kernel(
        const tsx_crc_tables* __restrict__ tab, const uint8_t* __restrict__ src, const tsx_chunk_desc* __restrict__ descs,
        uint32_t max_sub, uint32_t* __restrict__ partials, int use_dst_side) {
    // 20 KiB = exactly 16 of the 1280-byte granules LDS is allocated in: 8 workgroups per CU.  (The fold's four words live in the tables'
    // first bytes once every lane is done with the tables: 16 bytes more cost a whole granule and the eighth workgroup.)
    __shared__ uint32_t lds_tab[20 * 256];
    static_assert(sizeof(lds_tab) == 16 * 1280, "LDS granules");
    const uint32_t t = threadIdx.x;
    const uint32_t chunk = blockIdx.x / max_sub, sub = blockIdx.x % max_sub;
    uint64_t off; uint32_t len;
    crc_chunk_span(&descs[chunk], use_dst_side, &off, &len);
    const uint32_t q = len >> 4;                           // whole 16-byte pieces in the chunk
    const uint32_t p0 = sub * TSX_CRC_SUB_PIECES;
    if (p0 >= q) {                                         // nothing for this workgroup (uniform exit)
        if (t == 0) partials[(size_t)chunk * max_sub + sub] = 0;
        return;
    }
    const uint32_t p1 = min(p0 + (uint32_t)TSX_CRC_SUB_PIECES, q);
    {   // stage the 20 tables: 16 KiB slice + 4 KiB stride, coalesced 16-byte loads
        const uint4* g = reinterpret_cast<const uint4*>(&tab->slice[0][0]);
        uint4* l = reinterpret_cast<uint4*>(lds_tab);
        for (uint32_t i = t; i < 20 * 256 / 4; i += TSX_CRC_THREADS) l[i] = g[i];
    }
    __syncthreads();
    const uint4* in = reinterpret_cast<const uint4*>(src + off);
    uint32_t acc = 0, last = 0;
    bool any = false;
    for (uint32_t p = p0 + t; p < p1; p += TSX_CRC_THREADS) {
        uint4 w = in[p];
        if (p == 0) w.x ^= 0xFFFFFFFFu;                    // CRC init folded into the first four bytes
        // advance the running remainder by one row
        uint32_t a = lds_tab[(16 + 0) * 256 + (acc & 0xFF)] ^ lds_tab[(16 + 1) * 256 + ((acc >> 8) & 0xFF)] ^
                     lds_tab[(16 + 2) * 256 + ((acc >> 16) & 0xFF)] ^ lds_tab[(16 + 3) * 256 + (acc >> 24)];
        // reduce the 16 new bytes: byte k has 15-k bytes after it
        a ^= lds_tab[15 * 256 + (w.x & 0xFF)] ^ lds_tab[14 * 256 + ((w.x >> 8) & 0xFF)] ^
             lds_tab[13 * 256 + ((w.x >> 16) & 0xFF)] ^ lds_tab[12 * 256 + (w.x >> 24)];
        a ^= lds_tab[11 * 256 + (w.y & 0xFF)] ^ lds_tab[10 * 256 + ((w.y >> 8) & 0xFF)] ^
             lds_tab[9 * 256 + ((w.y >> 16) & 0xFF)] ^ lds_tab[8 * 256 + (w.y >> 24)];
        a ^= lds_tab[7 * 256 + (w.z & 0xFF)] ^ lds_tab[6 * 256 + ((w.z >> 8) & 0xFF)] ^
             lds_tab[5 * 256 + ((w.z >> 16) & 0xFF)] ^ lds_tab[4 * 256 + (w.z >> 24)];
        a ^= lds_tab[3 * 256 + (w.w & 0xFF)] ^ lds_tab[2 * 256 + ((w.w >> 8) & 0xFF)] ^
             lds_tab[1 * 256 + ((w.w >> 16) & 0xFF)] ^ lds_tab[0 * 256 + (w.w >> 24)];
        acc = a; last = p; any = true;
    }
    // align every thread's remainder to the end of the sub-block's last piece and fold the workgroup
    uint32_t v = any ? crc_mulmod(acc, tab->piece_pow[p1 - 1 - last]) : 0u;
    for (int o = 32; o; o >>= 1) v ^= __shfl_xor(v, o);
    __syncthreads();                                       // every lane has made its last table lookup (crc_mulmod reads no table)
    if ((t & 63) == 0) lds_tab[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
        uint32_t r = 0;
        for (int w = 0; w < TSX_CRC_THREADS / 64; w++) r ^= lds_tab[w];
        partials[(size_t)chunk * max_sub + sub] = r;
    }
}

// One thread per chunk: Horner over the sub-block remainders, then the <16-byte tail, then xorout.
__global__ void crc32c_final_kernel(const tsx_crc_tables* __restrict__ tab, const uint8_t* __restrict__ src,
                                    tsx_chunk_desc* __restrict__ descs, uint32_t n, uint32_t max_sub,
                                    const uint32_t* __restrict__ partials, int use_dst_side, tsx_chunk_desc* __restrict__ mirror) {
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= n) return;
    uint64_t off; uint32_t len;
    crc_chunk_span(&descs[chunk], use_dst_side, &off, &len);
    const uint32_t q = len >> 4;
    uint32_t s = q ? 0u : 0xFFFFFFFFu;                     // init already folded in when a piece exists
    for (uint32_t sub = 0; sub * TSX_CRC_SUB_PIECES < q; sub++) {
        uint32_t cnt = min((uint32_t)TSX_CRC_SUB_PIECES, q - sub * TSX_CRC_SUB_PIECES);
        uint32_t mult = cnt == TSX_CRC_SUB_PIECES ? tab->pow2[14] : crc_pow_pieces(tab, cnt);  // 2^14 pieces = 256 KiB
        s = crc_mulmod(s, mult) ^ partials[(size_t)chunk * max_sub + sub];
    }
    const uint8_t* p = src + off + ((size_t)q << 4);
    for (uint32_t i = 0; i < (len & 15u); i++) {
        s ^= p[i];
        for (int k = 0; k < 8; k++) s = crc_mulx(s);
    }
    descs[chunk].crc32c = ~s;
    if (mirror) mirror[chunk].crc32c = ~s;                 // the caller's copy of the descriptor, in pinned host memory (tsx_api.hip, launch_stages)
}

void tsx_launch_crc32c(hipStream_t st, const tsx_crc_tables* d_tab, const uint8_t* src, tsx_chunk_desc* d_descs,
                       uint32_t n, uint32_t max_len, uint32_t* d_partials, int use_dst_side, tsx_chunk_desc* hd_mirror) {
    if (!n) return;
    uint32_t max_sub = (max_len + TSX_CRC_SUB_BYTES - 1) / TSX_CRC_SUB_BYTES;
    if (max_sub == 0) max_sub = 1;
    hipLaunchKernelGGL(crc32c_partial_kernel, dim3(n * max_sub), dim3(TSX_CRC_THREADS), 0, st, d_tab, src,
                       (const tsx_chunk_desc*)d_descs, max_sub, d_partials, use_dst_side);
    hipLaunchKernelGGL(crc32c_final_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_tab, src, d_descs, n, max_sub,
                       (const uint32_t*)d_partials, use_dst_side, hd_mirror);
}
