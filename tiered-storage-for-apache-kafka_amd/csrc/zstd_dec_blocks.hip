// Zstandard frame decoder, block-parallel form — gfx950.  For SMALL batches (a fetch: one chunk, or the few chunks of a prefetch
// window - DefaultChunkManager.java:50-70, ChunkCache.java:159-184), where zstd_dec.hip's one-workgroup-per-chunk pipeline is all
// latency: a 4 MiB chunk is 32 blocks one after the other, 25-50 ms however idle the chip is.  Here the unit of work is the BLOCK:
//
//   zb_index_kernel    one wave per chunk.  Frame header; lane 0 walks the block headers (a chain of <= 264 three-byte reads); then
//                      one lane per block parses the literals-section and sequences-section headers - sizes, table modes, where
//                      each table description and the bit stream start - and lane 0 assigns every block its place in the literal
//                      and sequence arenas and notes which earlier block holds the Huffman tree (treeless literals) and the FSE
//                      tables (Repeat mode) in force.
//   zb_decode_kernel   one workgroup (two waves) per block, all blocks of all chunks at once.  Wave 1 decodes the literals, wave 0
//                      the sequences - the same stages as zstd_dec.hip's (11-bit multi-symbol Huffman table, the three FSE state
//                      machines in lanes 0-2, field extraction on all lanes) - but a block that inherits a tree or a table rebuilds
//                      it from the earlier block's bytes, and repeat offsets are resolved SYMBOLICALLY: a block does not know the
//                      history it starts from, so an offset that comes out of the history is recorded as "incoming entry i minus d"
//                      and the block's outgoing history is a function of the incoming one.
//   zb_scatter_kernel  one workgroup (8 waves) per block.  Chains the block summaries up to its own block (output position = sum of the
//                      regenerated sizes before it, incoming history = composition of the outgoing ones: O(1) per block), shares the
//                      block's groups of 64 sequences out over its waves and writes ONE WORD PER OUTPUT BYTE: the byte itself for a
//                      literal, the position it copies from for a match byte.
//   zb_jump_kernel     <= log3(size) + 1 passes of pointer jumping (two jumps each, in place: three hops guaranteed) over those words: "where I copy from" becomes "where that copies from"
//                      until every word is a literal - the execution stage without any order between sequences, blocks or
//                      workgroups (in-order execution is ONE dependency chain through the whole chunk: see the comment there).
//   zb_emit_kernel     words -> bytes.
//
// This form is a fast path, not a second authority: anything it does not like (more than 264 blocks, a chunk above 16 MiB, a
// malformed frame, an offset out of range, a copy chain that does not end in a literal) clears the chunk's `mode` word and zstd_decompress_kernel - which
// is launched behind it with that word as its skip list - decodes the chunk and reports the error code.  Bytes are either final
// and correct or rewritten by the fallback.
#include "zstd_dec_dev.h"
#include "zstd_dec_blocks.h"
#ifdef ZB_DEBUG
#include <stdio.h>
#endif

#define ZB_SYM 0x80000000u                      /* a history-relative offset: ZB_SYM | entry << 28 | decrement */
#define ZB_SYM_ENTRY(v) (((v) >> 28) & 7u)
#define ZB_SYM_DEC(v) ((v) & 0x0FFFFFFFu)

#ifdef HIPEMU
#define ZB_LOAD_AGENT(p) (*(volatile const uint32_t*)(p))
#define ZB_STORE_AGENT(p, v) do { *(volatile uint32_t*)(p) = (v); } while (0)
#else
#define ZB_LOAD_AGENT(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ZB_STORE_AGENT(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

__device__ static inline uint32_t zb_sym_dec(uint32_t v) { return (v & ZB_SYM) ? v + 1 : v - 1; }     // "rep0 - 1" on either kind of value
__device__ static inline uint32_t zb_subst(uint32_t v, uint32_t h0, uint32_t h1, uint32_t h2) {        // a recorded offset given the incoming history
    if (!(v & ZB_SYM)) return v;
    const uint32_t e = ZB_SYM_ENTRY(v), r = e == 0 ? h0 : e == 1 ? h1 : h2;
    return r - ZB_SYM_DEC(v);                                           // 0 or wrapped when the frame is corrupt: caught where it is used
}

// literals-section header of a compressed block -> section size (0 = malformed)
struct ZbLit { uint32_t ltype, hl, litSize, csize, streams, section; };
__device__ static inline ZbLit zb_lit_header(const uint8_t* blk, uint32_t bsize) {
    ZbLit h; h.section = 0; h.csize = 0; h.streams = 1;
    const uint32_t b0 = blk[0], sf = (b0 >> 2) & 3;
    h.ltype = b0 & 3;
    if (h.ltype < 2) {
        if (sf == 0 || sf == 2) { h.litSize = b0 >> 3; h.hl = 1; }
        else if (sf == 1) { if (bsize < 2) return h; h.litSize = (b0 >> 4) + ((uint32_t)blk[1] << 4); h.hl = 2; }
        else { if (bsize < 3) return h; h.litSize = (b0 >> 4) + ((uint32_t)blk[1] << 4) + ((uint32_t)blk[2] << 12); h.hl = 3; }
        if (h.litSize > ZS_BLOCK_MAX) return h;
        const uint32_t sec = h.ltype == 0 ? h.hl + h.litSize : h.hl + 1;
        if (sec > bsize) return h;
        h.section = sec;
    } else {
        uint32_t bits;
        if (sf == 0) { h.hl = 3; bits = 10; h.streams = 1; }
        else if (sf == 1) { h.hl = 3; bits = 10; h.streams = 4; }
        else if (sf == 2) { h.hl = 4; bits = 14; h.streams = 4; }
        else { h.hl = 5; bits = 18; h.streams = 4; }
        if (h.hl > bsize) return h;
        uint64_t v = 0;
        for (uint32_t i = 0; i < h.hl; i++) v |= (uint64_t)blk[i] << (8 * i);
        h.litSize = (uint32_t)(v >> 4) & ((1u << bits) - 1);
        h.csize = (uint32_t)(v >> (4 + bits)) & ((1u << bits) - 1);
        if (h.litSize > ZS_BLOCK_MAX || h.hl + h.csize > bsize || h.litSize == 0) return h;
        h.section = h.hl + h.csize;
    }
    return h;
}

// ---------------------------------------------------------------------------------------------------
// index: one wave per chunk
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LANES) void zb_index_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                         const tsx_chunk_desc* __restrict__ descs, const int32_t* __restrict__ status,
                                                         uint8_t* __restrict__ hdrs, uint32_t lit_cap, uint32_t seq_cap) {
    __shared__ uint32_t sOff[ZB_MAX_BLOCKS], sSize[ZB_MAX_BLOCKS];
    __shared__ uint8_t sType[ZB_MAX_BLOCKS];
    __shared__ uint32_t sN, sBad;
    const uint32_t lane = threadIdx.x, chunk = blockIdx.x;
    ZbChunk* const C = (ZbChunk*)(hdrs + (size_t)chunk * ZB_CHUNK_HDR_BYTES);
    if (lane < 32) C->live[lane] = 0;
    if (lane == 0) C->mode = 0;                                       // "not taken" until the last line of this kernel says otherwise (no memset launch: nothing
                                                                      // else of the header is read before this kernel has written it)
    if (status[chunk] != TSX_OK) return;                              // nothing to decode, nothing to fall back to
    const tsx_chunk_desc d = descs[chunk];
    const uint8_t* __restrict__ src = from_mid ? frames + (uint64_t)chunk * mid_stride : frames + d.src_off;
    const uint32_t srcSize = from_mid ? (d.src_len >= 28 ? d.src_len - 28 : 0) : d.src_len;
    if (lane == 0) {
        // every early return below leaves mode = 0: the chunk-serial kernel decodes the chunk (and owns the error codes)
        uint32_t n = 0, bad = 1;
        do {
            if (srcSize < 6 || d.dst_cap > ZB_MAX_CHUNK) break;
            if (src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) break;
            const uint32_t fhd = src[4], single = (fhd >> 5) & 1, dictFlag = fhd & 3, fcsFlag = fhd >> 6;
            const bool hasChecksum = (fhd >> 2) & 1;
            if (fhd & 8) break;
            uint32_t p = 5;
            if (!single) { if (p >= srcSize || (src[p] >> 3) > 21) break; p++; }
            const uint32_t dl = dictFlag == 0 ? 0 : dictFlag == 1 ? 1 : dictFlag == 2 ? 2 : 4;
            if (p + dl > srcSize) break;
            uint32_t dictId = 0; for (uint32_t i = 0; i < dl; i++) dictId |= (uint32_t)src[p + i] << (8 * i);
            if (dictId) break;
            p += dl;
            const uint32_t fl = fcsFlag == 0 ? single : fcsFlag == 1 ? 2 : fcsFlag == 2 ? 4 : 8;
            if (fl == 0 || p + fl > srcSize) break;
            uint64_t contentSize = 0;
            for (uint32_t i = 0; i < fl; i++) contentSize |= (uint64_t)src[p + i] << (8 * i);
            if (fl == 2) contentSize += 256;
            p += fl;
            if (contentSize > d.dst_cap) break;
            C->contentSize = (uint32_t)contentSize;
            bool closed = false;
            while (n < ZB_MAX_BLOCKS) {                                 // the chain of block headers
                if (p + 3 > srcSize) break;
                const uint32_t bh = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
                p += 3;
                const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
                if (btype == 3) break;
                if (btype == 2 && (bsize > ZS_BLOCK_MAX || bsize < 2)) break;
                if (btype != 2 && bsize > ZS_BLOCK_MAX) break;         // a raw / RLE block regenerates Block_Size bytes: <= Block_Maximum_Size
                const uint32_t body = btype == 1 ? 1 : bsize;
                if (p + body > srcSize) break;
                sOff[n] = p; sSize[n] = bsize; sType[n] = (uint8_t)(btype | (last << 2));
                n++; p += body;
                if (last) { if (hasChecksum) { if (p + 4 > srcSize) break; p += 4; } closed = p == srcSize; break; }
            }
            if (closed) bad = 0;
        } while (0);
        sN = n; sBad = bad;
    }
    __syncthreads();
    const uint32_t n = sN;
    if (sBad) return;
    // ---- one lane per block: the section headers ----
    for (uint32_t b = lane; b < n; b += LANES) {
        ZbBlock B;
        B.off = sOff[b]; B.bsize = sSize[b]; B.btype = sType[b] & 3; B.last = sType[b] >> 2; B.ltype = 0; B.modes = 0;
        B.litSize = 0; B.q = 0; B.nbSeq = 0; B.litAt = 0; B.seqAt = 0; B.hufSrc = 0xFFFF; B.tblSrc[0] = B.tblSrc[1] = B.tblSrc[2] = 0xFFFF;
        B.tOff[0] = B.tOff[1] = B.tOff[2] = 0; B.streamOff = 0; B.regen = B.btype == 2 ? 0 : B.bsize; B.endHist[0] = B.endHist[1] = B.endHist[2] = 0; B.ok = 0;
        bool bad = false;
        if (B.btype == 2) {
            const uint8_t* const blk = src + B.off;
            const ZbLit h = zb_lit_header(blk, B.bsize);
            if (!h.section) bad = true;
            else {
                B.ltype = (uint8_t)h.ltype; B.litSize = h.litSize; B.q = h.section;
                uint32_t q = h.section;
                if (q >= B.bsize) bad = true;
                else {
                    uint32_t nbSeq = blk[q];
                    if (nbSeq < 128) q += 1;
                    else if (nbSeq < 255) { if (q + 2 > B.bsize) bad = true; else { nbSeq = ((nbSeq - 128) << 8) + blk[q + 1]; q += 2; } }
                    else { if (q + 3 > B.bsize) bad = true; else { nbSeq = blk[q + 1] + ((uint32_t)blk[q + 2] << 8) + 0x7F00; q += 3; } }
                    if (!bad && nbSeq > ZS_BLOCK_MAX / 3 + 1) bad = true;
                    B.nbSeq = nbSeq;
                    if (!bad && nbSeq == 0 && q != B.bsize) bad = true;
                    if (!bad && nbSeq) {
                        if (q >= B.bsize) bad = true;
                        else {
                            const uint32_t modes = blk[q];
                            uint32_t t = q + 1;
                            if (modes & 3) bad = true;
                            B.modes = (uint8_t)modes;
                            // (unrolled: B stays in registers - a dynamic index into B.tOff would put the whole struct into scratch, and no
                            //  kernel on the fetch path may use scratch: see zstd_dec.hip, zstd_decompress_fallback_kernel)
#pragma unroll
                            for (int k = 0; k < 3; k++) {
                                if (bad) continue;
                                const uint32_t mode = (modes >> (6 - 2 * k)) & 3;
                                const uint32_t maxSymK = k == 0 ? 35 : k == 1 ? 31 : 52, maxLogK = k == 0 ? 9 : k == 1 ? 8 : 9;
                                B.tOff[k] = t;
                                if (mode == 1) { if (t >= B.bsize) bad = true; t++; }
                                else if (mode == 2) {
                                    if (t >= B.bsize) { bad = true; continue; }
                                    const uint32_t used = fse_skipNCount(maxSymK, blk + t, B.bsize - t, maxLogK);
                                    if (!used) bad = true;
                                    t += used;
                                }
                            }
                            if (!bad && t >= B.bsize) bad = true;
                            B.streamOff = t;
                        }
                    }
                }
            }
        }
        if (bad) sBad = 1;                                             // (any lane: same value)
        C->blk[b] = B;
    }
    __threadfence_block();
    __syncthreads();
    if (sBad) return;
    // ---- lane 0: arenas, inheritance ----
    if (lane == 0) {
        uint32_t litAt = 0, seqAt = 0, huf = 0xFFFF, tb[3] = {0xFFFF, 0xFFFF, 0xFFFF};
        bool bad = false;
        for (uint32_t b = 0; b < n && !bad; b++) {
            ZbBlock* const B = &C->blk[b];
            if (B->btype != 2) continue;
            if (B->ltype == 2) huf = b;
            if (B->ltype == 3) { if (huf == 0xFFFF) bad = true; B->hufSrc = (uint16_t)huf; }
            if (B->ltype != 0) { B->litAt = litAt; litAt += (B->litSize + 64 + 15) & ~15u; }
            if (litAt > lit_cap) bad = true;
            if (B->nbSeq) {
                for (int k = 0; k < 3; k++) {
                    const uint32_t mode = (B->modes >> (6 - 2 * k)) & 3;
                    if (mode != 3) tb[k] = b; else if (tb[k] == 0xFFFF) bad = true;
                    B->tblSrc[k] = (uint16_t)tb[k];
                }
                B->seqAt = seqAt; seqAt += (B->nbSeq + 63) & ~63u;
                if (seqAt > seq_cap) bad = true;
            }
        }
        if (!bad) { C->nblocks = n; __threadfence(); C->mode = 1; }
    }
}

// ---------------------------------------------------------------------------------------------------
// decode: one workgroup (wave 0 sequences, wave 1 literals) per block
// ---------------------------------------------------------------------------------------------------
#define ZB_FAIL() do { if (lane == 0) ZB_STORE_AGENT(&C->mode, 0u); return; } while (0)
#ifdef TSX_PROF2
static unsigned long long* g_zbprof_out = nullptr;                    // 8 u64 per (chunk, block): phase laps of zb_decode_kernel (tools/zb_phase_laps.py)
extern "C" void tsx_debug_set_zbprof(void* dev_ptr) { g_zbprof_out = (unsigned long long*)dev_ptr; }
#define ZLT(k) do { const unsigned long long n_ = (unsigned long long)clock64(); zlt_[k] += n_ - zlast_; zlast_ = n_; } while (0)
#else
#define ZLT(k) do {} while (0)
#endif

// The 1 or 4 Huffman streams of a literals section, on lanes 0-3 through per-stream LDS windows (zstd_dec.hip's literal stage):
// payload = the section's bytes behind the tree description.  Returns false (wave-uniform) on a malformed stream.
__device__ static bool zb_huf_streams(DecLds& L, const uint8_t* __restrict__ pay, uint32_t payload, uint32_t streams, uint32_t litSize, uint8_t* __restrict__ lit, uint32_t lane) {
    uint32_t sOff[5], sCnt[4];
    if (streams == 1) { sOff[0] = 0; sOff[1] = payload; sOff[2] = sOff[3] = sOff[4] = payload; sCnt[0] = litSize; sCnt[1] = sCnt[2] = sCnt[3] = 0; }
    else {
        if (payload < 10) return false;
        const uint32_t s1 = pay[0] | (pay[1] << 8), s2 = pay[2] | (pay[3] << 8), s3 = pay[4] | (pay[5] << 8);
        if (6 + (uint64_t)s1 + s2 + s3 >= payload) return false;
        sOff[0] = 6; sOff[1] = 6 + s1; sOff[2] = sOff[1] + s2; sOff[3] = sOff[2] + s3; sOff[4] = payload;
        const uint32_t seg = (litSize + 3) / 4;
        if (3 * seg > litSize) return false;
        sCnt[0] = sCnt[1] = sCnt[2] = seg; sCnt[3] = litSize - 3 * seg;
    }
    bool ok = true;
    const bool mine = lane < streams;
    uint32_t o = 0; for (uint32_t k = 0; k < lane && k < 4; k++) o += mine ? sCnt[k] : 0;
    const uint32_t cnt = mine ? sCnt[lane] : 0, sn = mine ? sOff[lane + 1] - sOff[lane] : 0, sbeg = mine ? sOff[lane] : 0;
    uint8_t* const outp = lit + o;
    uint32_t hi = 0, Bh = 0; bool hdone = !mine;
    if (mine) {
        const uint32_t lastByte = sn ? pay[sbeg + sn - 1] : 0;
        if (lastByte == 0) { ok = false; hdone = true; }
        else Bh = 8 * (sn - 1) + dhb32(lastByte);
    }
    const uint32_t tableLog = L.hufLog, tmask = (1u << tableLog) - 1;
    for (;;) {
        const uint32_t myTop = hdone ? 0 : (Bh >> 3) + 8;
        const uint32_t myWb = myTop > ZS_HWIN ? (myTop - ZS_HWIN + 15) & ~15u : 0;
        for (uint32_t s_ = 0; s_ < streams; s_++) {
            const uint32_t top = (uint32_t)__builtin_amdgcn_readlane(myTop, (int)s_), wb = (uint32_t)__builtin_amdgcn_readlane(myWb, (int)s_), beg = (uint32_t)__builtin_amdgcn_readlane(sbeg, (int)s_), n_ = (uint32_t)__builtin_amdgcn_readlane(sn, (int)s_);
            const uint32_t k = lane * 16;
            if (wb + k < top) {
                uint4 v;
                if (wb + k + 16 <= n_) __builtin_memcpy(&v, pay + beg + wb + k, 16);
                else { uint8_t tmp[16]; for (uint32_t j = 0; j < 16; j++) tmp[j] = wb + k + j < n_ ? pay[beg + wb + k + j] : 0; __builtin_memcpy(&v, tmp, 16); }
                *reinterpret_cast<uint4*>(&L.hwin[s_ * (ZS_HWIN + 16) + k]) = v;
            }
        }
        __threadfence_block();
        WAVE_SYNC();
        if (!hdone) {
            const uint8_t* const win = &L.hwin[lane * (ZS_HWIN + 16)];
            while (hi + 16 <= cnt && Bh >= 56 && ((Bh - 56) >> 3) >= myWb) {
                const uint32_t lo = Bh - 56;
                const uint64_t c = wld64(win, myWb, lo >> 3) >> (lo & 7);
                uint32_t used = 0;
                #pragma unroll
                for (int k = 0; k < 5; k++) {
                    const uint32_t e = L.hufX[(uint32_t)(c >> (45 - used)) & 0x7FF];
                    const uint32_t sy = e & 0xFFFFFF;
                    __builtin_memcpy(outp + hi, &sy, 4);
                    hi += e >> 28; used += (e >> 24) & 15;
                }
                Bh -= used;
            }
            while (hi < cnt && (hi + 16 > cnt || Bh < 56)) {
                const uint32_t need = Bh < tableLog ? Bh : tableLog, lo = Bh - need;
                if ((lo >> 3) < myWb) break;
                const uint32_t bits = (uint32_t)(wld64(win, myWb, lo >> 3) >> (lo & 7)) & ((1u << need) - 1);
                const uint32_t e = huf_decode1(L, (bits << (tableLog - need)) & tmask, tableLog);
                if ((e >> 8) > Bh) { ok = false; hdone = true; break; }
                outp[hi++] = (uint8_t)e; Bh -= e >> 8;
            }
            if (!hdone && hi >= cnt) { if (Bh != 0) ok = false; hdone = true; }
        }
        WAVE_SYNC();
        if (__all(hdone)) break;
    }
    return !__any(!ok);
}

// One of the three sequence tables of a block from its description: mode 0 predefined, 1 RLE, 2 FSE-compressed (never 3 here: the
// caller has followed a Repeat back to the block that defines the table).  desc / avail: the description's bytes.
__device__ static bool zb_seq_table(DecLds& L, int k, uint32_t mode, const uint8_t* __restrict__ desc, uint32_t avail, uint32_t lane) {
    SeqD* const dt = k == 0 ? L.ll : k == 1 ? L.of : L.ml;
    uint32_t* const logp = k == 0 ? &L.llLog : k == 1 ? &L.ofLog : &L.mlLog;
    const uint32_t maxSymK = k == 0 ? 35 : k == 1 ? 31 : 52, maxLogK = k == 0 ? 9 : k == 1 ? 8 : 9;
    if (mode == 0) {
        const short* const dn = k == 0 ? dLLnorm : k == 1 ? dOFnorm : dMLnorm;
        const uint32_t dmax = k == 0 ? 35 : k == 1 ? 28 : 52, dlog = k == 1 ? 5 : 6;
        if (lane <= dmax) L.norm[lane] = dn[lane];
        if (lane == 0) *logp = dlog;
        __threadfence_block();
        WAVE_SYNC();
        return fse_buildSeqTable_wave(dt, L, dmax, dlog, k, lane);
    }
    if (mode == 1) {
        if (avail < 1) return false;
        const uint32_t sym = DUNI(desc[0]);
        if (sym > maxSymK) return false;
        if (lane == 0) { dt[0] = SEQD(0, 0, seq_ebits(L, sym, k), sym); *logp = 0; }
        __threadfence_block();
        WAVE_SYNC();
        return true;
    }
    if (avail < 1) return false;
    if (lane == 0) {
        uint32_t ms = maxSymK, tl = 0;
        const uint32_t used = fse_readNCount(L.norm, &ms, &tl, desc, avail, maxLogK);
        L.scal[0] = used; L.scal[3] = ms; L.scal[4] = tl;
        if (used) *logp = tl;
    }
    __threadfence_block();
    WAVE_SYNC();
    const uint32_t used = DUNI(L.scal[0]), ms = DUNI(L.scal[3]), tl = DUNI(L.scal[4]);
    WAVE_SYNC();
    return used && fse_buildSeqTable_wave(dt, L, ms, tl, k, lane);
}

__global__ __launch_bounds__(2 * LANES) void zb_decode_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                              const tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ hdrs, uint8_t* __restrict__ arenas,
                                                              uint64_t astride, uint32_t lit_cap, uint32_t seq_cap
#ifdef TSX_PROF2
                                                              , unsigned long long* __restrict__ zbprof
#endif
                                                              ) {
    __shared__ DecLds L;
    const uint32_t lane = threadIdx.x & (LANES - 1), role = DUNI(threadIdx.x >> 6), b = blockIdx.x, chunk = blockIdx.y;
#ifdef TSX_PROF2
    unsigned long long zlt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, zlast_ = (unsigned long long)clock64();
#endif
    ZbChunk* const C = (ZbChunk*)(hdrs + (size_t)chunk * ZB_CHUNK_HDR_BYTES);
    if (DUNI(C->mode) != 1 || b >= DUNI(C->nblocks)) return;
    ZbBlock* const B = &C->blk[b];
    if (DUNI(B->btype) != 2) return;
    const tsx_chunk_desc d = descs[chunk];
    const uint8_t* __restrict__ src = from_mid ? frames + (uint64_t)chunk * mid_stride : frames + d.src_off;
    uint8_t* const litArena = arenas + (size_t)chunk * astride;
    uint32_t* const seqArena = (uint32_t*)(litArena + lit_cap);
    const uint8_t* const blk = src + DUNI(B->off);
    const uint32_t bsize = DUNI(B->bsize);
    if (role == 1) {
        // ---- literals ----
        const ZbLit h = zb_lit_header(blk, bsize);                      // (validated by the index kernel)
        if (h.ltype == 0) return;                                       // raw literals are read in place
        uint8_t* const lit = litArena + DUNI(B->litAt);
        if (h.ltype == 1) { const uint8_t v = blk[h.hl]; for (uint32_t i = lane; i < h.litSize; i += LANES) lit[i] = v; return; }
        uint32_t t = h.hl;
        const uint8_t* tree = blk + h.hl; uint32_t treeAvail = h.csize;
        if (h.ltype == 3) {                                             // treeless: the tree of the latest block that carried one
            const ZbBlock* const S = &C->blk[DUNI(B->hufSrc)];
            const uint8_t* const sblk = src + DUNI(S->off);
            const ZbLit sh = zb_lit_header(sblk, DUNI(S->bsize));
            if (sh.ltype != 2 || !sh.section) ZB_FAIL();
            tree = sblk + sh.hl; treeAvail = sh.csize;
        }
        if (lane == 0) { L.hufValid = 0; L.scalH[0] = huf_readTable(L, tree, treeAvail); }
        __threadfence_block();
        WAVE_SYNC();
        const uint32_t used = DUNI(L.scalH[0]);
        WAVE_SYNC();
        if (!used) ZB_FAIL();
        huf_buildX_wave(L, lane);
        __threadfence_block();
        WAVE_SYNC();
        ZLT(6);
        if (h.ltype == 2) t += used;
        if (t > h.hl + h.csize) ZB_FAIL();
        if (!zb_huf_streams(L, blk + t, h.hl + h.csize - t, h.streams, h.litSize, lit, lane)) ZB_FAIL();
        ZLT(7);
#ifdef TSX_PROF2
        if (lane == 0 && zbprof) { unsigned long long* const zp = zbprof + ((size_t)chunk * ZB_MAX_BLOCKS + b) * 8; zp[6] = zlt_[6]; zp[7] = zlt_[7]; }
#endif
        return;
    }
    // ---- sequences ----
    const uint32_t nbSeq = DUNI(B->nbSeq), litSize = DUNI(B->litSize);
    if (nbSeq == 0) { if (lane == 0) { B->regen = litSize; B->endHist[0] = ZB_SYM; B->endHist[1] = ZB_SYM | (1u << 28); B->endHist[2] = ZB_SYM | (2u << 28); B->ok = 1; } return; }
    if (lane == 0) L.zeroEntry = 0;
    if (lane < 36) { L.cLLbase[lane] = dLLbase[lane]; L.cLLbits[lane] = dLLbits[lane]; }
    if (lane < 53) { L.cMLbase[lane] = dMLbase[lane]; L.cMLbits[lane] = dMLbits[lane]; }
    __threadfence_block();
    WAVE_SYNC();
    for (int k = 0; k < 3; k++) {
        const ZbBlock* const S = &C->blk[DUNI(B->tblSrc[k])];           // this block itself unless the table is a Repeat
        const uint32_t mode = (DUNI(S->modes) >> (6 - 2 * k)) & 3, to = DUNI(S->tOff[k]), sb = DUNI(S->bsize);
        if (mode == 3 || to > sb) ZB_FAIL();
        if (!zb_seq_table(L, k, mode, src + DUNI(S->off) + to, sb - to, lane)) ZB_FAIL();
    }
    ZLT(0);
    uint32_t* const sLL = seqArena + DUNI(B->seqAt); uint32_t* const sML = sLL + seq_cap; uint32_t* const sOF = sML + seq_cap;
    const uint32_t llLog = DUNI(L.llLog), ofLog = DUNI(L.ofLog), mlLog = DUNI(L.mlLog);
    const uint8_t* const win = L.swin + ZS_DPAD;
    const uint32_t t = DUNI(B->streamOff);
    const uint32_t n = bsize - t;                                       // >= 1 (index kernel)
    const uint8_t* const stream = blk + t;
    const uint32_t lastByte = DUNI(stream[n - 1]);
    if (lastByte == 0) ZB_FAIL();
    uint32_t Bc = 8 * (n - 1) + dhb32(lastByte), wbase = 0;             // bits of the stream not read yet
    const SeqD* const tbl = lane == 0 ? L.ll : lane == 1 ? L.ml : lane == 2 ? L.of : &L.zeroEntry;
    uint16_t* const recp = &L.rec[lane < 3 ? lane : 3];
    uint32_t st = 0;
    bool filled = false;
    uint32_t r0 = ZB_SYM, r1 = ZB_SYM | (1u << 28), r2 = ZB_SYM | (2u << 28);      // the history this block starts from, whatever it is
    uint32_t sumLL = 0, sumML = 0;
    for (uint32_t g = 0; g < nbSeq; g += LANES) {
        const uint32_t cnt = DUNI(nbSeq - g < LANES ? nbSeq - g : LANES);
        if (!filled || (wbase != 0 && (Bc >> 3) < wbase + 736)) {
            WAVE_SYNC();
            const uint32_t top = (Bc >> 3) + 8;
            wbase = top > ZS_DWIN ? (top - ZS_DWIN) & ~15u : 0;
            for (uint32_t k = lane * 16; wbase + k < top; k += LANES * 16) {
                uint4 v;
                if (wbase + k + 16 <= n) __builtin_memcpy(&v, stream + wbase + k, 16);
                else { uint8_t tmp[16]; for (uint32_t j = 0; j < 16; j++) tmp[j] = wbase + k + j < n ? stream[wbase + k + j] : 0; __builtin_memcpy(&v, tmp, 16); }
                *reinterpret_cast<uint4*>(&L.swin[ZS_DPAD + k]) = v;
            }
            if (lane < ZS_DPAD / 4) reinterpret_cast<uint32_t*>(L.swin)[lane] = 0;      // the margin in front of the window
            __threadfence_block();
            WAVE_SYNC();
            if (!filled) {
                filled = true;
                const uint32_t lo = Bc - (llLog + ofLog + mlLog);
                if ((int32_t)lo < 0) ZB_FAIL();
                const uint32_t w = DUNI((uint32_t)(wld64(win, wbase, lo >> 3) >> (lo & 7)));
                const uint32_t sm = w & ((1u << mlLog) - 1), so = (w >> mlLog) & ((1u << ofLog) - 1), sl = (w >> (mlLog + ofLog)) & ((1u << llLog) - 1);
                st = lane == 0 ? sl : lane == 1 ? sm : lane == 2 ? so : 0;
                Bc = lo;
            }
        }
        ZLT(1);
        // pass 1: the chain (lanes 0, 1, 2 = the LL, ML, OF state machines; see zstd_dec.hip)
        uint32_t bad = 0;
        const uint32_t Bgroup = Bc;
        const uint32_t upd = g + cnt < nbSeq ? cnt : cnt - 1;
        uint32_t j = 0;
        for (; j + 2 <= upd; j += 2) {                                    // two steps per trip: rec offsets become immediates, half the loop control
            seq_chain_step(tbl, recp + j * 4, win, wbase, st, Bc, bad);
            seq_chain_step(tbl, recp + j * 4 + 4, win, wbase, st, Bc, bad);
        }
        if (j < upd) seq_chain_step(tbl, recp + j * 4, win, wbase, st, Bc, bad);
        if (upd < cnt) {
            const uint32_t e_ = tbl[st];
            recp[upd * 4] = (uint16_t)st;
            const uint32_t eb = SEQD_EBITS(e_);
            const int32_t raw = (int32_t)(Bc - DUNI(eb + DPP_SHL(eb, 1) + DPP_SHL(eb, 2)));
            bad |= (uint32_t)raw;
            Bc = (uint32_t)(raw < 0 ? 0 : raw);
        }
        if (bad >> 31) ZB_FAIL();
        __threadfence_block();
        WAVE_SYNC();
        ZLT(2);
        // pass 2: every lane decodes the fields of its own sequence
        const bool valid = lane < cnt;
        uint32_t ll = 0, ml = 0, offBase = 4;
        {
            uint32_t el = 0, eo = 0, em = 0, mine = 0;
            if (valid) {
                uint64_t r; __builtin_memcpy(&r, &L.rec[lane * 4], 8);
                el = L.ll[(uint32_t)r & 0xFFFF]; em = L.ml[(uint32_t)(r >> 16) & 0xFFFF]; eo = L.of[(uint32_t)(r >> 32) & 0xFFFF];
                mine = SEQD_TOT(el) + SEQD_TOT(eo) + SEQD_TOT(em);
                if (g + lane + 1 == nbSeq) mine = SEQD_EBITS(el) + SEQD_EBITS(eo) + SEQD_EBITS(em);
            }
            uint32_t incl = mine;
            for (uint32_t o = 1; o < LANES; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (valid) {
                const uint32_t oc = SEQD_EBITS(eo), mbits = SEQD_EBITS(em), lbits = SEQD_EBITS(el);
                const uint32_t lbase = L.cLLbase[SEQD_SYM(el)], mbase = L.cMLbase[SEQD_SYM(em)];
                const uint32_t lo1 = Bgroup - (incl - mine) - oc;
                offBase = (1u << oc) + ((uint32_t)(wld64(win, wbase, lo1 >> 3) >> (lo1 & 7)) & ((1u << oc) - 1));
                const uint32_t lo2 = lo1 - mbits - lbits;
                const uint32_t w2 = (uint32_t)(wld64(win, wbase, lo2 >> 3) >> (lo2 & 7));
                ll = lbase + (w2 & ((1u << lbits) - 1));
                ml = mbase + ((w2 >> lbits) & ((1u << mbits) - 1));
            }
        }
        if (__any(valid && offBase > 3 && offBase - 3 >= ZB_SYM)) ZB_FAIL();     // an offset of 2 GiB or more: not in a frame this form takes
        ZLT(3);
        // pass 3: repeat offsets, on values that are either offsets or references into the incoming history
        uint32_t off = offBase - 3;
        {
            const unsigned long long ll0 = __ballot(valid && ll == 0);
            unsigned long long users = __ballot(valid && offBase <= 3);
            uint32_t prev = 0;
            for (;;) {
                const uint32_t j = users ? (uint32_t)__ffsll((long long)users) - 1 : cnt;
                const uint32_t gap = j - prev;
                const uint32_t a1 = __builtin_amdgcn_readlane(offBase, (int)(j >= 1 ? j - 1 : 0)) - 3;
                const uint32_t a2 = __builtin_amdgcn_readlane(offBase, (int)(j >= 2 ? j - 2 : 0)) - 3;
                const uint32_t a3 = __builtin_amdgcn_readlane(offBase, (int)(j >= 3 ? j - 3 : 0)) - 3;
                const uint32_t n2 = gap >= 3 ? a3 : gap == 2 ? r0 : gap == 1 ? r1 : r2;
                const uint32_t n1 = gap >= 2 ? a2 : gap == 1 ? r0 : r1;
                const uint32_t n0 = gap >= 1 ? a1 : r0;
                r0 = n0; r1 = n1; r2 = n2;
                if (!users) break;
                users &= users - 1;
                const uint32_t ob = __builtin_amdgcn_readlane(offBase, (int)j);
                const uint32_t idx = ob - 1 + (uint32_t)((ll0 >> j) & 1);
                const uint32_t c01 = idx == 0 ? r0 : r1, c23 = idx == 2 ? r2 : zb_sym_dec(r0);
                const uint32_t o_ = idx < 2 ? c01 : c23;
                r2 = idx >= 2 ? r1 : r2;
                r1 = idx >= 1 ? r0 : r1;
                r0 = o_;
                off = tsx_writelane(o_, j, off);
                prev = j + 1;
            }
        }
        ZLT(4);
        if (valid) { sLL[g + lane] = ll; sML[g + lane] = ml; sOF[g + lane] = off; }
        uint32_t a = ll, m = ml;
        for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); m += __shfl_xor(m, o); }
        sumLL += DUNI(a); sumML += DUNI(m);
        if (sumLL > litSize || litSize + sumML > ZS_BLOCK_MAX) ZB_FAIL();   // a block regenerates at most Block_Maximum_Size bytes
        ZLT(5);
    }
#ifdef TSX_PROF2
    if (lane == 0 && zbprof) { unsigned long long* const zp = zbprof + ((size_t)chunk * ZB_MAX_BLOCKS + b) * 8; for (int k = 0; k < 6; k++) zp[k] = zlt_[k]; }
#endif
    if (Bc != 0) ZB_FAIL();                                             // every bit of the stream was used
    if (lane == 0) { B->regen = litSize + sumML; B->endHist[0] = r0; B->endHist[1] = r1; B->endHist[2] = r2; B->ok = 1; }
}

// ---------------------------------------------------------------------------------------------------
// execution by pointer jumping: scatter -> ceil(log3(size)) + 1 jump passes -> emit
//
// Executing sequences in order is a dependency chain through the whole chunk: in log-like content every record copies its field
// names from the record before it (offset ~ one record), so byte p of record r is a copy of a copy ... of record 0 - the first
// sequences of a block read the last bytes of the block before it, and 32 blocks "side by side" still run one after the other
// (measured: 24 ms per chunk with cross-block waits against 33 ms for the chunk-serial kernel).  What breaks the chain is not
// order but TRANSITIVITY: give every output byte a word - the byte itself when it is a literal, else the position it copies
// from (p - offset, always < p) - and replace "where I copy from" by "where THAT copies from" until every word is a literal:
// pointer jumping, two jumps per pass = three hops guaranteed (zb_jump_kernel): ceil(log3(chain depth)) + 1 <= ceil(log3(size)) + 1
// passes, all bytes of all blocks at once, no ordering between workgroups at all (a word read while another thread replaces it
// holds either ancestor - both are valid - so the update is done in place).
// ---------------------------------------------------------------------------------------------------
#define ZB_LIT 0x80000000u                      /* src word: ZB_LIT | byte (resolved), else the chunk position this byte copies from */

// One workgroup of ZB_SC_WAVES waves per block.  Every wave walks the block summaries (positions, incoming repeat-offset history:
// O(1) per block, wave-uniform), the waves share out the block's groups of 64 sequences: a first sweep leaves every group's literal
// and output byte counts in LDS, their prefix sums place the groups, and then every group is scattered on its own - one word per
// output byte.  (One wave per block took 0.9 ms of a single chunk's 2.7 ms: ~85 groups one after the other, each behind its own
// loads - profiles/r03_dec_single_chunk_kernel_stats.txt.)
#define ZB_SC_WAVES 8u
#define ZB_SC_GROUPS ((ZS_BLOCK_MAX / 3u + LANES - 1) / LANES + 1)      /* a sequence regenerates >= 3 bytes, a block <= 128 KiB */
__global__ __launch_bounds__(ZB_SC_WAVES * LANES) void zb_scatter_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                                         tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ hdrs, uint8_t* __restrict__ arenas,
                                                                         uint64_t astride, uint32_t lit_cap, uint32_t seq_cap) {
    __shared__ uint32_t sRegen[ZB_MAX_BLOCKS];                          // regenerated size of every block
    __shared__ uint32_t sHist[3][ZB_MAX_BLOCKS];                        // outgoing history of every block (symbolic in its incoming one)
    __shared__ uint8_t sFlag[ZB_MAX_BLOCKS];                            // 1 compressed, 2 decoded fine, 4 has sequences
    __shared__ uint32_t sLit[ZB_SC_GROUPS + 1], sTot[ZB_SC_GROUPS + 1]; // per group of 64 sequences: literal / output bytes, then their exclusive prefixes
    __shared__ uint32_t gStart[ZB_SC_WAVES][LANES + 1], gLL[ZB_SC_WAVES][LANES], gLit[ZB_SC_WAVES][LANES], gSrc[ZB_SC_WAVES][LANES];   // the group a wave is scattering
    __shared__ uint32_t sFail;
    const uint32_t tid = threadIdx.x, lane = tid & (LANES - 1), wv = DUNI(tid >> 6), b = blockIdx.x, chunk = blockIdx.y;
    ZbChunk* const C = (ZbChunk*)(hdrs + (size_t)chunk * ZB_CHUNK_HDR_BYTES);
    if (DUNI(ZB_LOAD_AGENT(&C->mode)) != 1) return;
    const uint32_t nb = DUNI(C->nblocks);
    if (b >= nb) return;
    const tsx_chunk_desc d = descs[chunk];
    const uint8_t* __restrict__ src = from_mid ? frames + (uint64_t)chunk * mid_stride : frames + d.src_off;
    const uint8_t* const litArena = arenas + (size_t)chunk * astride;
    const uint32_t* const seqArena = (const uint32_t*)(litArena + lit_cap);
    uint32_t* const words = (uint32_t*)(litArena + lit_cap + 12u * (size_t)seq_cap);      // one per output byte
    for (uint32_t i = tid; i < nb; i += ZB_SC_WAVES * LANES) {
        const ZbBlock* const S = &C->blk[i];
        sRegen[i] = S->regen;
        sHist[0][i] = S->endHist[0]; sHist[1][i] = S->endHist[1]; sHist[2][i] = S->endHist[2];
        sFlag[i] = (uint8_t)((S->btype == 2 ? 1 : 0) | (S->ok ? 2 : 0) | (S->nbSeq ? 4 : 0));
    }
    if (tid == 0) sFail = 0;
    __threadfence_block();
    __syncthreads();
    uint32_t h0 = 1, h1 = 4, h2 = 8, myStart = 0, regen = 0;
    {
        uint32_t pos = 0; bool okAll = true;
        for (uint32_t i = 0; i < nb; i++) {                             // wave-uniform; O(1) per block; every wave for itself
            const uint32_t rg = DUNI(sRegen[i]), fl = DUNI(sFlag[i]);
            if (i == b) { myStart = pos; regen = rg; }
            if (fl & 1) {
                if (!(fl & 2)) okAll = false;
                if (i < b && (fl & 4)) {
                    const uint32_t e0 = DUNI(sHist[0][i]), e1 = DUNI(sHist[1][i]), e2 = DUNI(sHist[2][i]);
                    const uint32_t n0 = zb_subst(e0, h0, h1, h2), n1 = zb_subst(e1, h0, h1, h2), n2 = zb_subst(e2, h0, h1, h2);
                    h0 = n0; h1 = n1; h2 = n2;
                }
            }
            pos += rg;
            if (pos > ZB_MAX_CHUNK) { okAll = false; break; }
        }
        if (!okAll || pos != DUNI(C->contentSize)) { if (tid == 0) ZB_STORE_AGENT(&C->mode, 0u); return; }       // every wave of the chunk sees the same sums
    }
    const ZbBlock* const B = &C->blk[b];
    const uint32_t btype = DUNI(B->btype), boff = DUNI(B->off);
    const uint32_t contentSize = DUNI(C->contentSize);
    if (b == 0 && tid == 0) descs[chunk].dst_len = contentSize;
    if (btype == 0) { for (uint32_t i = tid; i < regen; i += ZB_SC_WAVES * LANES) words[myStart + i] = ZB_LIT | src[boff + i]; return; }
    if (btype == 1) { const uint32_t v = ZB_LIT | src[boff]; for (uint32_t i = tid; i < regen; i += ZB_SC_WAVES * LANES) words[myStart + i] = v; return; }
    const uint32_t litSize = DUNI(B->litSize), nbSeq = DUNI(B->nbSeq);
    const uint8_t* litPtr = litArena + DUNI(B->litAt);
    if (DUNI(B->ltype) == 0) { const ZbLit h = zb_lit_header(src + boff, DUNI(B->bsize)); litPtr = src + boff + h.hl; }
    const uint32_t* const sLL = seqArena + DUNI(B->seqAt); const uint32_t* const sML = sLL + seq_cap; const uint32_t* const sOF = sML + seq_cap;
    const uint32_t ngroups = (nbSeq + LANES - 1) / LANES;
    if (ngroups > ZB_SC_GROUPS) { if (tid == 0) ZB_STORE_AGENT(&C->mode, 0u); return; }   // (the decode kernel bounds the match lengths' sum: cannot happen)
    // sweep 1: the groups' byte counts
    for (uint32_t gi = wv; gi < ngroups; gi += ZB_SC_WAVES) {
        const uint32_t q = gi * LANES + lane;
        uint32_t a = q < nbSeq ? sLL[q] : 0, t = q < nbSeq ? a + sML[q] : 0;
        for (int o = 32; o; o >>= 1) { a += __shfl_xor(a, o); t += __shfl_xor(t, o); }
        if (lane == 0) { sLit[gi] = a; sTot[gi] = t; }
    }
    __threadfence_block();
    __syncthreads();
    if (wv == 0) {                                                       // exclusive prefixes, 64 groups per step
        uint32_t cl = 0, ct = 0;
        for (uint32_t g0 = 0; g0 <= ngroups; g0 += LANES) {
            const uint32_t gi = g0 + lane;
            const uint32_t a = gi < ngroups ? sLit[gi] : 0, t = gi < ngroups ? sTot[gi] : 0;
            uint32_t ia = a, it = t;
            for (int o = 1; o < LANES; o <<= 1) {
                const uint32_t x = __shfl_up(ia, o), y = __shfl_up(it, o);
                if (lane >= (uint32_t)o) { ia += x; it += y; }
            }
            if (gi <= ngroups) { sLit[gi] = cl + ia - a; sTot[gi] = ct + it - t; }
            cl += (uint32_t)__builtin_amdgcn_readlane(ia, LANES - 1); ct += (uint32_t)__builtin_amdgcn_readlane(it, LANES - 1);
        }
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t allLit = DUNI(sLit[ngroups]), allTot = DUNI(sTot[ngroups]);
    if (allLit > litSize || allTot + (litSize - allLit) != regen) { if (tid == 0) ZB_STORE_AGENT(&C->mode, 0u); return; }    // (sums the decode kernel has checked)
    // sweep 2: every group on its own
    bool fail = false;
    for (uint32_t gi = wv; gi < ngroups; gi += ZB_SC_WAVES) {
        const uint32_t g = gi * LANES;
        const uint32_t cnt = nbSeq - g < LANES ? nbSeq - g : LANES;
        const uint32_t lp = DUNI(sLit[gi]), opos = myStart + DUNI(sTot[gi]), groupTot = DUNI(sTot[gi + 1]) - DUNI(sTot[gi]);
        const bool valid = lane < cnt;
        const uint32_t ll = valid ? sLL[g + lane] : 0, ml = valid ? sML[g + lane] : 0;
        uint32_t off = valid ? sOF[g + lane] : 0;
        if (off & ZB_SYM) off = zb_subst(off, h0, h1, h2);
        uint32_t litIncl = ll, totIncl = ll + ml;
        for (int o = 1; o < LANES; o <<= 1) {
            const uint32_t a = __shfl_up(litIncl, o), t = __shfl_up(totIncl, o);
            if (lane >= (uint32_t)o) { litIncl += a; totIncl += t; }
        }
        const uint32_t myLit = lp + litIncl - ll, myOut = opos + totIncl - (ll + ml), mOut = myOut + ll;
        if (__any(valid && ml && (off == 0 || off > mOut))) { fail = true; break; }
        // the group's ~1.5 KB of output, one word per byte, written by all lanes side by side: position -> its sequence by a
        // binary search over the 64 start positions (a lane walking its own run would serialise a 100 KB match on one lane)
        WAVE_SYNC();
        gStart[wv][lane] = valid ? myOut : opos + groupTot; gLL[wv][lane] = ll; gLit[wv][lane] = myLit; gSrc[wv][lane] = mOut - off;
        if (lane == 0) gStart[wv][LANES] = opos + groupTot;
        __threadfence_block();
        WAVE_SYNC();
        for (uint32_t p = opos + lane; p < opos + groupTot; p += LANES) {
            uint32_t i = 0;
            for (uint32_t s_ = 32; s_; s_ >>= 1) if (gStart[wv][i + s_] <= p) i += s_;     // the last sequence that starts at or before p
            const uint32_t rel = p - gStart[wv][i], l_ = gLL[wv][i];
            words[p] = rel < l_ ? (ZB_LIT | litPtr[gLit[wv][i] + rel]) : gSrc[wv][i] + (rel - l_);
        }
    }
    if (fail && lane == 0) sFail = 1;
    // the literals behind the last sequence
    for (uint32_t k = tid; k < litSize - allLit; k += ZB_SC_WAVES * LANES) words[myStart + allTot + k] = ZB_LIT | litPtr[allLit + k];
    __threadfence_block();
    __syncthreads();
    if (tid == 0 && sFail) ZB_STORE_AGENT(&C->mode, 0u);                // the chunk-serial kernel behind this launch redoes the chunk
}

// one jump pass: every unresolved word takes its source's word, twice (four words per thread).  Chain depths are small in practice
// (log-like content: every word resolved after 7 jumps = 4 passes, its matches reach ~100 KB back, not to the previous record) while the
// launcher must queue the passes the WORST case needs (a 4 MiB chain at offset 1 or 2: three guaranteed hops per pass, 15 passes queued): a pass notes whether it left
// anything unresolved, and the passes behind one that did not return at their first instruction.
#ifdef HIPEMU
// Test harness only (tests/test_emu_zstd.py::test_jump_passes_cover_the_worst_store_order): the emulator runs a grid's threads one after
// the other, so an in-place pass sees every earlier thread's stores - the BEST order for pointer jumping.  With a snapshot every read of
// a pass sees the words as they were before it - the worst order the device can produce (three hops per pass) - and the pass bound of
// tsx_launch_zstd_decompress_blocks can be checked against it on the CPU.
static const uint8_t* g_zb_snap = nullptr;
#endif
__global__ __launch_bounds__(256) void zb_jump_kernel(uint8_t* __restrict__ hdrs, uint8_t* __restrict__ arenas, uint64_t astride, uint32_t lit_cap, uint32_t seq_cap,
                                                      uint32_t round) {
    const uint32_t chunk = blockIdx.y;
    ZbChunk* const C = (ZbChunk*)(hdrs + (size_t)chunk * ZB_CHUNK_HDR_BYTES);
    if (C->mode != 1) return;
    if (round > 0 && C->live[round - 1] == 0) return;
    const uint32_t n = C->contentSize, p = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p >= n) return;
    uint32_t* const words = (uint32_t*)(arenas + (size_t)chunk * astride + lit_cap + 12u * (size_t)seq_cap);
    const uint32_t* rd = words;                                         // what this pass reads (the array itself: the update is in place)
#ifdef HIPEMU
    if (g_zb_snap) rd = (const uint32_t*)(g_zb_snap + ((const uint8_t*)words - arenas));
#endif
    if (p + 4 <= n) {
        uint4 v = *reinterpret_cast<const uint4*>(rd + p);
        if ((v.x & v.y & v.z & v.w) & ZB_LIT) return;                   // all four resolved already
        // two jumps per pass: a word that is still a position after the first takes its source's word once more (a pass costs one
        // read and one write of the word array whatever it resolves)
        uint32_t a = (v.x & ZB_LIT) ? v.x : rd[v.x], b_ = (v.y & ZB_LIT) ? v.y : rd[v.y], c = (v.z & ZB_LIT) ? v.z : rd[v.z], d_ = (v.w & ZB_LIT) ? v.w : rd[v.w];
        if (!((a & b_ & c & d_) & ZB_LIT)) {
            a = (a & ZB_LIT) ? a : rd[a]; b_ = (b_ & ZB_LIT) ? b_ : rd[b_]; c = (c & ZB_LIT) ? c : rd[c]; d_ = (d_ & ZB_LIT) ? d_ : rd[d_];
        }
        v.x = a; v.y = b_; v.z = c; v.w = d_;
        *reinterpret_cast<uint4*>(words + p) = v;
        if (!((a & b_ & c & d_) & ZB_LIT) && round < 32) C->live[round] = 1;       // (every writer stores the same value)
    } else {
        for (uint32_t q = p; q < n; q++) {
            const uint32_t v = rd[q];
            if (!(v & ZB_LIT)) { uint32_t w = rd[v]; if (!(w & ZB_LIT)) w = rd[w]; words[q] = w; if (!(w & ZB_LIT) && round < 32) C->live[round] = 1; }
        }
    }
}

// words -> bytes; a word that is still a position after the last round means a corrupt chain: the chunk goes back to the fallback
__global__ __launch_bounds__(256) void zb_emit_kernel(const tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ dst_base, uint8_t* __restrict__ hdrs,
                                                      uint8_t* __restrict__ arenas, uint64_t astride, uint32_t lit_cap, uint32_t seq_cap) {
    const uint32_t chunk = blockIdx.y;
    ZbChunk* const C = (ZbChunk*)(hdrs + (size_t)chunk * ZB_CHUNK_HDR_BYTES);
    if (C->mode != 1) return;
    const uint32_t n = C->contentSize, p = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (p >= n) return;
    const uint32_t* const words = (const uint32_t*)(arenas + (size_t)chunk * astride + lit_cap + 12u * (size_t)seq_cap);
    uint8_t* const out = dst_base + descs[chunk].dst_off;                // slots are 16-byte aligned
    uint32_t all = ZB_LIT;
    if (p + 16 <= n) {
        uint32_t w[4];
        for (int k = 0; k < 4; k++) {
            const uint4 v = *reinterpret_cast<const uint4*>(words + p + 4 * k);
            all &= v.x & v.y & v.z & v.w;
            w[k] = (v.x & 0xFF) | (v.y & 0xFF) << 8 | (v.z & 0xFF) << 16 | (v.w & 0xFF) << 24;
        }
        *reinterpret_cast<uint4*>(out + p) = make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        for (uint32_t q = p; q < n; q++) { const uint32_t v = words[q]; all &= v; out[q] = (uint8_t)v; }
    }
    if (!(all & ZB_LIT)) ZB_STORE_AGENT(&C->mode, 0u);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
static inline uint32_t zb_lit_cap(uint32_t max_out) { return ((max_out + 80u * ZB_MAX_BLOCKS) + 255u) & ~255u; }
static inline uint32_t zb_seq_cap(uint32_t max_out) { return ((max_out / 3u + 64u * ZB_MAX_BLOCKS + 64u) + 63u) & ~63u; }
static inline size_t zb_arena_stride(uint32_t max_out) {
    return ((size_t)zb_lit_cap(max_out) + 12u * (size_t)zb_seq_cap(max_out) + 4u * ((size_t)max_out + 64u) + 255u) & ~(size_t)255u;
}
// workspace of a batch: [n chunk headers][n arenas: literals | literal lengths | match lengths | offsets | one word per output byte]
size_t tsx_zstd_blockmode_bytes(uint32_t n, uint32_t max_out) { return (size_t)n * (ZB_CHUNK_HDR_BYTES + zb_arena_stride(max_out)); }
bool tsx_zstd_blockmode_takes(uint32_t max_out) { return max_out <= ZB_MAX_CHUNK; }
// the list zstd_decompress_kernel skips by: word i * stride == 1 <=> chunk i was decoded here
const uint32_t* tsx_zstd_blockmode_skip(const void* bwork, uint32_t* stride_words) {
    *stride_words = (uint32_t)(ZB_CHUNK_HDR_BYTES / 4);
    return (const uint32_t*)((const uint8_t*)bwork + offsetof(ZbChunk, mode));
}

uint32_t tsx_launch_zstd_decompress_blocks(hipStream_t st, const uint8_t* frames, int from_mid, uint64_t mid_stride, tsx_chunk_desc* d_descs, uint32_t n,
                                           uint32_t max_out, uint8_t* dst, int32_t* d_status, void* bwork) {
    if (!n) return 0;
    uint8_t* const hdrs = (uint8_t*)bwork;
    uint8_t* const arenas = hdrs + (size_t)n * ZB_CHUNK_HDR_BYTES;
    const size_t astride = zb_arena_stride(max_out);
    const uint32_t lit_cap = zb_lit_cap(max_out), seq_cap = zb_seq_cap(max_out);
    hipLaunchKernelGGL(zb_index_kernel, dim3(n), dim3(LANES), 0, st, frames, from_mid, mid_stride, (const tsx_chunk_desc*)d_descs, (const int32_t*)d_status, hdrs, lit_cap, seq_cap);
    hipLaunchKernelGGL(zb_decode_kernel, dim3(ZB_MAX_BLOCKS, n), dim3(2 * LANES), 0, st, frames, from_mid, mid_stride, (const tsx_chunk_desc*)d_descs, hdrs, arenas, (uint64_t)astride, lit_cap, seq_cap
#ifdef TSX_PROF2
                       , g_zbprof_out
#endif
                       );
    hipLaunchKernelGGL(zb_scatter_kernel, dim3(ZB_MAX_BLOCKS, n), dim3(ZB_SC_WAVES * LANES), 0, st, frames, from_mid, mid_stride, d_descs, hdrs, arenas, (uint64_t)astride, lit_cap, seq_cap);
    // A copy chain is at most as long as the chunk.  A pass makes two jumps IN PLACE: the first reads its source's word, the second the
    // word of what that named - either may still hold its value from before the pass (another thread has not stored yet), so what a pass
    // guarantees is old[old[old[q]]]: three hops of the chain as it was, not four.  ceil(log3(max_out)) + 1 passes therefore resolve every
    // word whatever the order of the stores (14 + 1 for a 4 MiB run at offset 1; round 3 queued log4 + 1 = 13 and such chunks silently
    // took the chunk-serial kernel as well).  The passes behind one that resolved everything return at their first instruction.
    uint32_t rounds = 1;
    for (uint64_t reach = 1; reach < (uint64_t)max_out; reach *= 3) rounds++;
    static_assert(sizeof(((ZbChunk*)0)->live) / sizeof(((ZbChunk*)0)->live[0]) >= 18, "live[] covers the passes of a 16 MiB chunk");
    const uint32_t tiles = (max_out + 1023) / 1024;                     // 256 threads x 4 words
#ifdef HIPEMU
    // test harness: TSX_EMU_JUMP_SNAPSHOT=1 gives every pass the word array as it was before the pass (the worst store order),
    // TSX_EMU_JUMP_ROUNDS=k queues k passes instead of the bound above (to show what too few do)
    uint8_t* snap = getenv("TSX_EMU_JUMP_SNAPSHOT") ? (uint8_t*)malloc((size_t)n * astride) : nullptr;
    if (const char* e = getenv("TSX_EMU_JUMP_ROUNDS")) { const long v = atol(e); if (v >= 1 && v <= 31) rounds = (uint32_t)v; }
#endif
    for (uint32_t r = 0; r < rounds; r++) {
#ifdef HIPEMU
        if (snap) { memcpy(snap, arenas, (size_t)n * astride); g_zb_snap = snap; }
#endif
        hipLaunchKernelGGL(zb_jump_kernel, dim3(tiles, n), dim3(256), 0, st, hdrs, arenas, (uint64_t)astride, lit_cap, seq_cap, r);
    }
#ifdef HIPEMU
    if (snap) { g_zb_snap = nullptr; free(snap); }                     // (snapshot mode is a single-threaded test: callers without it never touch the word)
#endif
    hipLaunchKernelGGL(zb_emit_kernel, dim3((max_out + 4095) / 4096, n), dim3(256), 0, st, (const tsx_chunk_desc*)d_descs, dst, hdrs, arenas, (uint64_t)astride, lit_cap, seq_cap);
    return 4 + rounds;
}
