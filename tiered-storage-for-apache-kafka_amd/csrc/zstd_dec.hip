// Zstandard frame decoder — gfx950, one workgroup of three 64-lane wavefronts per chunk.
//
// Replaces zstd-jni's  Zstd.decompressedSize(chunk) / Zstd.decompress(chunk, size)
//   core/src/main/java/io/aiven/kafka/tieredstorage/transform/DecompressionChunkEnumeration.java:39-46
// and accepts any single Zstandard frame (RFC 8878) with a known content size and no dictionary — not only
// the frames our own compressor writes: raw / RLE / compressed blocks, raw / RLE / Huffman (1 or 4 streams,
// treeless) literals, predefined / RLE / FSE / repeat sequence tables, repeat offsets.
// Errors are per chunk: TSX_E_BAD_SIZE when the frame declares no usable content size (the reference throws
// "Invalid decompressed size"), TSX_E_DST_TOO_SMALL, TSX_E_BAD_FRAME for anything malformed.  A content
// checksum, when present, is skipped, not verified (the reference's writer never emits one).
//
// A block goes through three stages, each a serial chain that keeps only a few lanes busy, so the chunk's three
// waves run them one block apart and meet at one workgroup barrier per block:
//   wave 1  block headers + literals: the 1 or 4 Huffman streams on lanes 0-3, through per-stream LDS windows and an
//           11-bit table that yields up to three symbols per read                          -> literal buffer k % 3
//   wave 0  sequence stream: the LL / ML / OF state machines in lanes 0-2 (one table read, two DPP adds per sequence),
//           then every lane extracts its own sequence's extra bits; repeat offsets resolved in order -> arrays k & 1
//   wave 2  execution: 64 sequences per step, positions by prefix sums, copies in dependency rounds  -> the output
// Every loop is bounded by sizes read from the frame; every index into LDS or the workspace is checked against them.
#include "zstd_dec_dev.h"
#ifdef TSX_PROF2
static unsigned long long* g_dprof_out = nullptr;                     // 8 u64 per chunk: phase laps of the decoder
extern "C" void tsx_debug_set_dprof(void* dev_ptr) { g_dprof_out = (unsigned long long*)dev_ptr; }
#define DLT(k) do { const unsigned long long n_ = (unsigned long long)clock64(); dlt_[k] += n_ - dlast_; dlast_ = n_; } while (0)
#else
#define DLT(k) do {} while (0)
#endif

// ---- the kernel -------------------------------------------------------------------------------------------
#define FAIL(code) do { err = (code); goto done; } while (0)                 /* both waves, before the block loop */
#define RFAIL(code) do { myErr = (code); goto block_end; } while (0)        /* one wave, inside its role */

__device__ __forceinline__ static void zstd_decompress_body(DecLds& L, const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                            tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ dst_base,
                                                            int32_t* __restrict__ status, uint8_t* __restrict__ work,
                                                            const uint32_t* __restrict__ skip, uint32_t skip_stride
#ifdef TSX_PROF2
                                                            , unsigned long long* __restrict__ dprof
#endif
                                                            ) {
    const uint32_t lane = threadIdx.x & (LANES - 1), role = DUNI(threadIdx.x >> 6), chunk = blockIdx.x;   // wave 0: sequence streams, wave 1: literals, wave 2: execution
    if (status[chunk] != TSX_OK) return;
    if (skip && skip[(size_t)chunk * skip_stride] == 1) return;       // decoded by the block-parallel form (zstd_dec_blocks.hip)
    const tsx_chunk_desc d = descs[chunk];
    const uint8_t* __restrict__ src = from_mid ? frames + (uint64_t)chunk * mid_stride : frames + d.src_off;
    const uint32_t srcSize = from_mid ? d.src_len - 28 : d.src_len;
    uint8_t* __restrict__ out = dst_base + d.dst_off;
    uint8_t* const ws = work + (size_t)chunk * ZS_WS_BYTES;
    // literals of block k -> litBuf[k % 3], its sequences -> seqBuf[k & 1] (three arrays of ZS_DSEQ_CAP dwords); the decoder needs no hash tables
    uint8_t* const litBuf[3] = {ws + ZS_WS_LIT, ws + ZS_WS_HASHLONG, ws + ZS_WS_HASHLONG + (192u << 10)};
    uint8_t* const seqBuf[2] = {ws + ZS_WS_SEQS, ws + ZS_WS_STBITS};
    int32_t err = TSX_OK;
#ifdef TSX_PROF2
    unsigned long long dlt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast_ = (unsigned long long)clock64();
#endif
    uint32_t opos = 0;
    uint32_t rep0 = 1, rep1 = 4, rep2 = 8;                           // repeat-offset history, carried across the blocks of the frame (wave-uniform)
    uint64_t contentSize = 0;
    uint32_t p = 0;
    bool hasChecksum = false, prodDone = false, fseDone = false, frameDone = false;
    // ---- frame header ----
    if (srcSize < 6) FAIL(DERR_FRAME);
    if (src[0] != 0x28 || src[1] != 0xB5 || src[2] != 0x2F || src[3] != 0xFD) FAIL(DERR_FRAME);
    {   const uint32_t fhd = src[4];
        const uint32_t single = (fhd >> 5) & 1, dictFlag = fhd & 3, fcsFlag = fhd >> 6;
        hasChecksum = (fhd >> 2) & 1;
        if (fhd & 8) FAIL(DERR_FRAME);                                  // reserved bit
        p = 5;
        if (!single) { if (p >= srcSize) FAIL(DERR_FRAME); if ((src[p] >> 3) > 21) FAIL(DERR_FRAME); p++; }
        const uint32_t dl = dictFlag == 0 ? 0 : dictFlag == 1 ? 1 : dictFlag == 2 ? 2 : 4;
        if (p + dl > srcSize) FAIL(DERR_FRAME);
        {   uint32_t dictId = 0; for (uint32_t i = 0; i < dl; i++) dictId |= (uint32_t)src[p + i] << (8 * i);
            if (dictId) FAIL(DERR_FRAME); }                             // dictionaries are not supported (the reference uses none)
        p += dl;
        const uint32_t fl = fcsFlag == 0 ? single : fcsFlag == 1 ? 2 : fcsFlag == 2 ? 4 : 8;
        if (fl == 0) FAIL(TSX_E_BAD_SIZE);                              // unknown content size: "Invalid decompressed size"
        if (p + fl > srcSize) FAIL(DERR_FRAME);
        for (uint32_t i = 0; i < fl; i++) contentSize |= (uint64_t)src[p + i] << (8 * i);
        if (fl == 2) contentSize += 256;
        p += fl;
    }
    if (contentSize > d.dst_cap) FAIL(TSX_E_DST_TOO_SMALL);
    if (role == 0 && lane == 0) { L.hufValid = 0; L.llValid = 0; L.ofValid = 0; L.mlValid = 0; L.zeroEntry = 0; L.err = TSX_OK; }
    if (role == 0 && lane < 36) { L.cLLbase[lane] = dLLbase[lane]; L.cLLbits[lane] = dLLbits[lane]; }
    if (role == 0 && lane < 53) { L.cMLbase[lane] = dMLbase[lane]; L.cMLbits[lane] = dMLbits[lane]; }
    __syncthreads();
    // ---- blocks: wave 1 (literals) works one block ahead of wave 0 (sequences + execution) ----
    // Iteration `it`: wave 1 parses block it's header and literals section (Huffman streams -> litBuf[it & 1]) and publishes
    // L.desc[it & 1]; wave 0 decodes and executes the sequences of block it - 1.  One workgroup barrier per iteration; an error
    // of either wave is posted in L.err and ends both after the barrier.  Neither phase of a block waits for the other's
    // memory latency any more: the serial chains of the two phases run side by side.
    for (uint32_t it = 0;; it++) {
        int32_t myErr = TSX_OK;
        // Was block it - 1 the frame's last one?  Each wave answers from its own registers (a descriptor slot may already be
        // rewritten by the other wave when the barrier opens): wave 1 finished producing, wave 0 consumed a block marked last.
        {
            if (role == 1) {
                if (!prodDone) {
                    uint8_t* const lit = litBuf[it % 3];
                    if (p + 3 > srcSize) RFAIL(DERR_FRAME);
                    const uint32_t bh = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
                    p += 3;
                    const uint32_t last = bh & 1, btype = (bh >> 1) & 3, bsize = bh >> 3;
                    BlkDesc bd; bd.off = p; bd.bsize = bsize; bd.btype = btype; bd.last = last; bd.litInPlace = 0; bd.litOff = 0; bd.litSize = 0; bd.q = 0;
                    if (btype == 0) { if (p + bsize > srcSize) RFAIL(DERR_FRAME); p += bsize; }
                    else if (btype == 1) { if (p + 1 > srcSize) RFAIL(DERR_FRAME); p += 1; }
                    else if (btype == 2) {
                        if (bsize > ZS_BLOCK_MAX || p + bsize > srcSize || bsize < 2) RFAIL(DERR_FRAME);
                        const uint8_t* const blk = src + p;
                        const uint32_t b0 = blk[0], ltype = b0 & 3, sf = (b0 >> 2) & 3;
                        uint32_t litSize = 0, q = 0;
                        uint32_t litInPlace = 0, litOff = 0;
                        if (ltype < 2) {
                            uint32_t hl;
                            if (sf == 0 || sf == 2) { litSize = b0 >> 3; hl = 1; }
                            else if (sf == 1) { if (bsize < 2) RFAIL(DERR_FRAME); litSize = (b0 >> 4) + ((uint32_t)blk[1] << 4); hl = 2; }
                            else { if (bsize < 3) RFAIL(DERR_FRAME); litSize = (b0 >> 4) + ((uint32_t)blk[1] << 4) + ((uint32_t)blk[2] << 12); hl = 3; }
                            if (litSize > ZS_BLOCK_MAX) RFAIL(DERR_FRAME);
                            if (ltype == 0) { if (hl + litSize > bsize) RFAIL(DERR_FRAME); litInPlace = 1; litOff = hl; q = hl + litSize; }
                            else {
                                if (hl + 1 > bsize) RFAIL(DERR_FRAME);
                                const uint8_t b = blk[hl];
                                for (uint32_t i = lane; i < litSize; i += LANES) lit[i] = b;
                                q = hl + 1;
                            }
                        } else {
                            uint32_t hl, bits, streams; uint64_t v = 0;
                            if (sf == 0) { hl = 3; bits = 10; streams = 1; }
                            else if (sf == 1) { hl = 3; bits = 10; streams = 4; }
                            else if (sf == 2) { hl = 4; bits = 14; streams = 4; }
                            else { hl = 5; bits = 18; streams = 4; }
                            if (hl > bsize) RFAIL(DERR_FRAME);
                            for (uint32_t i = 0; i < hl; i++) v |= (uint64_t)blk[i] << (8 * i);
                            litSize = (uint32_t)(v >> 4) & ((1u << bits) - 1);
                            const uint32_t csize = (uint32_t)(v >> (4 + bits)) & ((1u << bits) - 1);
                            if (litSize > ZS_BLOCK_MAX || hl + csize > bsize || litSize == 0) RFAIL(DERR_FRAME);
                            uint32_t t = hl;
                            if (ltype == 2) {
                                if (lane == 0) L.scalH[0] = huf_readTable(L, blk + hl, csize);
                                WAVE_SYNC();
                                const uint32_t used = L.scalH[0];
                                WAVE_SYNC();
                                if (!used) RFAIL(DERR_FRAME);
                                huf_buildX_wave(L, lane);
                                WAVE_SYNC();
                                t += used;
                            } else if (!L.hufValid) RFAIL(DERR_FRAME);
                            const uint32_t payload = hl + csize - t;
                            // stream layout
                            uint32_t sOff[5], sCnt[4];
                            if (streams == 1) { sOff[0] = 0; sOff[1] = payload; sCnt[0] = litSize; }
                            else {
                                if (payload < 10) RFAIL(DERR_FRAME);
                                const uint32_t s1 = blk[t] | (blk[t + 1] << 8), s2 = blk[t + 2] | (blk[t + 3] << 8), s3 = blk[t + 4] | (blk[t + 5] << 8);
                                if (6 + (uint64_t)s1 + s2 + s3 >= payload) RFAIL(DERR_FRAME);
                                sOff[0] = 6; sOff[1] = 6 + s1; sOff[2] = sOff[1] + s2; sOff[3] = sOff[2] + s3; sOff[4] = payload;
                                const uint32_t seg = (litSize + 3) / 4;
                                if (3 * seg > litSize) RFAIL(DERR_FRAME);
                                sCnt[0] = sCnt[1] = sCnt[2] = seg; sCnt[3] = litSize - 3 * seg;
                            }
                            // The (1 or 4) Huffman streams decode on lanes 0..3, each through its own LDS window of the stream, refilled by
                            // the whole wave whenever a lane gets close to its window's lower edge (a reload from global memory would be a
                            // dependent round trip every four symbols).
                            bool ok = true;
                            {
                                const bool mine = lane < streams;
                                uint32_t o = 0; for (uint32_t k = 0; k < lane && k < 4; k++) o += mine ? sCnt[k] : 0;
                                const uint32_t cnt = mine ? sCnt[lane] : 0, sn = mine ? sOff[lane + 1] - sOff[lane] : 0, sbeg = mine ? t + sOff[lane] : 0;
                                uint8_t* const outp = lit + o;
                                // Bh = bits of the stream not read yet (cursor from the top; the last byte carries the end mark).  A step
                                // decodes four symbols (<= 44 bits) from ONE 8-byte window read at the cursor, no branches inside; the last
                                // symbols of a stream (fewer than four left, or fewer than 44 bits) go one at a time, with the bits below
                                // the stream's first one read as zeros like libzstd's container does.
                                uint32_t hi = 0, Bh = 0; bool hdone = !mine;
                                if (mine) {
                                    const uint32_t lastByte = sn ? blk[sbeg + sn - 1] : 0;
                                    if (lastByte == 0) { ok = false; hdone = true; }
                                    else Bh = 8 * (sn - 1) + dhb32(lastByte);
                                }
                                const uint32_t tableLog = L.hufLog, tmask = (1u << tableLog) - 1;
                                for (;;) {
                                    const uint32_t myTop = hdone ? 0 : (Bh >> 3) + 8;                                // bytes past the stream's end are zeros
                                    const uint32_t myWb = myTop > ZS_HWIN ? (myTop - ZS_HWIN + 15) & ~15u : 0;     // top - wb <= ZS_HWIN = one 16-byte piece per lane
                                    for (uint32_t s_ = 0; s_ < streams; s_++) {
                                        const uint32_t top = (uint32_t)__builtin_amdgcn_readlane(myTop, (int)s_), wb = (uint32_t)__builtin_amdgcn_readlane(myWb, (int)s_), beg = (uint32_t)__builtin_amdgcn_readlane(sbeg, (int)s_), n_ = (uint32_t)__builtin_amdgcn_readlane(sn, (int)s_);
                                        const uint32_t k = lane * 16;
                                        if (wb + k < top) {
                                            uint4 v;
                                            if (wb + k + 16 <= n_) __builtin_memcpy(&v, blk + beg + wb + k, 16);
                                            else { uint8_t tmp[16]; for (uint32_t j = 0; j < 16; j++) tmp[j] = wb + k + j < n_ ? blk[beg + wb + k + j] : 0; __builtin_memcpy(&v, tmp, 16); }
                                            *reinterpret_cast<uint4*>(&L.hwin[s_ * (ZS_HWIN + 16) + k]) = v;
                                        }
                                    }
                                    __threadfence_block();
                                    WAVE_SYNC();
                                    if (!hdone) {
                                        const uint8_t* const win = &L.hwin[lane * (ZS_HWIN + 16)];
                                        // five table reads per 8-byte window read: each yields the one to three symbols coded in the next 11 bits
                                        while (hi + 16 <= cnt && Bh >= 56 && ((Bh - 56) >> 3) >= myWb) {
                                            const uint32_t lo = Bh - 56;
                                            const uint64_t c = wld64(win, myWb, lo >> 3) >> (lo & 7);               // bits [lo, lo + 56) of the stream
                                            uint32_t used = 0;
                                            #pragma unroll
                                            for (int k = 0; k < 5; k++) {
                                                const uint32_t e = L.hufX[(uint32_t)(c >> (45 - used)) & 0x7FF];
                                                const uint32_t sy = e & 0xFFFFFF;                                  // one byte of slack behind the symbols
                                                __builtin_memcpy(outp + hi, &sy, 4);
                                                hi += e >> 28; used += (e >> 24) & 15;
                                            }
                                            Bh -= used;
                                        }
                                        while (hi < cnt && (hi + 16 > cnt || Bh < 56)) {                             // the stream's tail, one symbol at a time
                                            const uint32_t need = Bh < tableLog ? Bh : tableLog, lo = Bh - need;
                                            if ((lo >> 3) < myWb) break;                                             // behind the window: refill first
                                            const uint32_t bits = (uint32_t)(wld64(win, myWb, lo >> 3) >> (lo & 7)) & ((1u << need) - 1);
                                            const uint32_t e = huf_decode1(L, (bits << (tableLog - need)) & tmask, tableLog);
                                            if ((e >> 8) > Bh) { ok = false; hdone = true; break; }                   // reads past the stream's first bit
                                            outp[hi++] = (uint8_t)e; Bh -= e >> 8;
                                        }
                                        if (!hdone && hi >= cnt) { if (Bh != 0) ok = false; hdone = true; }           // every bit used, none missing
                                    }
                                    WAVE_SYNC();
                                    if (__all(hdone)) break;
                                }
                            }
                            if (__any(!ok)) RFAIL(DERR_FRAME);
                            q = hl + csize;
                        }
                        bd.litInPlace = litInPlace; bd.litOff = litOff; bd.litSize = litSize; bd.q = q;
                        p += bsize;
                    } else RFAIL(DERR_FRAME);
                    if (last) {
                        if (hasChecksum) { if (p + 4 > srcSize) RFAIL(DERR_FRAME); p += 4; }
                        if (p != srcSize) RFAIL(DERR_FRAME);
                        prodDone = true;
                    }
                    if (lane == 0) L.desc[it % 3] = bd;
                    DLT(0);                                             // 0: block header + literals section
                }
            } else if (role == 2) {
                if (it >= 2) {
                    // ---- wave 2: execute block it - 2 (its literals were decoded two iterations ago, its sequences one) ----
                    const BlkDesc* const bdp = &L.desc[(it - 2) % 3];
                    const uint32_t bsize = DUNI(bdp->bsize), btype = DUNI(bdp->btype), boff = DUNI(bdp->off);
                    frameDone = DUNI(bdp->last) != 0;
                    if (btype == 0) {                                   // raw
                        if (opos + (uint64_t)bsize > contentSize) RFAIL(DERR_FRAME);
                        for (uint32_t i = lane; i < bsize; i += LANES) out[opos + i] = src[boff + i];
                        opos += bsize;
                        __threadfence_block();
                    } else if (btype == 1) {                            // RLE
                        if (opos + (uint64_t)bsize > contentSize) RFAIL(DERR_FRAME);
                        const uint8_t b = src[boff];
                        for (uint32_t i = lane; i < bsize; i += LANES) out[opos + i] = b;
                        opos += bsize;
                        __threadfence_block();
                    } else {
                        const uint32_t litSize = DUNI(bdp->litSize), nbSeq = DUNI(L.nseq[(it - 2) & 1]);
                        const uint8_t* const litPtr = DUNI(bdp->litInPlace) ? src + boff + DUNI(bdp->litOff) : litBuf[(it - 2) % 3];
                        const uint32_t* const sLL = (const uint32_t*)seqBuf[(it - 2) & 1]; const uint32_t* const sML = sLL + ZS_DSEQ_CAP; const uint32_t* const sOF = sML + ZS_DSEQ_CAP;
                        uint32_t lp = 0;
                        for (uint32_t g = 0; g < nbSeq; g += LANES) {
                            const uint32_t cnt = nbSeq - g < LANES ? nbSeq - g : LANES;
                            const bool valid = lane < cnt;
                            const uint32_t ll = valid ? sLL[g + lane] : 0, ml = valid ? sML[g + lane] : 0, off = valid ? sOF[g + lane] : 0;
                            // Execution.  Positions come from prefix sums, so literal runs and every match whose source lies before the
                            // group's first output byte are copied by their own lane, all at once; only matches that read bytes produced
                            // inside the same group (short offsets) are replayed in order with wave-wide copies.
                            uint32_t litIncl = ll, totIncl = ll + ml;
                            for (int o = 1; o < LANES; o <<= 1) {
                                const uint32_t a = __shfl_up(litIncl, o), t = __shfl_up(totIncl, o);
                                if (lane >= (uint32_t)o) { litIncl += a; totIncl += t; }
                            }
                            const uint32_t groupLit = (uint32_t)__builtin_amdgcn_readlane(litIncl, LANES - 1), groupTot = (uint32_t)__builtin_amdgcn_readlane(totIncl, LANES - 1);
                            if (lp + groupLit > litSize || (uint64_t)opos + groupTot > contentSize) RFAIL(DERR_FRAME);
                            const uint32_t myLit = lp + litIncl - ll, myOut = opos + totIncl - (ll + ml), mOut = myOut + ll;
                            if (__any(valid && ml && (off == 0 || off > mOut))) RFAIL(DERR_FRAME);
                            // Short runs are copied by their own lane (all lanes at once), long ones (> ZS_LONG_RUN bytes) by the whole wave.
                            // A match is ready when its source bytes are final: before the group's first output byte, or - after the
                            // fence that follows each round - inside literal runs and matches already copied.  Each round copies every
                            // pending match whose source touches no earlier pending match's destination; a match that overlaps its own
                            // destination (offset < length) is replayed by the whole wave, in 64-byte steps or as a periodic pattern.
                            const uint32_t s0 = mOut - off;
                            exec_copies(out + myOut, litPtr + myLit, ll, valid && ll, lane);
                            unsigned long long pend = __ballot(valid && ml);
                            bool first = true;
                            do {
                                const bool mineP = (pend >> lane) & 1;
                                bool blocked = mineP && off < ml;
                                if (first) blocked = mineP && s0 + ml > opos;                          // round 0: only sources before the group
                                else
                                    for (unsigned long long m = pend; m; m &= m - 1) {
                                        const int j = __ffsll((long long)m) - 1;
                                        const uint32_t dj = __builtin_amdgcn_readlane(mOut, j), ej = dj + __builtin_amdgcn_readlane(ml, j);
                                        if ((uint32_t)j < lane && s0 < ej && s0 + ml > dj) blocked = true;
                                    }
                                const unsigned long long ready = __ballot(mineP && !blocked);
                                if (ready || first) {
                                    exec_copies(out + mOut, out + s0, ml, (ready >> lane) & 1, lane);
                                    pend &= ~ready;
                                    first = false;
                                } else {                                                                // the first pending match overlaps itself
                                    const int i = __ffsll((long long)pend) - 1;
                                    pend &= pend - 1;
                                    const uint32_t dpos = __builtin_amdgcn_readlane(mOut, i), o_ = __builtin_amdgcn_readlane(off, i), m_ = __builtin_amdgcn_readlane(ml, i);
                                    const uint32_t from = dpos - o_;
                                    if (o_ >= LANES) {
                                        for (uint32_t k = 0; k < m_; k += LANES) {
                                            if (k) __threadfence_block();                               // a 64-byte step may read bytes written by the previous step
                                            if (k + lane < m_) out[dpos + k + lane] = out[from + k + lane];
                                        }
                                    } else {
                                        for (uint32_t k = lane; k < m_; k += LANES) out[dpos + k] = out[from + (k % o_)];   // periodic pattern
                                    }
                                }
                                __threadfence_block();
                            } while (pend);
                            lp += groupLit; opos += groupTot;
                        }
                        const uint32_t tail = litSize - lp;
                        if ((uint64_t)opos + tail > contentSize) RFAIL(DERR_FRAME);
                        for (uint32_t k = lane; k < tail; k += LANES) out[opos + k] = litPtr[lp + k];
                        opos += tail;
                        __threadfence_block();
                    }
                    DLT(3);                                             // 3: execution
                }
            } else if (it >= 1 && !fseDone) {
                // ---- wave 0: the sequences of block it - 1 -> (literal length, match length, offset) arrays in seqBuf[(it - 1) & 1] ----
                const BlkDesc* const bdp = &L.desc[(it - 1) % 3];
                const uint32_t bsize = DUNI(bdp->bsize), btype = DUNI(bdp->btype), boff = DUNI(bdp->off);
                if (DUNI(bdp->last)) fseDone = true;
                if (btype == 2) {
                    uint32_t* const sLL = (uint32_t*)seqBuf[(it - 1) & 1]; uint32_t* const sML = sLL + ZS_DSEQ_CAP; uint32_t* const sOF = sML + ZS_DSEQ_CAP;
                    const uint8_t* const blk = src + boff;
                    uint32_t q = DUNI(bdp->q);
                    if (q >= bsize) RFAIL(DERR_FRAME);
                    uint32_t nbSeq = blk[q];
                    if (nbSeq == 0) q += 1;
                    else if (nbSeq < 128) q += 1;
                    else if (nbSeq < 255) { if (q + 2 > bsize) RFAIL(DERR_FRAME); nbSeq = ((nbSeq - 128) << 8) + blk[q + 1]; q += 2; }
                    else { if (q + 3 > bsize) RFAIL(DERR_FRAME); nbSeq = blk[q + 1] + ((uint32_t)blk[q + 2] << 8) + 0x7F00; q += 3; }
                    nbSeq = DUNI(nbSeq);                                        // loaded through the vector path: pin it (and every loop bound derived from it) to SGPRs
                    if (nbSeq > ZS_BLOCK_MAX / 3 + 1) RFAIL(DERR_FRAME);            // 128 KiB / minMatch 3 = 43691 at most in a valid block
                    if (nbSeq) {
                        // The three sequence tables (literal lengths, offsets, match lengths).  Lane 0 parses each table description
                        // (a short serial bit parse); the decoding table itself is built by the whole wave (fse_buildSeqTable_wave).
                        if (q >= bsize) RFAIL(DERR_FRAME);
                        const uint32_t modes = DUNI(blk[q]);
                        uint32_t t = q + 1;
                        if (modes & 3) RFAIL(DERR_FRAME);
                        for (int k = 0; k < 3; k++) {
                            const uint32_t mode = (modes >> (6 - 2 * k)) & 3;
                            SeqD* const dt = k == 0 ? L.ll : k == 1 ? L.of : L.ml;
                            uint32_t* const logp = k == 0 ? &L.llLog : k == 1 ? &L.ofLog : &L.mlLog;
                            int* const validp = k == 0 ? &L.llValid : k == 1 ? &L.ofValid : &L.mlValid;
                            const uint32_t maxSymK = k == 0 ? 35 : k == 1 ? 31 : 52, maxLogK = k == 0 ? 9 : k == 1 ? 8 : 9;
                            if (mode == 0) {                                    // predefined distribution
                                const short* const dn = k == 0 ? dLLnorm : k == 1 ? dOFnorm : dMLnorm;
                                const uint32_t dmax = k == 0 ? 35 : k == 1 ? 28 : 52, dlog = k == 1 ? 5 : 6;
                                if (lane <= dmax) L.norm[lane] = dn[lane];
                                if (lane == 0) { *logp = dlog; *validp = 1; }
                                __threadfence_block();
                                WAVE_SYNC();
                                if (!fse_buildSeqTable_wave(dt, L, dmax, dlog, k, lane)) RFAIL(DERR_FRAME);
                            } else if (mode == 1) {                             // RLE: one symbol, no state bits
                                if (t >= bsize) RFAIL(DERR_FRAME);
                                const uint32_t sym = DUNI(blk[t]);
                                t++;
                                if (sym > maxSymK) RFAIL(DERR_FRAME);
                                if (lane == 0) { dt[0] = SEQD(0, 0, seq_ebits(L, sym, k), sym); *logp = 0; *validp = 1; }
                            } else if (mode == 2) {                             // FSE-compressed distribution
                                if (t >= bsize) RFAIL(DERR_FRAME);
                                if (lane == 0) {
                                    uint32_t ms = maxSymK, tl = 0;
                                    const uint32_t used = fse_readNCount(L.norm, &ms, &tl, blk + t, bsize - t, maxLogK);
                                    L.scal[0] = used; L.scal[3] = ms; L.scal[4] = tl;
                                    if (used) { *logp = tl; *validp = 1; }
                                }
                                __threadfence_block();
                                WAVE_SYNC();
                                const uint32_t used = DUNI(L.scal[0]), ms = DUNI(L.scal[3]), tl = DUNI(L.scal[4]);
                                if (!used || !fse_buildSeqTable_wave(dt, L, ms, tl, k, lane)) RFAIL(DERR_FRAME);
                                t += used;
                            } else {                                            // repeat the previous block's table
                                WAVE_SYNC();
                                if (!*validp) RFAIL(DERR_FRAME);
                            }
                        }
                        if (t >= bsize) RFAIL(DERR_FRAME);
                        if (lane == 0) L.scal[2] = t;
                        __threadfence_block();
                        WAVE_SYNC();
                        DLT(1);                                                 // 1: sequence tables
                    } else if (q != bsize) RFAIL(DERR_FRAME);
                    // ---- decode and execute the sequences, 64 at a time (one per lane) ----
                    // The sequence bit stream is one serial chain (read backwards; each FSE state transition says how many bits the
                    // next one reads), staged through an LDS window that the whole wave refills.  Only the part of a sequence that IS
                    // serial runs serially: pass 1 walks the three state machines for up to 64 sequences as wave-uniform scalar code
                    // (three 4-byte table reads and one bit-window read per sequence) and drops each sequence's states and bit cursor
                    // into its own lane (v_writelane); pass 2 lets every lane pull its sequence's extra bits out of the window and form
                    // (literal length, match length, offset code) - all 64 at once; pass 3 resolves the repeat offsets in order (a short
                    // uniform loop over lane values), which makes the execution below order-free.
                    {
                        const uint32_t llLog = DUNI(L.llLog), ofLog = DUNI(L.ofLog), mlLog = DUNI(L.mlLog);
                        const uint8_t* const win = L.swin + ZS_DPAD;
                        const uint8_t* stream = blk; uint32_t n = 0;
                        uint32_t B = 0, wbase = 0, e = 0;                               // B: bits of the stream not read yet (the cursor, from the top)
                        // lanes 0, 1, 2 = the LL, ML, OF state machines; the others carry state 0 through an all-zero entry
                        const SeqD* const tbl = lane == 0 ? L.ll : lane == 1 ? L.ml : lane == 2 ? L.of : &L.zeroEntry;
                        uint16_t* const recp = &L.rec[lane < 3 ? lane : 3];
                        uint32_t st = 0;
                        bool filled = false;
                        if (nbSeq) {
                            const uint32_t t = DUNI(L.scal[2]);
                            n = DUNI(bsize - t); stream = blk + t;                  // n >= 1 (checked with the tables)
                            const uint32_t lastByte = DUNI(stream[n - 1]);          // BIT_initDStream: the last byte carries the end mark
                            if (lastByte == 0) RFAIL(DERR_FRAME);
                            B = 8 * (n - 1) + dhb32(lastByte);
                        }
                        TSX_SETPRIO(3);                                         // the sequence stage is the chunk's critical path: its chain goes first
                        for (uint32_t g = 0; g < nbSeq; g += LANES) {
                            const uint32_t cnt = DUNI(nbSeq - g < LANES ? nbSeq - g : LANES);
                            // 64 sequences read at most 64 * 89 bits = 712 bytes below the cursor; every read is an 8-byte load at
                            // byte (bit >> 3), so the window holds [wbase, (B >> 3) + 8) with the bytes past the stream's end as zeros
                            if (!filled || (wbase != 0 && (B >> 3) < wbase + 736)) {
                                WAVE_SYNC();                                    // everyone is done with the previous window
                                const uint32_t top = (B >> 3) + 8;
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
                                if (!filled) {                                      // initial states: LL, OF, ML (ZSTD_initFseState order)
                                    filled = true;
                                    const uint32_t lo = B - (llLog + ofLog + mlLog);              // <= 26 bits
                                    if ((int32_t)lo < 0) RFAIL(DERR_FRAME);
                                    const uint32_t w = DUNI((uint32_t)(wld64(win, wbase, lo >> 3) >> (lo & 7)));
                                    const uint32_t sm = w & ((1u << mlLog) - 1), so = (w >> mlLog) & ((1u << ofLog) - 1), sl = (w >> (mlLog + ofLog)) & ((1u << llLog) - 1);
                                    st = lane == 0 ? sl : lane == 1 ? sm : lane == 2 ? so : 0;
                                    B = lo;
                                }
                            }
                            // pass 1: the chain, on the vector unit.  Lanes 0, 1, 2 run the LL, ML and OF state machines (that is the order
                            // in which a sequence's state-update bits sit in the stream, highest first); one table read serves all three,
                            // two DPP adds give every machine the bits below its own field and lane 0 the sequence's bit total, and the
                            // 8 bytes that hold the update bits are read together with the entries from the cursor alone ([B - 56.., B));
                            // only a sequence that reads more than 56 bits needs a second, dependent read.  The scalar unit - ONE per CU,
                            // shared by every wave - keeps just the cursor and the loop.  The states go to LDS (rec) for pass 2; an
                            // over-read shows as a negative cursor (collected in `bad`, checked once per group) and is clamped so that no
                            // load leaves the window.  The last sequence of a block reads no update bits: peeled off the loop.
                            uint32_t bad = 0;
                            const uint32_t Bgroup = B;
                            const uint32_t upd = g + cnt < nbSeq ? cnt : cnt - 1;
                            uint32_t j = 0;
                            for (; j + 2 <= upd; j += 2) {                                    // two steps per trip: rec offsets become immediates, half the loop control
                                seq_chain_step(tbl, recp + j * 4, win, wbase, st, B, bad);
                                seq_chain_step(tbl, recp + j * 4 + 4, win, wbase, st, B, bad);
                            }
                            if (j < upd) seq_chain_step(tbl, recp + j * 4, win, wbase, st, B, bad);
                            if (upd < cnt) {
                                const uint32_t e_ = tbl[st];
                                recp[upd * 4] = (uint16_t)st;
                                const uint32_t eb = SEQD_EBITS(e_);
                                const int32_t raw = (int32_t)(B - DUNI(eb + DPP_SHL(eb, 1) + DPP_SHL(eb, 2)));
                                bad |= (uint32_t)raw;
                                B = (uint32_t)(raw < 0 ? 0 : raw);
                            }
                            e = bad >> 31;
                            if (e) RFAIL(DERR_FRAME);                                // the stream is shorter than its sequences need
                            __threadfence_block();
                            WAVE_SYNC();
                            // pass 2: every lane decodes the fields of its own sequence from the window; its cursor is the group's minus
                            // the bits of the sequences before it (prefix sum)
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
                                    const uint32_t lo1 = Bgroup - (incl - mine) - oc;           // offset bits first (<= 31), then ML, then LL (<= 16 each)
                                    offBase = (1u << oc) + ((uint32_t)(wld64(win, wbase, lo1 >> 3) >> (lo1 & 7)) & ((1u << oc) - 1));
                                    const uint32_t lo2 = lo1 - mbits - lbits;
                                    const uint32_t w2 = (uint32_t)(wld64(win, wbase, lo2 >> 3) >> (lo2 & 7));
                                    ll = lbase + (w2 & ((1u << lbits) - 1));
                                    ml = mbase + ((w2 >> lbits) & ((1u << mbits) - 1));
                                }
                            }
                            // pass 3: repeat offsets.  A sequence with a new offset (code > 3) knows it already and only pushes it onto the
                            // history; the scalar loop visits just the sequences that USE the history (codes 1..3), in order, first
                            // folding in the new offsets pushed since the previous visit (only the last three matter).  Code c names
                            // history entry idx = c - 1 (+ 1 when the literal length is 0; idx 3 = rep0 - 1); idx >= 2 pushes the whole
                            // history down, idx 1 swaps the first two, idx 0 leaves it alone.  All on wave-uniform values, no branches.
                            uint32_t off = offBase - 3;
                            {
                                const unsigned long long ll0 = __ballot(valid && ll == 0);
                                unsigned long long users = __ballot(valid && offBase <= 3);
                                uint32_t r0 = DUNI(rep0), r1 = DUNI(rep1), r2 = DUNI(rep2);
                                uint32_t prev = 0;                                                  // first sequence not folded in yet
                                for (;;) {
                                    const uint32_t j = users ? (uint32_t)__ffsll((long long)users) - 1 : cnt;      // next user, or the group's end
                                    const uint32_t gap = j - prev;                                  // new offsets pushed by sequences [prev, j)
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
                                    const uint32_t idx = ob - 1 + (uint32_t)((ll0 >> j) & 1);       // 0..3
                                    const uint32_t c01 = idx == 0 ? r0 : r1, c23 = idx == 2 ? r2 : r0 - 1;
                                    const uint32_t o_ = idx < 2 ? c01 : c23;
                                    r2 = idx >= 2 ? r1 : r2;
                                    r1 = idx >= 1 ? r0 : r1;
                                    r0 = o_;
                                    off = tsx_writelane(o_, j, off);
                                    prev = j + 1;
                                }
                                rep0 = r0; rep1 = r1; rep2 = r2;
                            }
                            if (valid) { sLL[g + lane] = ll; sML[g + lane] = ml; sOF[g + lane] = off; }
                            DLT(2);                                                 // 2: FSE sequence decode
                        }
                        TSX_SETPRIO(0);
                        if (nbSeq && B != 0) RFAIL(DERR_FRAME);                     // BIT_endOfDStream: every bit of the stream was used
                    }
                    if (lane == 0) L.nseq[(it - 1) & 1] = nbSeq;
                }
            }
        }
block_end:
        if (myErr != TSX_OK && lane == 0) L.err = myErr;
        DLT(4);                                                         // 4: this wave's work outside the named phases
        __threadfence_block();
        __syncthreads();
        DLT(5);                                                         // 5: waiting for the other wave
        const int32_t posted = (int32_t)DUNI(L.err);
        if (posted != TSX_OK) { err = posted; break; }
        if (role == 0 ? fseDone : role == 1 ? prodDone : frameDone) break;     // a wave that has nothing left to do leaves; the barrier counts live waves
    }
    if (role == 2 && err == TSX_OK && opos != contentSize) err = DERR_FRAME;
done:
#ifdef TSX_PROF2
    if (lane == 0 && dprof) {                                           // laps: wave 1 -> 0 (literals), 6 (wait); wave 2 -> 3 (execution), 5 (wait); wave 0 -> 1, 2, 4, 7 (wait)
        unsigned long long* const dp = dprof + (size_t)chunk * 8;
        if (role == 1) { dp[0] = dlt_[0] + dlt_[4]; dp[6] = dlt_[5]; }
        else if (role == 2) { dp[3] = dlt_[3] + dlt_[4]; dp[5] = dlt_[5]; }
        else { dp[1] = dlt_[1]; dp[2] = dlt_[2]; dp[4] = dlt_[4]; dp[7] = dlt_[5]; }
    }
#endif
    if (role == 2 && lane == 0) {
        if (err != TSX_OK) { status[chunk] = err; descs[chunk].dst_len = 0; }
        else descs[chunk].dst_len = opos;
    }
}

#ifdef TSX_PROF2
#define ZD_PROF_PARAM , unsigned long long* __restrict__ dprof
#define ZD_PROF_ARG , dprof
#else
#define ZD_PROF_PARAM
#define ZD_PROF_ARG
#endif
// Two builds of the same body.  The batch decoder is shaped for residency: 6 waves per SIMD (80 VGPRs, a few spilled to scratch).  The one
// that runs BEHIND the block-parallel form of a fetch (skip list: it only decodes what that form handed back) must not touch scratch at
// all: a queue's first scratch-using dispatch makes the runtime (re)size that queue's scratch, and while the compressor service's
// long-lived kernel holds its own (large) scratch that request waits for the kernel to end - measured: the first fetch after uploads
// began took 18.6 s, every later one 4 ms (gpurun r05a).  No fetch-path kernel uses scratch (tests/test_boundary.py checks the code object).
__global__ __launch_bounds__(3 * LANES) __attribute__((amdgpu_waves_per_eu(6, 6))) void zstd_decompress_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                                tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ dst_base,
                                                                int32_t* __restrict__ status, uint8_t* __restrict__ work,
                                                                const uint32_t* __restrict__ skip, uint32_t skip_stride ZD_PROF_PARAM) {
    __shared__ DecLds L;
    zstd_decompress_body(L, frames, from_mid, mid_stride, descs, dst_base, status, work, skip, skip_stride ZD_PROF_ARG);
}
__global__ __launch_bounds__(3 * LANES) void zstd_decompress_fallback_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                                tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ dst_base,
                                                                int32_t* __restrict__ status, uint8_t* __restrict__ work,
                                                                const uint32_t* __restrict__ skip, uint32_t skip_stride ZD_PROF_PARAM) {
    __shared__ DecLds L;
    zstd_decompress_body(L, frames, from_mid, mid_stride, descs, dst_base, status, work, skip, skip_stride ZD_PROF_ARG);
}

uint32_t tsx_launch_zstd_decompress(hipStream_t st, const tsx_zstd_consts* /*d_zc*/, const uint8_t* frames, int from_mid, uint64_t mid_stride,
                                    tsx_chunk_desc* d_descs, uint32_t n, uint8_t* dst, int32_t* d_status, void* d_work,
                                    const uint32_t* skip, uint32_t skip_stride, bool no_scratch) {
    if (!n) return 0;
    if (skip || no_scratch) hipLaunchKernelGGL(zstd_decompress_fallback_kernel, dim3(n), dim3(3 * LANES), 0, st, frames, from_mid, mid_stride, d_descs, dst, d_status, (uint8_t*)d_work, skip, skip_stride
#ifdef TSX_PROF2
                       , g_dprof_out
#endif
                       );
    else hipLaunchKernelGGL(zstd_decompress_kernel, dim3(n), dim3(3 * LANES), 0, st, frames, from_mid, mid_stride, d_descs, dst, d_status, (uint8_t*)d_work, skip, skip_stride
#ifdef TSX_PROF2
                       , g_dprof_out
#endif
                       );
    return 1;
}
