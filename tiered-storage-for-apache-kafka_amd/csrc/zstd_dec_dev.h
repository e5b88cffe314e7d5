// Device-side pieces shared by the two frame-decoder kernels (zstd_dec.hip: one workgroup per chunk, blocks pipelined;
// zstd_dec_blocks.hip: one workgroup per block, for small batches): format tables, bit readers, FSE / Huffman table builders,
// the LDS state of a decoding workgroup and the copy helpers of the execution stage.  Included by those two files only.
#pragma once
#include "zstd_common.h"

#define LANES 64
#define DERR_FRAME TSX_E_BAD_FRAME

__device__ static const uint32_t dLLbase[36] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536};
__device__ static const uint8_t dLLbits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__device__ static const uint32_t dMLbase[53] = {3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539};
__device__ static const uint8_t dMLbits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
__device__ static const short dLLnorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__device__ static const short dOFnorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
__device__ static const short dMLnorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};

__device__ static inline uint32_t dhb32(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ static inline uint64_t dld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

#define ZS_DWIN 2048u
#define ZS_DPAD 16u
#define ZS_HWIN 256u
struct FseD { uint16_t base; uint8_t sym; uint8_t nb; };
// sequence decoding entry, one dword: next-state base (bits 0-8) | nbBits (9-13) | nbBits + the symbol's number of extra
// bits (14-20) | the symbol (21-26).  Bits 9-20 are laid out so that ONE add sums both counts over the three tables (no carry:
// 3 x 9 < 32, 3 x 40 < 128).  The symbol's base value comes from a small per-code table when the lanes decode the fields.
#define DUNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))
// v_writelane_b32: a wave-uniform value lands in lane `lane` of a per-lane register (this clang has the intrinsic, not the builtin)
#ifdef HIPEMU
#define tsx_writelane(v, lane, old) __builtin_amdgcn_writelane((uint32_t)(v), (uint32_t)(lane), (uint32_t)(old))
#else
extern "C" __device__ uint32_t tsx_writelane(uint32_t v, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
#endif
typedef uint32_t SeqD;
#define SEQD(base_, nb_, ebits_, sym_) ((uint32_t)(base_) | ((uint32_t)(nb_) << 9) | ((uint32_t)((nb_) + (ebits_)) << 14) | ((uint32_t)(sym_) << 21))
#define SEQD_BASE(e_) ((e_) & 0x1FFu)
#define SEQD_NB(e_) (((e_) >> 9) & 0x1Fu)
#define SEQD_TOT(e_) (((e_) >> 14) & 0x7Fu)
#define SEQD_EBITS(e_) (SEQD_TOT(e_) - SEQD_NB(e_))
#define SEQD_SYM(e_) (((e_) >> 21) & 0x3Fu)
#define SEQD_COUNTS(e_) (((e_) >> 9) & 0xFFFu)                        /* nbBits | (nbBits + extra bits) << 5 */
#ifdef HIPEMU
#define TSX_SCHED_BARRIER() do {} while (0)
#define TSX_SETPRIO(p_) do {} while (0)
#else
#define TSX_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define TSX_SETPRIO(p_) __builtin_amdgcn_s_setprio(p_)          /* issue priority of this wave among the SIMD's waves, 0..3 */
#endif
// row_shl:n - lane i reads lane i + n of its row of 16, 0 past the row's end
#define DPP_SHL(v_, n_) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v_), 0x100 + (n_), 0xF, 0xF, true))
// what the literal wave hands to the sequence wave for one block
struct BlkDesc { uint32_t off, bsize, btype, last, litInPlace, litOff, litSize, q; };
// Wave-level rendezvous: the lanes of ONE wave run in lockstep, so this is only a memory fence + a compiler barrier on the
// device (a workgroup barrier here would wait for the chunk's other wave, which is somewhere else entirely).
#ifdef HIPEMU
#define WAVE_SYNC() hipemu::wave_barrier()
#else
#define WAVE_SYNC() do { __threadfence_block(); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
struct DecLds {
    // Literal Huffman table, indexed by the next 11 bits of a stream (Max_Number_of_Bits of a literals tree is 11, RFC 8878
    // 4.2.1): up to three whole symbols that those bits decode to | total bits << 24 | number of symbols << 28
    uint32_t hufX[2048];
    uint16_t hufRs[16], hufSymStart[16];   // canonical form: first table index / first entry of hufSorted of each weight
    uint8_t hufSorted[256];                // symbols by (weight, symbol)
    uint32_t hufLog; int hufValid;
    SeqD ll[512], of[256], ml[512];
    SeqD zeroEntry;              // the 'table' of the lanes that run no state machine
    alignas(8) uint16_t rec[LANES * 4];   // pass 1 -> pass 2: the three states of each of the group's 64 sequences
    uint8_t cellSym[512];        // table construction scratch: symbol of each cell
    uint32_t cLLbase[36], cMLbase[53]; uint8_t cLLbits[36], cMLbits[53];   // LDS copies of the length code tables
    FseD wt[64];                 // FSE table of the Huffman-weight stream (tableLog <= 6)
    uint32_t llLog, ofLog, mlLog; int llValid, ofValid, mlValid;
    uint8_t weights[256];
    uint32_t rankCount[16], rankStart[16];
    short norm[64];
    uint16_t symNext[64];
    uint32_t scal[16];
    uint32_t streamOff[5];
    // the literal wave's own scratch and windows (the two waves of a chunk run concurrently)
    short normH[16]; uint16_t symNextH[16]; uint8_t cellSymH[64]; uint32_t scalH[4];
    BlkDesc desc[3];             // block k in slot k % 3: written by the literal stage, read by the sequence and execution stages
    uint32_t nseq[2];            // sequence stage -> execution stage: number of sequences of block k in slot k & 1
    int32_t err;                 // first error of either wave
    alignas(16) uint8_t hwin[4 * (ZS_HWIN + 16)];   // one window per Huffman stream
    alignas(16) uint8_t swin[ZS_DPAD + ZS_DWIN + 32];   // the sequence bit stream's window behind ZS_DPAD zero bytes (seq_chain_step reads up to 7 bytes in front of the stream)
};

// ---- backward bit reader (BIT_DStream) -----------------------------------------------------------------
struct BitR { const uint8_t* start; const uint8_t* ptr; uint64_t c; uint32_t consumed; bool bad; };
__device__ static void br_init(BitR& b, const uint8_t* src, uint32_t n) {
    b.start = src; b.bad = false; b.consumed = 0; b.c = 0; b.ptr = src;
    if (n == 0) { b.bad = true; return; }
    const uint8_t last = src[n - 1];
    if (last == 0) { b.bad = true; return; }
    if (n >= 8) {
        b.ptr = src + n - 8; b.c = dld64(b.ptr);
        b.consumed = 8 - dhb32(last);
    } else {
        uint64_t c = 0;
        for (uint32_t i = 0; i < n; i++) c |= (uint64_t)src[i] << (8 * i);
        b.c = c;
        b.consumed = 8 - dhb32(last) + (8 - n) * 8;
    }
}
__device__ static inline uint64_t br_look(const BitR& b, uint32_t nb) {          // nb >= 1
    return (b.c << (b.consumed & 63)) >> (64 - nb);
}
__device__ static inline uint64_t br_read(BitR& b, uint32_t nb) {
    if (!nb) return 0;
    const uint64_t v = br_look(b, nb);
    b.consumed += nb;
    return v;
}
// returns false once the stream is over-read
__device__ static inline bool br_reload(BitR& b) {
    if (b.consumed > 64) return false;
    if (b.ptr >= b.start + 8) { b.ptr -= b.consumed >> 3; b.consumed &= 7; b.c = dld64(b.ptr); return true; }
    if (b.ptr == b.start) return true;
    uint32_t nbBytes = b.consumed >> 3;
    if (b.ptr - nbBytes < b.start) nbBytes = (uint32_t)(b.ptr - b.start);
    b.ptr -= nbBytes; b.consumed -= nbBytes * 8;
    // fewer than 8 bytes may remain readable behind ptr only when the stream itself is shorter than 8 bytes
    b.c = dld64(b.ptr);
    return true;
}

// An 8-byte read from a window of a bit stream held in LDS: win[0 ..) = stream bytes [wbase ..), positions are offsets in the
// stream, so no pointer ever leaves the window array.
__device__ static inline uint64_t wld64(const uint8_t* win, uint32_t wbase, uint32_t pos) { return dld64(win + (pos - wbase)); }

// One step of the sequence chain (pass 1 of the sequence stage in zstd_dec.hip and zstd_dec_blocks.hip), on the vector unit.  Lanes 0, 1, 2
// run the LL, ML and OF state machines (that is the order in which a sequence's state-update bits sit in the stream, highest first); the
// other lanes carry state 0 through an all-zero entry.  One table read serves all three, two DPP adds give every machine the bits below
// its own field and lane 0 the sequence's bit total, and the 8 bytes that hold the update bits are read together with the entries from
// the cursor alone ([B - 56.., B)): only a sequence that reads more than 56 bits needs a second, dependent read.  win = the window
// behind its ZS_DPAD zero bytes: near the stream's start the 8 bytes begin up to 7 bytes in front of it, which costs no select.  The
// state goes to *rec for pass 2; an over-read shows as a negative cursor (collected in bad, checked once per group) and is clamped so that
// no load leaves the window.  The wave is alone on its SIMD, so a step costs its instruction count (~5 cycles each) plus one LDS latency:
// profiles/r03_zb_phase_laps.txt.
__device__ __forceinline__ void seq_chain_step(const SeqD* __restrict__ tbl, uint16_t* __restrict__ rec, const uint8_t* __restrict__ win, uint32_t wbase,
                                               uint32_t& st, uint32_t& B, uint32_t& bad) {
    const int32_t p8 = (int32_t)(B >> 3) - 7;                           // >= wbase when wbase != 0 (the window's margin), >= -7 else
    const uint32_t e_ = tbl[st];                                        // first: LDS answers in order, and this is the read the chain waits for
    uint64_t c8 = dld64(win + (p8 - (int32_t)wbase));
    *rec = (uint16_t)st;
    TSX_SCHED_BARRIER();                                                // both reads are in flight before anything waits
    const uint32_t pc = SEQD_COUNTS(e_);
    const uint32_t below = DPP_SHL(pc, 1) + DPP_SHL(pc, 2);             // the machines below this one
    // extra bits of the offset, match length, literal length, then the state updates: LL, ML, OF (ZSTD_decodeSequence order)
    const int32_t raw = (int32_t)(B - DUNI((pc + below) >> 5));
    bad |= (uint32_t)raw;
    const uint32_t lo = (uint32_t)(raw < 0 ? 0 : raw);
    int32_t sh = (int32_t)lo - 8 * p8;
    if (__builtin_expect(sh < 0, 0)) { c8 = wld64(win, wbase, lo >> 3); sh = (int32_t)(lo & 7); }       // rare
    st = SEQD_BASE(e_) + ((uint32_t)(c8 >> ((uint32_t)sh + (below & 31))) & ((1u << (pc & 31)) - 1));
    B = lo;
}

// ---- FSE table description + decoding table (lane 0) ---------------------------------------------------------
// returns bytes consumed, 0 on error
// (STORE = false walks the description without keeping the counts: how long it is, and whether it is well formed)
template <bool STORE>
__device__ static uint32_t fse_walkNCount(short* norm, uint32_t* maxSymPtr, uint32_t* tableLogPtr, const uint8_t* src, uint32_t n, uint32_t maxLogAllowed) {
    if (n < 1) return 0;
    // bounded forward bit reader over at most n bytes
    uint64_t bitpos = 0;
    // 4 bytes at the bit cursor: one unaligned load while they are all inside the description, byte by byte (zeros past its end) otherwise
    #define NC_PEEK(k) ({ uint32_t v_ = 0; const uint64_t b0_ = bitpos >> 3; \
                          if (b0_ + 4 <= n) __builtin_memcpy(&v_, src + b0_, 4); \
                          else for (int i_ = 0; i_ < 4; i_++) { const uint64_t b_ = b0_ + i_; v_ |= (uint32_t)(b_ < n ? src[b_] : 0) << (8 * i_); } \
                          (v_ >> (bitpos & 7)) & ((1u << (k)) - 1); })
    const uint32_t tableLog = NC_PEEK(4) + 5; bitpos += 4;
    if (tableLog > maxLogAllowed) return 0;
    int remaining = (1 << tableLog) + 1, threshold = 1 << tableLog, nbBits = (int)tableLog + 1;
    uint32_t sym = 0; const uint32_t maxSym = *maxSymPtr;
    bool prev0 = false;
    while (remaining > 1 && sym <= maxSym) {
        if (prev0) {
            for (;;) {
                const uint32_t r = NC_PEEK(2); bitpos += 2;
                for (uint32_t k = 0; k < r; k++) { if (sym > maxSym) return 0; if (STORE) norm[sym] = 0; sym++; }
                if (r != 3) break;
                if ((bitpos >> 3) > n + 4) return 0;
            }
            prev0 = false;
            continue;
        }
        const int mx = (2 * threshold - 1) - remaining;
        int count;
        const uint32_t lo = NC_PEEK(nbBits - 1);
        if ((int)lo < mx) { count = (int)lo; bitpos += nbBits - 1; }
        else { count = (int)NC_PEEK(nbBits); if (count >= threshold) count -= mx; bitpos += nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        if (sym > maxSym) return 0;
        if (STORE) norm[sym] = (short)count;
        sym++;
        prev0 = count == 0;
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
        if ((bitpos >> 3) > n + 4) return 0;
    }
    #undef NC_PEEK
    if (remaining != 1) return 0;
    const uint32_t used = (uint32_t)((bitpos + 7) >> 3);
    if (used > n) return 0;
    *maxSymPtr = sym - 1; *tableLogPtr = tableLog;
    return used;
}
__device__ static inline uint32_t fse_readNCount(short* norm, uint32_t* maxSymPtr, uint32_t* tableLogPtr, const uint8_t* src, uint32_t n, uint32_t maxLogAllowed) {
    return fse_walkNCount<true>(norm, maxSymPtr, tableLogPtr, src, n, maxLogAllowed);
}
__device__ static inline uint32_t fse_skipNCount(uint32_t maxSym, const uint8_t* src, uint32_t n, uint32_t maxLogAllowed) {
    uint32_t ms = maxSym, tl = 0;
    return fse_walkNCount<false>(nullptr, &ms, &tl, src, n, maxLogAllowed);
}

template <class E, class Fill>
__device__ static bool fse_buildDTable(E* dt, uint8_t* cellSym, const short* norm, uint32_t maxSym, uint32_t tableLog, uint16_t* symNext, Fill fill) {
    const uint32_t size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint32_t high = size - 1;
    for (uint32_t s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) { cellSym[high--] = (uint8_t)s; symNext[s] = 1; }
        else symNext[s] = (uint16_t)norm[s];
    }
    uint32_t pos = 0;
    for (uint32_t s = 0; s <= maxSym; s++)
        for (int i = 0; i < norm[s]; i++) {
            cellSym[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    if (pos != 0) return false;
    for (uint32_t u = 0; u < size; u++) {
        const uint8_t s = cellSym[u];
        const uint32_t ns = symNext[s]++;
        const uint32_t nb = tableLog - dhb32(ns);
        E e; e.nb = (uint8_t)nb; e.base = (uint16_t)((ns << nb) - size);
        fill(e, s);
        dt[u] = e;
    }
    return true;
}
// kind 0 = literal lengths, 1 = offsets, 2 = match lengths: extra bits / base value of a code
__device__ static inline uint32_t seq_ebits(const DecLds& L, uint32_t sym, int kind) { return kind == 0 ? L.cLLbits[sym] : kind == 1 ? sym : L.cMLbits[sym]; }
// The same table (FSE_buildDTable: spread the symbols with the odd stride `step`, number each symbol's cells in ascending
// position) built by the whole wave instead of one lane walking 2 x 512 cells through dependent LDS accesses:
//  * the spread visits the cells in the order (i * step) & mask, i = 0, 1, ..., skipping the cells above `high` that the
//    low-probability symbols (count -1) own; so cell u is the r-th one filled, r = i(u) - #{low cells visited before},
//    i(u) = u * step^-1 mod size, and its symbol is the one whose run of the cumulative counts holds r;
//  * cells are numbered 64 at a time in ascending position: a cell's state number is its symbol's running count (kept in
//    lane `symbol`) plus its rank among the same-symbol lanes of the round.
// norm[] is in LDS (L.norm), maxSym < 64.  Returns false (wave-uniform) when the counts do not fill the table.
__device__ static bool fse_buildSeqTable_wave(SeqD* dt, DecLds& L, uint32_t maxSym, uint32_t tableLog, int kind, uint32_t lane) {
    const uint32_t size = 1u << tableLog, mask = size - 1, step = (size >> 1) + (size >> 3) + 3;
    uint16_t* const cum = (uint16_t*)L.cellSym;                           // [64] exclusive prefix of the positive counts
    uint16_t* const lowI = cum + 64;                                      // [<= 64] visit index of each low-probability cell
    uint8_t* const lowSym = L.cellSym + 256;                              // [<= 64] symbol of the k-th low-probability cell
    const int nv = lane <= maxSym ? (int)L.norm[lane] : 0;
    const uint32_t c = nv > 0 ? (uint32_t)nv : 0;
    const bool lowp = nv == -1;
    uint32_t incl = c;
    for (uint32_t o = 1; o < LANES; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, LANES - 1);
    const unsigned long long lowMask = __ballot(lowp);
    const uint32_t nLow = (uint32_t)__popcll(lowMask);
    if (total + nLow != size) return false;
    uint32_t inv = step;                                                  // odd: step * step = 1 (mod 8); each Newton step doubles the bits
    for (int it = 0; it < 3; it++) inv *= 2u - step * inv;
    inv &= mask;
    WAVE_SYNC();                                                          // the scratch may still be read as the previous table's
    cum[lane] = (uint16_t)(incl - c);
    if (lowp) {
        const uint32_t k = (uint32_t)__popcll(lowMask & ((1ull << lane) - 1));   // low cells go to the top, in symbol order
        lowI[k] = (uint16_t)(((size - 1 - k) * inv) & mask); lowSym[k] = (uint8_t)lane;
    }
    WAVE_SYNC();
    uint32_t next = lowp ? 1u : c;                                        // symNext of symbol `lane`
    const uint32_t high = size - 1 - nLow;
    for (uint32_t base = 0; base < size; base += LANES) {
        const uint32_t u = base + lane;
        const bool active = u < size;
        uint32_t sym = 0;
        if (active) {
            if (u > high) sym = lowSym[size - 1 - u];
            else {
                const uint32_t i = (u * inv) & mask;
                uint32_t r = i;
                for (uint32_t k = 0; k < nLow; k++) r -= lowI[k] < i ? 1u : 0u;
                for (uint32_t b = 32; b; b >>= 1) if (cum[sym + b] <= r) sym += b;     // the last symbol whose run starts at or before r
            }
        }
        uint32_t ns = 1;
        unsigned long long rem = __ballot(active);
        while (rem) {
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane(sym, __ffsll((long long)rem) - 1);     // v_readlane: no LDS round trip
            const unsigned long long m = __ballot(active && sym == cur);
            const uint32_t first = (uint32_t)__builtin_amdgcn_readlane(next, (int)cur);
            if (active && sym == cur) ns = first + (uint32_t)__popcll(m & ((1ull << lane) - 1));
            if (lane == cur) next += (uint32_t)__popcll(m);
            rem &= ~m;
        }
        if (active) {
            const uint32_t nb = tableLog - dhb32(ns);
            dt[u] = SEQD((ns << nb) - size, nb, seq_ebits(L, sym, kind), sym);
        }
    }
    WAVE_SYNC();
    return true;
}
__device__ static uint32_t huf_readTable(DecLds& L, const uint8_t* src, uint32_t n) {
    if (n < 1) return 0;
    const uint32_t hb = src[0];
    uint32_t nw = 0, used;
    if (hb >= 128) {
        nw = hb - 127; used = 1 + (nw + 1) / 2;
        if (used > n) return 0;
        for (uint32_t i = 0; i < nw; i++) { const uint8_t b = src[1 + i / 2]; L.weights[i] = (i & 1) ? (b & 15) : (b >> 4); }
    } else {
        used = 1 + hb;
        if (hb < 2 || used > n) return 0;
        uint32_t maxSym = 12, tl;
        const uint32_t h = fse_readNCount(L.normH, &maxSym, &tl, src + 1, hb, 6);
        if (!h) return 0;
        if (!fse_buildDTable(L.wt, L.cellSymH, L.normH, maxSym, tl, L.symNextH, [](FseD& e, uint32_t sym) { e.sym = (uint8_t)sym; })) return 0;
        BitR b; br_init(b, src + 1 + h, hb - h);
        if (b.bad) return 0;
        uint32_t s1 = (uint32_t)br_read(b, tl), s2 = (uint32_t)br_read(b, tl);
        br_reload(b);
        for (;;) {
            if (nw > 253) return 0;
            { const FseD e = L.wt[s1]; L.weights[nw++] = e.sym; s1 = e.base + (uint32_t)br_read(b, e.nb); }
            if (!br_reload(b)) { L.weights[nw++] = L.wt[s2].sym; break; }
            if (nw > 253) return 0;
            { const FseD e = L.wt[s2]; L.weights[nw++] = e.sym; s2 = e.base + (uint32_t)br_read(b, e.nb); }
            if (!br_reload(b)) { L.weights[nw++] = L.wt[s1].sym; break; }
        }
    }
    // the last weight is implicit
    uint32_t total = 0;
    for (uint32_t i = 0; i < 16; i++) L.rankCount[i] = 0;
    for (uint32_t i = 0; i < nw; i++) { const uint32_t w = L.weights[i]; if (w > 12) return 0; L.rankCount[w]++; total += w ? (1u << (w - 1)) : 0; }
    if (total == 0) return 0;
    const uint32_t tableLog = dhb32(total) + 1;
    if (tableLog > 11) return 0;
    const uint32_t rest = (1u << tableLog) - total;
    if (rest & (rest - 1)) return 0;
    const uint32_t lastW = dhb32(rest) + 1;
    L.weights[nw++] = (uint8_t)lastW; L.rankCount[lastW]++;
    if (L.rankCount[1] < 2 || (L.rankCount[1] & 1)) return 0;
    // canonical layout: weight-1 symbols (the longest codes, tableLog bits) own the lowest table indices, one index each; a
    // symbol of weight w owns 1 << (w - 1) consecutive indices and is coded on tableLog + 1 - w bits
    uint32_t next = 0, cnt = 0;
    for (uint32_t w = 1; w <= tableLog; w++) {
        L.hufRs[w] = (uint16_t)next; next += L.rankCount[w] << (w - 1);
        L.hufSymStart[w] = (uint16_t)cnt; L.rankStart[w] = cnt; cnt += L.rankCount[w];
    }
    L.hufRs[tableLog + 1] = (uint16_t)next;
    for (uint32_t sy = 0; sy < nw; sy++) { const uint32_t w = L.weights[sy]; if (w) L.hufSorted[L.rankStart[w]++] = (uint8_t)sy; }
    L.hufLog = tableLog; L.hufValid = 1;
    return used;
}

// One symbol from the canonical form: idx = the next tableLog bits (zero-padded below the stream's first bit).  -> symbol | nbBits << 8
__device__ static inline uint32_t huf_decode1(const DecLds& L, uint32_t idx, uint32_t tableLog) {
    uint32_t w = 1;
    for (uint32_t ww = 2; ww <= tableLog; ww++) if (L.hufRs[ww] <= idx) w = ww;          // empty weights share their start with the next one
    const uint32_t sym = L.hufSorted[L.hufSymStart[w] + ((idx - L.hufRs[w]) >> (w - 1))];
    return sym | ((tableLog + 1 - w) << 8);
}
// The 11-bit multi-symbol table, built by the whole wave: entry x = the (up to three) symbols whose codes fit entirely in x.
__device__ static void huf_buildX_wave(DecLds& L, uint32_t lane) {
    const uint32_t tl = L.hufLog, mask = (1u << tl) - 1;
    for (uint32_t x = lane; x < 2048; x += LANES) {
        uint32_t pos = 0, ns = 0, syms = 0;
        for (uint32_t k = 0; k < 3 && pos < 11; k++) {
            const uint32_t rem = 11 - pos;
            const uint32_t idx = rem >= tl ? (x >> (rem - tl)) & mask : (x << (tl - rem)) & mask;
            const uint32_t e = huf_decode1(L, idx, tl);
            if ((e >> 8) > rem) break;                                 // this code runs past the 11 bits
            syms |= (e & 0xFF) << (8 * k); pos += e >> 8; ns++;
        }
        L.hufX[x] = syms | (pos << 24) | (ns << 28);
    }
}

// Per-lane copy of a short, non-overlapping run (a literal run, or a match whose source is already final): up to four
// 8-byte loads are in flight before the first store, so a run of <= 32 bytes costs one memory round trip.
__device__ static __forceinline__ void copy_small(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n) {
    uint32_t k = 0;
    while (n - k >= 32) {
        const uint64_t a = dld64(src + k), b = dld64(src + k + 8), c = dld64(src + k + 16), d = dld64(src + k + 24);
        __builtin_memcpy(dst + k, &a, 8); __builtin_memcpy(dst + k + 8, &b, 8); __builtin_memcpy(dst + k + 16, &c, 8); __builtin_memcpy(dst + k + 24, &d, 8);
        k += 32;
    }
    const uint32_t r = n - k;                                       // < 32
    uint64_t q0 = 0, q1 = 0, q2 = 0; uint32_t w = 0; uint16_t h = 0; uint8_t b1 = 0;
    const uint32_t nq = r >> 3;
    if (nq > 0) q0 = dld64(src + k);
    if (nq > 1) q1 = dld64(src + k + 8);
    if (nq > 2) q2 = dld64(src + k + 16);
    uint32_t t = k + 8 * nq;
    if (r & 4) { __builtin_memcpy(&w, src + t, 4); }
    if (r & 2) { __builtin_memcpy(&h, src + t + (r & 4), 2); }
    if (r & 1) { b1 = src[t + (r & 6)]; }
    if (nq > 0) __builtin_memcpy(dst + k, &q0, 8);
    if (nq > 1) __builtin_memcpy(dst + k + 8, &q1, 8);
    if (nq > 2) __builtin_memcpy(dst + k + 16, &q2, 8);
    if (r & 4) __builtin_memcpy(dst + t, &w, 4);
    if (r & 2) __builtin_memcpy(dst + t + (r & 4), &h, 2);
    if (r & 1) dst[t + (r & 6)] = b1;
}

// Non-overlapping copy by the whole wave (every lane calls it with the same arguments).
__device__ static __forceinline__ void copy_wave(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, uint32_t lane) {
    for (uint32_t k = lane * 8; k + 8 <= n; k += LANES * 8) { const uint64_t v = dld64(src + k); __builtin_memcpy(dst + k, &v, 8); }
    const uint32_t t = n & ~7u;
    if (lane < (n & 7)) dst[t + lane] = src[t + lane];
}
#define ZS_LONG_RUN 128u
#define ZS_DSEQ_CAP 43712u            /* >= 128 KiB / 3 sequences per block; 3 x 4 x cap bytes fit ZS_WS_SEQS and ZS_WS_STBITS.. */
static_assert(12u * ZS_DSEQ_CAP <= 16u * (ZS_MAX_SEQ + 64) && 12u * ZS_DSEQ_CAP <= 6u * ZS_WS_CODE_STRIDE + ZS_BLOCKOUT_CAP, "sequence arrays fit the workspace regions they borrow");
static_assert(ZS_WS_HASHLONG + (192u << 10) + ZS_BLOCK_MAX + 256 <= ZS_WS_SEQS, "literal buffers fit the hash-table region");
// One run per lane (mine = this lane has one): the short ones all at once, each by its own lane; the long ones one after the
// other, each by the whole wave (a single lane would spend one memory round trip per 32 bytes on them).
__device__ static __forceinline__ void exec_copies(uint8_t* dst, const uint8_t* src, uint32_t n, bool mine, uint32_t lane) {
    const bool big = mine && n > ZS_LONG_RUN;
    if (mine && !big) copy_small(dst, src, n);
    unsigned long long bigm = __ballot(big);
    const uint64_t d64 = (uint64_t)dst, s64 = (uint64_t)src;
    while (bigm) {
        const int i = __ffsll((long long)bigm) - 1;
        bigm &= bigm - 1;
        // readlane returns int: every half goes through uint32_t before it is widened (an OR-ed int sign-extends)
        const uint64_t d = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(d64 >> 32), i) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)d64, i);
        const uint64_t s_ = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)(s64 >> 32), i) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((uint32_t)s64, i);
        copy_wave((uint8_t*)d, (const uint8_t*)s_, (uint32_t)__builtin_amdgcn_readlane(n, i), lane);
    }
}
