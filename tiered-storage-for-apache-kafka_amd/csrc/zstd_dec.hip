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
#include "zstd_common.h"

#define LANES 64
#define DERR_FRAME TSX_E_BAD_FRAME
#ifdef TSX_PROF2
static unsigned long long* g_dprof_out = nullptr;                     // 8 u64 per chunk: phase laps of the decoder
extern "C" void tsx_debug_set_dprof(void* dev_ptr) { g_dprof_out = (unsigned long long*)dev_ptr; }
#define DLT(k) do { const unsigned long long n_ = (unsigned long long)clock64(); dlt_[k] += n_ - dlast_; dlast_ = n_; } while (0)
#else
#define DLT(k) do {} while (0)
#endif

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
    alignas(16) uint8_t swin[ZS_DWIN + 32];        // the sequence bit stream's window
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

// ---- FSE table description + decoding table (lane 0) ---------------------------------------------------------
// returns bytes consumed, 0 on error
__device__ static uint32_t fse_readNCount(short* norm, uint32_t* maxSymPtr, uint32_t* tableLogPtr, const uint8_t* src, uint32_t n, uint32_t maxLogAllowed) {
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
                for (uint32_t k = 0; k < r; k++) { if (sym > maxSym) return 0; norm[sym++] = 0; }
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
        norm[sym++] = (short)count;
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

// ---- the kernel -------------------------------------------------------------------------------------------
#define FAIL(code) do { err = (code); goto done; } while (0)                 /* both waves, before the block loop */
#define RFAIL(code) do { myErr = (code); goto block_end; } while (0)        /* one wave, inside its role */

__global__ __launch_bounds__(3 * LANES) __attribute__((amdgpu_waves_per_eu(6, 6))) void zstd_decompress_kernel(const uint8_t* __restrict__ frames, int from_mid, uint64_t mid_stride,
                                                                tsx_chunk_desc* __restrict__ descs, uint8_t* __restrict__ dst_base,
                                                                int32_t* __restrict__ status, uint8_t* __restrict__ work
#ifdef TSX_PROF2
                                                                , unsigned long long* __restrict__ dprof
#endif
                                                                ) {
    __shared__ DecLds L;
    const uint32_t lane = threadIdx.x & (LANES - 1), role = DUNI(threadIdx.x >> 6), chunk = blockIdx.x;   // wave 0: sequence streams, wave 1: literals, wave 2: execution
    if (status[chunk] != TSX_OK) return;
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
                        const uint8_t* const win = L.swin;
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
                                    *reinterpret_cast<uint4*>(&L.swin[k]) = v;
                                }
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
                            for (uint32_t j = 0; j < upd; j++) {
                                const uint32_t p8 = (B >> 3) > 7 ? (B >> 3) - 7 : 0;                            // >= wbase: the window's margin
                                uint64_t c8 = wld64(win, wbase, p8);
                                const uint32_t e_ = tbl[st];
                                recp[j * 4] = (uint16_t)st;
                                TSX_SCHED_BARRIER();                                                           // both reads are in flight before anything waits
                                const uint32_t pc = SEQD_COUNTS(e_);
                                const uint32_t qc = pc + DPP_SHL(pc, 1) + DPP_SHL(pc, 2);                       // own + the machines below
                                // extra bits of the offset, match length, literal length, then the state updates: LL, ML, OF (ZSTD_decodeSequence order)
                                const int32_t raw = (int32_t)(B - DUNI(qc >> 5));
                                bad |= (uint32_t)raw;
                                const uint32_t lo = (uint32_t)(raw < 0 ? 0 : raw);
                                uint32_t sh = lo - 8 * p8;
                                if (lo < 8 * p8) { c8 = wld64(win, wbase, lo >> 3); sh = lo & 7; }             // rare
                                st = SEQD_BASE(e_) + ((uint32_t)(c8 >> (sh + ((qc - pc) & 31))) & ((1u << (pc & 31)) - 1));
                                B = lo;
                            }
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

uint32_t tsx_launch_zstd_decompress(hipStream_t st, const tsx_zstd_consts* /*d_zc*/, const uint8_t* frames, int from_mid, uint64_t mid_stride,
                                    tsx_chunk_desc* d_descs, uint32_t n, uint8_t* dst, int32_t* d_status, void* d_work) {
    if (!n) return 0;
    hipLaunchKernelGGL(zstd_decompress_kernel, dim3(n), dim3(3 * LANES), 0, st, frames, from_mid, mid_stride, d_descs, dst, d_status, (uint8_t*)d_work
#ifdef TSX_PROF2
                       , g_dprof_out
#endif
                       );
    return 1;
}
