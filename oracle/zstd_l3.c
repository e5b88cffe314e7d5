/*
 * ORACLE (test infrastructure only — never linked into the product library).
 *
 * Serial restatement of the Zstandard compressor exactly as the reference drives it:
 *   core/src/main/java/io/aiven/kafka/tieredstorage/transform/CompressionChunkEnumeration.java:50-63
 *   (fresh context per chunk, pledged source size, content-size flag, default level 3, no checksum/dict)
 * The arithmetic is libzstd's (zstd-jni 1.5.6-9 -> libzstd 1.5.6, core/build.gradle:29), not under
 * /root/reference.  This file restates the published algorithm of the 1.5.x one-shot path at level 3:
 *   cparams table + ZSTD_adjustCParams, frame header, 128 KiB block loop (+ the 1.5.7 pre-block splitter),
 *   ZSTD_compressBlock_doubleFast (noDict), ZSTD_entropyCompressSeqStore: Huffman literals (HUF_sort /
 *   HUF_buildTree / HUF_setMaxHeight / weight FSE header, 1 or 4 streams, treeless repeat) and FSE
 *   sequences (ZSTD_selectEncodingType simple heuristics for strategy < lazy, FSE_normalizeCount,
 *   FSE_writeNCount, FSE_buildCTable, ZSTD_encodeSequences), raw/RLE block fall-backs.
 * PINNED against the real library, byte for byte, by tests/test_oracle_zstd_l3.py (libzstd 1.5.7 via
 * oracle/zstd_ref.c, profile 1; profile 0 = the same minus the 1.5.7 pre-splitter = 1.5.6's block loop —
 * parity vs a real 1.5.6 is UNPINNED because that library is not available here).
 * It exists to validate the HIP compressor's intermediate decisions (sequences, table choices) and as
 * the readable statement of what "bit-exact" means for that kernel.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef ORC_STAT_VISIT      /* analysis hooks (tools/stats/parse_stats.c); no-ops in the oracle build */
#define ORC_STAT_VISIT() do {} while (0)
#define ORC_STAT_EVENT(type, dist) do {} while (0)
#define ORC_STAT_IMMREP() do {} while (0)
#define ORC_STAT_CAND(isLong, inwin, dist, eq) do {} while (0)
#endif
#ifndef ORC_TRACE           /* table-access trace of the parse (tools/stats/step_sim.c); no-op in the oracle build */
#define ORC_TRACE(kind, a, b, c) do {} while (0)
#endif

typedef uint8_t BYTE; typedef uint16_t U16; typedef uint32_t U32; typedef uint64_t U64; typedef int64_t S64;

#define KB *(1u << 10)
#define ZSTD_BLOCKSIZE_MAX (128 KB)
#define MINMATCH 3
#define HASH_READ_SIZE 8
#define ZSTD_REP_NUM 3
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define DefaultMaxOff 28
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8
#define LitHufLog 11
#define HUF_TABLELOG_MAX 12
#define HUF_SYMBOLVALUE_MAX 255
#define FSE_MIN_TABLELOG 5
#define FSE_MAX_TABLELOG 12
#define FSE_DEFAULT_TABLELOG 11

static U32 highbit32(U32 v) { return 31 - (U32)__builtin_clz(v); }
static U32 rd32(const BYTE* p) { U32 v; memcpy(&v, p, 4); return v; }
static U64 rd64(const BYTE* p) { U64 v; memcpy(&v, p, 8); return v; }
static U16 rd16(const BYTE* p) { U16 v; memcpy(&v, p, 2); return v; }
static void wr16(BYTE* p, U32 v) { p[0] = (BYTE)v; p[1] = (BYTE)(v >> 8); }
static void wr24(BYTE* p, U32 v) { p[0] = (BYTE)v; p[1] = (BYTE)(v >> 8); p[2] = (BYTE)(v >> 16); }
static void wr32(BYTE* p, U32 v) { wr16(p, v); wr16(p + 2, v >> 16); }

static const BYTE LL_bits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
static const BYTE ML_bits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
static const short LL_defaultNorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
static const short OF_defaultNorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
static const short ML_defaultNorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};

static U32 LLcode(U32 ll) {
    static const BYTE c[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
                               22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
    return ll > 63 ? highbit32(ll) + 19 : c[ll];
}
static U32 MLcode(U32 mlBase) {
    static const BYTE c[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
                                32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
                                40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
                                42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42};
    return mlBase > 127 ? highbit32(mlBase) + 36 : c[mlBase];
}

/* ------------------------------------------------------------------------------------------------
 * compression parameters: ZSTD_defaultCParameters[*][3] + ZSTD_adjustCParams_internal
 * ---------------------------------------------------------------------------------------------- */
typedef struct { U32 windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; } cparams_t;

void orc_l3_cparams(U64 srcSize, cparams_t* out) {
    static const cparams_t tab[4] = {
        {21, 16, 17, 1, 5, 0, 2},   /* > 256 KB  */
        {18, 16, 16, 1, 4, 0, 2},   /* <= 256 KB */
        {17, 15, 16, 2, 5, 0, 2},   /* <= 128 KB */
        {14, 14, 15, 2, 4, 0, 2},   /* <= 16 KB  */
    };
    U32 id = (srcSize <= 256 KB) + (srcSize <= 128 KB) + (srcSize <= 16 KB);
    cparams_t c = tab[id];
    U32 t = (U32)srcSize;
    U32 srcLog = t < 64 ? 6 : highbit32(t - 1) + 1;
    if (c.windowLog > srcLog) c.windowLog = srcLog;
    if (c.hashLog > c.windowLog + 1) c.hashLog = c.windowLog + 1;
    if (c.chainLog > c.windowLog) c.chainLog = c.windowLog;     /* cycleLog == chainLog for dfast */
    if (c.windowLog < 10) c.windowLog = 10;
    *out = c;
}

/* ------------------------------------------------------------------------------------------------
 * sequence store
 * ---------------------------------------------------------------------------------------------- */
typedef struct { U32 offBase; U32 litLength; U32 mlBase; } seq_t;     /* full-width lengths (no U16 + longLength trick) */
typedef struct {
    seq_t* seqs; size_t nbSeq;
    BYTE* lit; size_t litSize;
    BYTE *llCode, *mlCode, *ofCode;
} seqstore_t;

/* debug tap: tests read the sequences of every block through this */
typedef void (*orc_l3_seq_tap)(void* ctx, U32 blockIndex, const seq_t* seqs, size_t nbSeq, size_t litSize, size_t blockSize);
static orc_l3_seq_tap g_tap; static void* g_tap_ctx;
void orc_l3_set_tap(orc_l3_seq_tap f, void* ctx) { g_tap = f; g_tap_ctx = ctx; }

static void storeSeq(seqstore_t* s, size_t litLength, const BYTE* literals, U32 offBase, size_t matchLength) {
    memcpy(s->lit + s->litSize, literals, litLength);
    s->litSize += litLength;
    s->seqs[s->nbSeq].litLength = (U32)litLength;
    s->seqs[s->nbSeq].offBase = offBase;
    s->seqs[s->nbSeq].mlBase = (U32)(matchLength - MINMATCH);
    s->nbSeq++;
}

/* ------------------------------------------------------------------------------------------------
 * double-fast match finder (ZSTD_compressBlock_doubleFast_noDict_generic, 1.5.4+ pipelined form)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const BYTE* base;        /* index 0; first source byte has index 2 (ZSTD_WINDOW_START_INDEX) */
    U32 dictLimit;           /* lowest valid index of the prefix (== 2, or slid by enforceMaxDist) */
    U32* hashLong; U32* hashSmall;
    cparams_t cp;
} matchstate_t;

static size_t hash4(U32 u, U32 h) { return (u * 2654435761U) >> (32 - h); }
static size_t hash5(U64 u, U32 h) { return (size_t)(((u << (64 - 40)) * 889523592379ULL) >> (64 - h)); }
static size_t hash6(U64 u, U32 h) { return (size_t)(((u << (64 - 48)) * 227718039650203ULL) >> (64 - h)); }
static size_t hash7(U64 u, U32 h) { return (size_t)(((u << (64 - 56)) * 58295818150454627ULL) >> (64 - h)); }
static size_t hash8(U64 u, U32 h) { return (size_t)((u * 0xCF1BBCDCB7A56463ULL) >> (64 - h)); }
static size_t hashPtr(const BYTE* p, U32 hBits, U32 mls) {
    switch (mls) {
        default: case 4: return hash4(rd32(p), hBits);
        case 5: return hash5(rd64(p), hBits);
        case 6: return hash6(rd64(p), hBits);
        case 7: return hash7(rd64(p), hBits);
        case 8: return hash8(rd64(p), hBits);
    }
}
static size_t count(const BYTE* ip, const BYTE* match, const BYTE* iend) {
    const BYTE* s = ip;
    while (ip < iend && *ip == *match) { ip++; match++; }
    return (size_t)(ip - s);
}

static size_t compressBlock_doubleFast(matchstate_t* ms, seqstore_t* ss, U32 rep[ZSTD_REP_NUM], const BYTE* src, size_t srcSize) {
    const U32 mls = ms->cp.minMatch, hBitsL = ms->cp.hashLog, hBitsS = ms->cp.chainLog;
    U32* const hashLong = ms->hashLong; U32* const hashSmall = ms->hashSmall;
    const BYTE* const base = ms->base;
    const BYTE* const istart = src;
    const BYTE* ip = istart; const BYTE* anchor = istart;
    const U32 endIndex = (U32)((size_t)(istart - base) + srcSize);
    /* ZSTD_getLowestPrefixIndex(ms, endIndex, windowLog) */
    const U32 maxDistance = 1u << ms->cp.windowLog;
    const U32 lowestValid = ms->dictLimit;
    const U32 prefixLowestIndex = (endIndex - lowestValid > maxDistance) ? endIndex - maxDistance : lowestValid;
    const BYTE* const prefixLowest = base + prefixLowestIndex;
    const BYTE* const iend = istart + srcSize;
    const BYTE* const ilimit = iend - HASH_READ_SIZE;
    U32 offset_1 = rep[0], offset_2 = rep[1];
    U32 offsetSaved1 = 0, offsetSaved2 = 0;
    size_t mLength; U32 offset; U32 curr = 0;
    const size_t kStepIncr = 1 << 8;
    const BYTE* nextStep; size_t step;
    size_t hl0, hl1 = 0; U32 idxl0, idxl1 = 0;
    const BYTE *matchl0, *matchs0, *matchl1 = NULL; const BYTE* ip1;

    ip += ((ip - prefixLowest) == 0);
    {   U32 const current = (U32)(ip - base);
        U32 const windowLow = (current - lowestValid > maxDistance) ? current - maxDistance : lowestValid;
        U32 const maxRep = current - windowLow;
        if (offset_2 > maxRep) offsetSaved2 = offset_2, offset_2 = 0;
        if (offset_1 > maxRep) offsetSaved1 = offset_1, offset_1 = 0;
    }
    if (srcSize < HASH_READ_SIZE) goto _cleanup;   /* ilimit would precede istart: nothing searchable */
    while (1) {
        step = 1; nextStep = ip + kStepIncr; ip1 = ip + step;
        ORC_TRACE('R', (U32)(ip - base), 0, 0);
        if (ip1 > ilimit) goto _cleanup;
        hl0 = hashPtr(ip, hBitsL, 8); idxl0 = hashLong[hl0]; matchl0 = base + idxl0;
        do {
            const size_t hs0 = hashPtr(ip, hBitsS, mls);
            const U32 idxs0 = hashSmall[hs0];
            curr = (U32)(ip - base);
            matchs0 = base + idxs0;
            hashLong[hl0] = hashSmall[hs0] = curr;
            ORC_STAT_VISIT();
            ORC_TRACE('V', curr, (U32)hl0, (U32)hs0);
            ORC_STAT_CAND(1, idxl0 >= prefixLowestIndex, (U32)(ip - matchl0), idxl0 >= prefixLowestIndex && rd64(matchl0) == rd64(ip));
            ORC_STAT_CAND(0, idxs0 >= prefixLowestIndex, (U32)(ip - matchs0), idxs0 >= prefixLowestIndex && rd32(matchs0) == rd32(ip));
            if ((offset_1 > 0) & (rd32(ip + 1 - offset_1) == rd32(ip + 1))) {
                mLength = count(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
                ip++;
                ORC_STAT_EVENT(1, offset_1);
                ORC_TRACE('E', 1, offset_1, (U32)mLength);
                storeSeq(ss, (size_t)(ip - anchor), anchor, 1 /* REPCODE1_TO_OFFBASE */, mLength);
                goto _match_stored;
            }
            hl1 = hashPtr(ip1, hBitsL, 8);
            /* candidates AT the lowest prefix index are valid (>=): pinned by a differential case whose only table entry for
             * a 5 KiB match sits exactly at endIndex - maxDistance (tests/golden/fuzz_regress/window_edge_*.bin) */
            if (idxl0 >= prefixLowestIndex) {
                if (rd64(matchl0) == rd64(ip)) {
                    mLength = count(ip + 8, matchl0 + 8, iend) + 8;
                    offset = (U32)(ip - matchl0);
                    ORC_STAT_EVENT(2, offset);
                    ORC_TRACE('E', 2, offset, (U32)mLength);
                    while (((ip > anchor) & (matchl0 > prefixLowest)) && (ip[-1] == matchl0[-1])) { ip--; matchl0--; mLength++; }
                    goto _match_found;
                }
            }
            idxl1 = hashLong[hl1]; matchl1 = base + idxl1;
            if (idxs0 >= prefixLowestIndex) {
                if (rd32(matchs0) == rd32(ip)) goto _search_next_long;
            }
            if (ip1 >= nextStep) { step++; nextStep += kStepIncr; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1; matchl0 = matchl1;
        } while (ip1 <= ilimit);
_cleanup:
        offsetSaved2 = ((offsetSaved1 != 0) && (offset_1 != 0)) ? offsetSaved1 : offsetSaved2;
        rep[0] = offset_1 ? offset_1 : offsetSaved1;
        rep[1] = offset_2 ? offset_2 : offsetSaved2;
        return (size_t)(iend - anchor);
_search_next_long:
        /* short match found: measure it, then prefer the long match at +1 only if it is strictly longer */
        mLength = count(ip + 4, matchs0 + 4, iend) + 4;
        offset = (U32)(ip - matchs0);
        if ((idxl1 >= prefixLowestIndex) && (rd64(matchl1) == rd64(ip1))) {
            size_t const l1len = count(ip1 + 8, matchl1 + 8, iend) + 8;
            if (l1len > mLength) { ip = ip1; mLength = l1len; offset = (U32)(ip - matchl1); matchs0 = matchl1; ORC_STAT_EVENT(4, offset); ORC_TRACE('E', 4, offset, (U32)mLength); }
            else { ORC_STAT_EVENT(3, offset); ORC_TRACE('E', 3, offset, (U32)mLength); }
        } else { ORC_STAT_EVENT(3, offset); ORC_TRACE('E', 3, offset, (U32)mLength); }
        while (((ip > anchor) & (matchs0 > prefixLowest)) && (ip[-1] == matchs0[-1])) { ip--; matchs0--; mLength++; }
_match_found:
        offset_2 = offset_1; offset_1 = offset;
        if (step < 4) { hashLong[hl1] = (U32)(ip1 - base); ORC_TRACE('1', (U32)(ip1 - base), (U32)hl1, 0); }
        storeSeq(ss, (size_t)(ip - anchor), anchor, offset + ZSTD_REP_NUM, mLength);
_match_stored:
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   U32 const indexToInsert = curr + 2;
                hashLong[hashPtr(base + indexToInsert, hBitsL, 8)] = indexToInsert;
                hashLong[hashPtr(ip - 2, hBitsL, 8)] = (U32)(ip - 2 - base);
                hashSmall[hashPtr(base + indexToInsert, hBitsS, mls)] = indexToInsert;
                hashSmall[hashPtr(ip - 1, hBitsS, mls)] = (U32)(ip - 1 - base);
                ORC_TRACE('C', indexToInsert, (U32)hashPtr(base + indexToInsert, hBitsL, 8), (U32)hashPtr(base + indexToInsert, hBitsS, mls));
                ORC_TRACE('D', (U32)(ip - base), (U32)hashPtr(ip - 2, hBitsL, 8), (U32)hashPtr(ip - 1, hBitsS, mls));
            }
            while ((ip <= ilimit) && ((offset_2 > 0) & (rd32(ip) == rd32(ip - offset_2)))) {
                size_t const rLength = count(ip + 4, ip + 4 - offset_2, iend) + 4;
                U32 const tmpOff = offset_2; offset_2 = offset_1; offset_1 = tmpOff;
                hashSmall[hashPtr(ip, hBitsS, mls)] = (U32)(ip - base);
                hashLong[hashPtr(ip, hBitsL, 8)] = (U32)(ip - base);
                ORC_STAT_IMMREP();
                ORC_TRACE('I', (U32)(ip - base), (U32)hashPtr(ip, hBitsL, 8), (U32)hashPtr(ip, hBitsS, mls));
                storeSeq(ss, 0, anchor, 1, rLength);
                ip += rLength; anchor = ip;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * bit stream (LSB-first, closed with a 1 bit)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { BYTE* start; BYTE* p; U64 acc; U32 n; } bitw_t;
static void bw_init(bitw_t* b, BYTE* dst) { b->start = b->p = dst; b->acc = 0; b->n = 0; }
static void bw_add(bitw_t* b, U64 v, U32 nb) {
    if (nb == 0) return;
    b->acc |= (v & ((nb >= 64) ? ~0ull : ((1ull << nb) - 1))) << b->n;
    b->n += nb;
    while (b->n >= 8) { *b->p++ = (BYTE)b->acc; b->acc >>= 8; b->n -= 8; }
}
static size_t bw_close(bitw_t* b) {
    bw_add(b, 1, 1);
    if (b->n) { *b->p++ = (BYTE)b->acc; b->acc = 0; b->n = 0; }
    return (size_t)(b->p - b->start);
}

/* ------------------------------------------------------------------------------------------------
 * FSE
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int deltaFindState; U32 deltaNbBits; } fse_tt;
typedef struct { U32 tableLog; U16 stateTable[1 << FSE_MAX_TABLELOG]; fse_tt tt[256]; } fse_ctable;

static U32 FSE_minTableLog(size_t srcSize, U32 maxSymbolValue) {
    U32 minBitsSrc = highbit32((U32)srcSize) + 1;
    U32 minBitsSymbols = highbit32(maxSymbolValue) + 2;
    return minBitsSrc < minBitsSymbols ? minBitsSrc : minBitsSymbols;
}
static U32 FSE_optimalTableLog_internal(U32 maxTableLog, size_t srcSize, U32 maxSymbolValue, U32 minus) {
    U32 maxBitsSrc = highbit32((U32)(srcSize - 1)) - minus;
    U32 tableLog = maxTableLog;
    U32 minBits = FSE_minTableLog(srcSize, maxSymbolValue);
    if (tableLog == 0) tableLog = FSE_DEFAULT_TABLELOG;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < FSE_MIN_TABLELOG) tableLog = FSE_MIN_TABLELOG;
    if (tableLog > FSE_MAX_TABLELOG) tableLog = FSE_MAX_TABLELOG;
    return tableLog;
}

static int FSE_normalizeM2(short* norm, U32 tableLog, const unsigned* cnt, size_t total, U32 maxSymbolValue, short lowProbCount) {
    short const NOT_YET_ASSIGNED = -2;
    U32 s, distributed = 0, ToDistribute;
    U32 const lowThreshold = (U32)(total >> tableLog);
    U32 lowOne = (U32)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSymbolValue; s++) {
        if (cnt[s] == 0) { norm[s] = 0; continue; }
        if (cnt[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= cnt[s]; continue; }
        if (cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; continue; }
        norm[s] = NOT_YET_ASSIGNED;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (U32)((total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSymbolValue; s++)
            if ((norm[s] == NOT_YET_ASSIGNED) && (cnt[s] <= lowOne)) { norm[s] = 1; distributed++; total -= cnt[s]; }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSymbolValue + 1) {
        U32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSymbolValue; s++) if (cnt[s] > maxC) { maxV = s; maxC = cnt[s]; }
        norm[maxV] += (short)ToDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSymbolValue + 1))
            if (norm[s] > 0) { ToDistribute--; norm[s]++; }
        return 0;
    }
    {   U64 const vStepLog = 62 - tableLog;
        U64 const mid = (1ULL << (vStepLog - 1)) - 1;
        U64 const rStep = ((((U64)1 << vStepLog) * ToDistribute) + mid) / (U32)total;
        U64 tmpTotal = mid;
        for (s = 0; s <= maxSymbolValue; s++) {
            if (norm[s] == NOT_YET_ASSIGNED) {
                U64 const end = tmpTotal + (cnt[s] * rStep);
                U32 const sStart = (U32)(tmpTotal >> vStepLog), sEnd = (U32)(end >> vStepLog);
                U32 const weight = sEnd - sStart;
                if (weight < 1) return -1;
                norm[s] = (short)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

/* returns tableLog, 0 for the rle special case, <0 on error */
static int FSE_normalizeCount(short* norm, U32 tableLog, const unsigned* cnt, size_t total, U32 maxSymbolValue, U32 useLowProbCount) {
    static U32 const rtbTable[] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};
    short const lowProbCount = useLowProbCount ? -1 : 1;
    U64 const scale = 62 - tableLog;
    U64 const step = ((U64)1 << 62) / (U32)total;
    U64 const vStep = 1ULL << (scale - 20);
    int stillToDistribute = 1 << tableLog;
    unsigned s, largest = 0; short largestP = 0;
    U32 lowThreshold = (U32)(total >> tableLog);
    if (tableLog < FSE_minTableLog(total, maxSymbolValue)) return -1;
    for (s = 0; s <= maxSymbolValue; s++) {
        if (cnt[s] == total) return 0;
        if (cnt[s] == 0) { norm[s] = 0; continue; }
        if (cnt[s] <= lowThreshold) { norm[s] = lowProbCount; stillToDistribute--; }
        else {
            short proba = (short)((cnt[s] * step) >> scale);
            if (proba < 8) {
                U64 restToBeat = vStep * rtbTable[proba];
                proba += (cnt[s] * step) - ((U64)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba;
            stillToDistribute -= proba;
        }
    }
    if (-stillToDistribute >= (norm[largest] >> 1)) {
        if (FSE_normalizeM2(norm, tableLog, cnt, total, maxSymbolValue, lowProbCount) < 0) return -1;
    } else norm[largest] += (short)stillToDistribute;
    return (int)tableLog;
}

static size_t FSE_writeNCount(BYTE* out0, const short* norm, unsigned maxSymbolValue, unsigned tableLog) {
    BYTE* out = out0;
    int nbBits; const int tableSize = 1 << tableLog; int remaining, threshold;
    U32 bitStream = 0; int bitCount = 0; unsigned symbol = 0; unsigned const alphabetSize = maxSymbolValue + 1; int previousIs0 = 0;
    bitStream += (tableLog - FSE_MIN_TABLELOG) << bitCount; bitCount += 4;
    remaining = tableSize + 1; threshold = tableSize; nbBits = (int)tableLog + 1;
    while ((symbol < alphabetSize) && (remaining > 1)) {
        if (previousIs0) {
            unsigned start = symbol;
            while ((symbol < alphabetSize) && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24;
                bitStream += 0xFFFFU << bitCount;
                out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2;
                bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int cnt = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= cnt < 0 ? -cnt : cnt;
            cnt++;
            if (cnt >= threshold) cnt += max;
            bitStream += (U32)cnt << bitCount;
            bitCount += nbBits;
            bitCount -= (cnt < max);
            previousIs0 = (cnt == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (BYTE)bitStream; out[1] = (BYTE)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (size_t)(out - out0);
}

static void FSE_buildCTable(fse_ctable* ct, const short* norm, unsigned maxSymbolValue, unsigned tableLog) {
    U32 const tableSize = 1u << tableLog, tableMask = tableSize - 1;
    U32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    U32 cumul[258]; BYTE tableSymbol[1 << FSE_MAX_TABLELOG];
    U32 highThreshold = tableSize - 1; U32 u;
    ct->tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSymbolValue + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[highThreshold--] = (BYTE)(u - 1); }
        else cumul[u] = cumul[u - 1] + (U32)norm[u - 1];
    }
    cumul[maxSymbolValue + 1] = tableSize + 1;
    {   U32 position = 0, symbol;
        for (symbol = 0; symbol <= maxSymbolValue; symbol++) {
            int n; int const freq = norm[symbol];
            for (n = 0; n < freq; n++) {
                tableSymbol[position] = (BYTE)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u = 0; u < tableSize; u++) { BYTE s = tableSymbol[u]; ct->stateTable[cumul[s]++] = (U16)(tableSize + u); }
    {   unsigned total = 0, s;
        for (s = 0; s <= maxSymbolValue; s++) {
            switch (norm[s]) {
                case 0: ct->tt[s].deltaNbBits = ((tableLog + 1) << 16) - (1 << tableLog); ct->tt[s].deltaFindState = 0; break;
                case -1: case 1:
                    ct->tt[s].deltaNbBits = (tableLog << 16) - (1 << tableLog);
                    ct->tt[s].deltaFindState = (int)(total - 1); total++; break;
                default: {
                    U32 const maxBitsOut = tableLog - highbit32((U32)norm[s] - 1);
                    U32 const minStatePlus = (U32)norm[s] << maxBitsOut;
                    ct->tt[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                    ct->tt[s].deltaFindState = (int)(total - (unsigned)norm[s]);
                    total += (unsigned)norm[s];
                }
            }
        }
    }
}
static void FSE_buildCTable_rle(fse_ctable* ct, BYTE symbol) {
    ct->tableLog = 0; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->tt[symbol].deltaNbBits = 0; ct->tt[symbol].deltaFindState = 0;
}
typedef struct { ptrdiff_t value; const fse_ctable* ct; } fse_cstate;
static void FSE_initCState2(fse_cstate* st, const fse_ctable* ct, U32 symbol) {
    fse_tt const tt = ct->tt[symbol];
    U32 nbBitsOut = (U32)((tt.deltaNbBits + (1 << 15)) >> 16);
    st->ct = ct;
    st->value = (ptrdiff_t)((nbBitsOut << 16) - tt.deltaNbBits);
    st->value = ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static void FSE_encodeSymbol(bitw_t* b, fse_cstate* st, U32 symbol) {
    fse_tt const tt = st->ct->tt[symbol];
    U32 const nbBitsOut = (U32)((st->value + tt.deltaNbBits) >> 16);
    bw_add(b, (U64)st->value, nbBitsOut);
    st->value = st->ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static void FSE_flushCState(bitw_t* b, const fse_cstate* st) { bw_add(b, (U64)st->value, st->ct->tableLog); }

/* ------------------------------------------------------------------------------------------------
 * Huffman
 * ---------------------------------------------------------------------------------------------- */
typedef struct { U32 count; U16 parent; BYTE byte; BYTE nbBits; } nodeElt;
typedef struct { U16 val[256]; BYTE nbBits[256]; U32 tableLog; U32 maxSymbolValue; } huf_ctable;
typedef struct { U16 base, curr; } rankPos;
#define RANK_POSITION_TABLE_SIZE 192
#define RANK_POSITION_LOG_BUCKETS_BEGIN 158
#define RANK_POSITION_DISTINCT_COUNT_CUTOFF (RANK_POSITION_LOG_BUCKETS_BEGIN + 7)  /* + highbit32(158) == 165 */
static U32 HUF_getIndex(U32 cnt) { return (cnt < RANK_POSITION_DISTINCT_COUNT_CUTOFF) ? cnt : highbit32(cnt) + RANK_POSITION_LOG_BUCKETS_BEGIN; }
static void HUF_swapNodes(nodeElt* a, nodeElt* b) { nodeElt t = *a; *a = *b; *b = t; }
static void HUF_insertionSort(nodeElt h[], int const low, int const high) {
    int i; int const size = high - low + 1;
    h += low;
    for (i = 1; i < size; ++i) {
        nodeElt const key = h[i]; int j = i - 1;
        while (j >= 0 && h[j].count < key.count) { h[j + 1] = h[j]; j--; }
        h[j + 1] = key;
    }
}
static int HUF_quickSortPartition(nodeElt arr[], int const low, int const high) {
    U32 const pivot = arr[high].count; int i = low - 1; int j = low;
    for (; j < high; j++) if (arr[j].count > pivot) { i++; HUF_swapNodes(&arr[i], &arr[j]); }
    HUF_swapNodes(&arr[i + 1], &arr[high]);
    return i + 1;
}
static void HUF_simpleQuickSort(nodeElt arr[], int low, int high) {
    int const kInsertionSortThreshold = 8;
    if (high - low < kInsertionSortThreshold) { HUF_insertionSort(arr, low, high); return; }
    while (low < high) {
        int const idx = HUF_quickSortPartition(arr, low, high);
        if (idx - low < high - idx) { HUF_simpleQuickSort(arr, low, idx - 1); low = idx + 1; }
        else { HUF_simpleQuickSort(arr, idx + 1, high); high = idx - 1; }
    }
}
static void HUF_sort(nodeElt huffNode[], const unsigned cnt[], U32 const maxSymbolValue, rankPos rankPosition[]) {
    U32 n; U32 const maxSymbolValue1 = maxSymbolValue + 1;
    memset(rankPosition, 0, sizeof(*rankPosition) * RANK_POSITION_TABLE_SIZE);
    for (n = 0; n < maxSymbolValue1; ++n) rankPosition[HUF_getIndex(cnt[n])].base++;
    for (n = RANK_POSITION_TABLE_SIZE - 1; n > 0; --n) {
        rankPosition[n - 1].base += rankPosition[n].base;
        rankPosition[n - 1].curr = rankPosition[n - 1].base;
    }
    for (n = 0; n < maxSymbolValue1; ++n) {
        U32 const c = cnt[n]; U32 const r = HUF_getIndex(c) + 1; U32 const pos = rankPosition[r].curr++;
        huffNode[pos].count = c; huffNode[pos].byte = (BYTE)n;
    }
    for (n = RANK_POSITION_DISTINCT_COUNT_CUTOFF; n < RANK_POSITION_TABLE_SIZE - 1; ++n) {
        int const bucketSize = rankPosition[n].curr - rankPosition[n].base;
        U32 const bucketStartIdx = rankPosition[n].base;
        if (bucketSize > 1) HUF_simpleQuickSort(huffNode + bucketStartIdx, 0, bucketSize - 1);
    }
}
#define STARTNODE (HUF_SYMBOLVALUE_MAX + 1)
static int HUF_buildTree(nodeElt* huffNode, U32 maxSymbolValue) {
    nodeElt* const huffNode0 = huffNode - 1;
    int nonNullRank; int lowS, lowN; int nodeNb = STARTNODE; int n, nodeRoot;
    nonNullRank = (int)maxSymbolValue;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (U16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = (U32)(1U << 30);
    huffNode0[0].count = (U32)(1U << 31);
    while (nodeNb <= nodeRoot) {
        int const n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        int const n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (U16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= STARTNODE; n--) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    return nonNullRank;
}
static U32 HUF_setMaxHeight(nodeElt* huffNode, U32 lastNonNull, U32 targetNbBits) {
    const U32 largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= targetNbBits) return largestBits;
    {   int totalCost = 0; const U32 baseCost = 1 << (largestBits - targetNbBits); int n = (int)lastNonNull;
        while (huffNode[n].nbBits > targetNbBits) {
            totalCost += baseCost - (1 << (largestBits - huffNode[n].nbBits));
            huffNode[n].nbBits = (BYTE)targetNbBits; n--;
        }
        while (huffNode[n].nbBits == targetNbBits) --n;
        totalCost >>= (largestBits - targetNbBits);
        {   U32 const noSymbol = 0xF0F0F0F0; U32 rankLast[HUF_TABLELOG_MAX + 2];
            memset(rankLast, 0xF0, sizeof(rankLast));
            {   U32 currentNbBits = targetNbBits; int pos;
                for (pos = n; pos >= 0; pos--) {
                    if (huffNode[pos].nbBits >= currentNbBits) continue;
                    currentNbBits = huffNode[pos].nbBits;
                    rankLast[targetNbBits - currentNbBits] = (U32)pos;
                }
            }
            while (totalCost > 0) {
                U32 nBitsToDecrease = highbit32((U32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    U32 const highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    {   U32 const highTotal = huffNode[highPos].count, lowTotal = 2 * huffNode[lowPos].count;
                        if (highTotal <= lowTotal) break; }
                }
                while ((nBitsToDecrease <= HUF_TABLELOG_MAX) && (rankLast[nBitsToDecrease] == noSymbol)) nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                huffNode[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (huffNode[rankLast[nBitsToDecrease]].nbBits != targetNbBits - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (huffNode[n].nbBits == targetNbBits) n--;
                    huffNode[n + 1].nbBits--;
                    rankLast[1] = (U32)(n + 1);
                    totalCost++;
                    continue;
                }
                huffNode[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return targetNbBits;
}
static U32 HUF_buildCTable(huf_ctable* ct, const unsigned* cnt, U32 maxSymbolValue, U32 maxNbBits) {
    nodeElt table[2 * (HUF_SYMBOLVALUE_MAX + 1) + 1]; nodeElt* const huffNode = table + 1;
    rankPos rankPosition[RANK_POSITION_TABLE_SIZE];
    int nonNullRank, n;
    memset(table, 0, sizeof table);
    HUF_sort(huffNode, cnt, maxSymbolValue, rankPosition);
    nonNullRank = HUF_buildTree(huffNode, maxSymbolValue);
    maxNbBits = HUF_setMaxHeight(huffNode, (U32)nonNullRank, maxNbBits);
    {   U16 nbPerRank[HUF_TABLELOG_MAX + 1] = {0}, valPerRank[HUF_TABLELOG_MAX + 1] = {0};
        int const alphabetSize = (int)(maxSymbolValue + 1);
        memset(ct, 0, sizeof *ct);
        for (n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
        {   U16 min = 0;
            for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; } }
        for (n = 0; n < alphabetSize; n++) ct->nbBits[huffNode[n].byte] = huffNode[n].nbBits;
        for (n = 0; n < alphabetSize; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
        ct->tableLog = maxNbBits; ct->maxSymbolValue = maxSymbolValue;
    }
    return maxNbBits;
}

static size_t FSE_compress_usingCTable(BYTE* dst, const BYTE* src, size_t srcSize, const fse_ctable* ct) {
    const BYTE* const istart = src; const BYTE* ip = istart + srcSize;
    bitw_t b; fse_cstate s1, s2;
    if (srcSize <= 2) return 0;
    bw_init(&b, dst);
    if (srcSize & 1) { FSE_initCState2(&s1, ct, *--ip); FSE_initCState2(&s2, ct, *--ip); FSE_encodeSymbol(&b, &s1, *--ip); }
    else { FSE_initCState2(&s2, ct, *--ip); FSE_initCState2(&s1, ct, *--ip); }
    while (ip > istart) { FSE_encodeSymbol(&b, &s2, *--ip); if (ip > istart) FSE_encodeSymbol(&b, &s1, *--ip); }
    FSE_flushCState(&b, &s2); FSE_flushCState(&b, &s1);
    return bw_close(&b);
}

static size_t HUF_compressWeights(BYTE* dst, const BYTE* weightTable, size_t wtSize) {
    BYTE* op = dst; unsigned maxSymbolValue = HUF_TABLELOG_MAX; U32 tableLog = 6;
    unsigned cnt[HUF_TABLELOG_MAX + 1] = {0}; short norm[HUF_TABLELOG_MAX + 1]; fse_ctable ct;
    if (wtSize <= 1) return 0;
    {   unsigned maxCount = 0; size_t i;
        for (i = 0; i < wtSize; i++) cnt[weightTable[i]]++;
        while (!cnt[maxSymbolValue]) maxSymbolValue--;
        for (i = 0; i <= maxSymbolValue; i++) if (cnt[i] > maxCount) maxCount = cnt[i];
        if (maxCount == wtSize) return 1;
        if (maxCount == 1) return 0;
    }
    tableLog = FSE_optimalTableLog_internal(tableLog, wtSize, maxSymbolValue, 2);
    if (FSE_normalizeCount(norm, tableLog, cnt, wtSize, maxSymbolValue, 0) < 0) return (size_t)-1;
    {   size_t const hSize = FSE_writeNCount(op, norm, maxSymbolValue, tableLog); op += hSize; }
    FSE_buildCTable(&ct, norm, maxSymbolValue, tableLog);
    {   size_t const cSize = FSE_compress_usingCTable(op, weightTable, wtSize, &ct);
        if (cSize == 0) return 0;
        op += cSize; }
    return (size_t)(op - dst);
}

static size_t HUF_writeCTable(BYTE* dst, const huf_ctable* ct, unsigned maxSymbolValue, unsigned huffLog) {
    BYTE bitsToWeight[HUF_TABLELOG_MAX + 1]; BYTE huffWeight[HUF_SYMBOLVALUE_MAX + 1]; BYTE* op = dst; U32 n;
    bitsToWeight[0] = 0;
    for (n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (BYTE)(huffLog + 1 - n);
    for (n = 0; n < maxSymbolValue; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    {   size_t const hSize = HUF_compressWeights(op + 1, huffWeight, maxSymbolValue);
        if (hSize == (size_t)-1) return (size_t)-1;
        if ((hSize > 1) & (hSize < maxSymbolValue / 2)) { op[0] = (BYTE)hSize; return hSize + 1; } }
    if (maxSymbolValue > (256 - 128)) return (size_t)-1;
    op[0] = (BYTE)(128 + (maxSymbolValue - 1));
    huffWeight[maxSymbolValue] = 0;
    for (n = 0; n < maxSymbolValue; n += 2) op[(n / 2) + 1] = (BYTE)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSymbolValue + 1) / 2) + 1;
}

static size_t HUF_compress1X_usingCTable(BYTE* dst, const BYTE* src, size_t srcSize, const huf_ctable* ct) {
    bitw_t b; size_t n;
    bw_init(&b, dst);
    for (n = srcSize; n > 0; n--) bw_add(&b, ct->val[src[n - 1]], ct->nbBits[src[n - 1]]);
    return bw_close(&b);
}
static size_t HUF_compress4X_usingCTable(BYTE* dst, const BYTE* src, size_t srcSize, const huf_ctable* ct) {
    size_t const segmentSize = (srcSize + 3) / 4; const BYTE* ip = src; const BYTE* const iend = src + srcSize;
    BYTE* const ostart = dst; BYTE* op = ostart;
    if (srcSize < 12) return 0;
    op += 6;
    for (int i = 0; i < 3; i++) {
        size_t const cSize = HUF_compress1X_usingCTable(op, ip, segmentSize, ct);
        if (cSize == 0 || cSize > 65535) return 0;
        wr16(ostart + 2 * i, (U32)cSize);
        op += cSize; ip += segmentSize;
    }
    {   size_t const cSize = HUF_compress1X_usingCTable(op, ip, (size_t)(iend - ip), ct);
        if (cSize == 0 || cSize > 65535) return 0;
        op += cSize; }
    return (size_t)(op - ostart);
}
static size_t HUF_compressCTable_internal(BYTE* ostart, BYTE* op, const BYTE* src, size_t srcSize, int singleStream, const huf_ctable* ct) {
    size_t const cSize = singleStream ? HUF_compress1X_usingCTable(op, src, srcSize, ct) : HUF_compress4X_usingCTable(op, src, srcSize, ct);
    if (cSize == 0) return 0;
    op += cSize;
    if ((size_t)(op - ostart) >= srcSize - 1) return 0;
    return (size_t)(op - ostart);
}
static size_t HUF_estimateCompressedSize(const huf_ctable* ct, const unsigned* cnt, unsigned maxSymbolValue) {
    size_t nbBits = 0;
    for (unsigned s = 0; s <= maxSymbolValue; ++s) nbBits += (size_t)ct->nbBits[s] * cnt[s];
    return nbBits >> 3;
}
static int HUF_validateCTable(const huf_ctable* ct, const unsigned* cnt, unsigned maxSymbolValue) {
    int bad = 0;
    if (ct->maxSymbolValue < maxSymbolValue) return 0;
    for (unsigned s = 0; s <= maxSymbolValue; ++s) bad |= (cnt[s] != 0) & (ct->nbBits[s] == 0);
    return !bad;
}

enum { HUF_repeat_none = 0, HUF_repeat_check = 1, HUF_repeat_valid = 2 };
#define HUF_flags_preferRepeat 4
#define HUF_flags_suspectUncompressible 8

/* HUF_compress_internal: returns compressed size, 0 = not compressible, 1 = single symbol (rle) */
static size_t HUF_compress_repeat(BYTE* dst, const BYTE* src, size_t srcSize, int singleStream, huf_ctable* oldHufTable, int* repeat, int flags) {
    BYTE* const ostart = dst; BYTE* op = ostart;
    unsigned cnt[256]; unsigned maxSymbolValue = HUF_SYMBOLVALUE_MAX; U32 huffLog = LitHufLog;
    huf_ctable table;
    if (!srcSize) return 0;
    if ((flags & HUF_flags_preferRepeat) && *repeat == HUF_repeat_valid)
        return HUF_compressCTable_internal(ostart, op, src, srcSize, singleStream, oldHufTable);
    if ((flags & HUF_flags_suspectUncompressible) && srcSize >= (4096 * 10)) {
        unsigned c2[256]; size_t largestTotal = 0, i; unsigned m;
        memset(c2, 0, sizeof c2); for (i = 0; i < 4096; i++) c2[src[i]]++; m = 0; for (i = 0; i < 256; i++) if (c2[i] > m) m = c2[i]; largestTotal += m;
        memset(c2, 0, sizeof c2); for (i = 0; i < 4096; i++) c2[src[srcSize - 4096 + i]]++; m = 0; for (i = 0; i < 256; i++) if (c2[i] > m) m = c2[i]; largestTotal += m;
        if (largestTotal <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {   size_t largest = 0, i;
        memset(cnt, 0, sizeof cnt);
        for (i = 0; i < srcSize; i++) cnt[src[i]]++;
        while (!cnt[maxSymbolValue]) maxSymbolValue--;
        for (i = 0; i <= maxSymbolValue; i++) if (cnt[i] > largest) largest = cnt[i];
        if (largest == srcSize) { *ostart = src[0]; return 1; }
        if (largest <= (srcSize >> 7) + 4) return 0;
    }
    if (*repeat == HUF_repeat_check && !HUF_validateCTable(oldHufTable, cnt, maxSymbolValue)) *repeat = HUF_repeat_none;
    if ((flags & HUF_flags_preferRepeat) && *repeat != HUF_repeat_none)
        return HUF_compressCTable_internal(ostart, op, src, srcSize, singleStream, oldHufTable);
    huffLog = FSE_optimalTableLog_internal(huffLog, srcSize, maxSymbolValue, 1);
    huffLog = HUF_buildCTable(&table, cnt, maxSymbolValue, huffLog);
    {   size_t const hSize = HUF_writeCTable(op, &table, maxSymbolValue, huffLog);
        if (hSize == (size_t)-1) return 0;   /* cannot describe the table: treat as not compressible */
        if (*repeat != HUF_repeat_none) {
            size_t const oldSize = HUF_estimateCompressedSize(oldHufTable, cnt, maxSymbolValue);
            size_t const newSize = HUF_estimateCompressedSize(&table, cnt, maxSymbolValue);
            if (oldSize <= hSize + newSize || hSize + 12 >= srcSize)
                return HUF_compressCTable_internal(ostart, op, src, srcSize, singleStream, oldHufTable);
        }
        if (hSize + 12ul >= srcSize) return 0;
        op += hSize;
        *repeat = HUF_repeat_none;
        *oldHufTable = table;
    }
    return HUF_compressCTable_internal(ostart, op, src, srcSize, singleStream, &table);
}

/* ------------------------------------------------------------------------------------------------
 * entropy state carried from block to block
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    huf_ctable huf; int huf_repeat;
    fse_ctable ll, of, ml; int ll_repeat, of_repeat, ml_repeat;      /* FSE_repeat_{none,check,valid} */
    U32 rep[3];
} blockstate_t;
enum { set_basic = 0, set_rle = 1, set_compressed = 2, set_repeat = 3 };
enum { FSE_repeat_none = 0, FSE_repeat_check = 1, FSE_repeat_valid = 2 };

static size_t noCompressLiterals(BYTE* dst, const BYTE* src, size_t srcSize) {
    U32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    switch (flSize) {
        case 1: dst[0] = (BYTE)((U32)set_basic + (srcSize << 3)); break;
        case 2: wr16(dst, (U32)set_basic + (1 << 2) + (U32)(srcSize << 4)); break;
        default: wr32(dst, (U32)set_basic + (3 << 2) + (U32)(srcSize << 4)); break;
    }
    memcpy(dst + flSize, src, srcSize);
    return srcSize + flSize;
}
static size_t compressRleLiteralsBlock(BYTE* dst, const BYTE* src, size_t srcSize) {
    U32 const flSize = 1 + (srcSize > 31) + (srcSize > 4095);
    switch (flSize) {
        case 1: dst[0] = (BYTE)((U32)set_rle + (srcSize << 3)); break;
        case 2: wr16(dst, (U32)set_rle + (1 << 2) + (U32)(srcSize << 4)); break;
        default: wr32(dst, (U32)set_rle + (3 << 2) + (U32)(srcSize << 4)); break;
    }
    dst[flSize] = *src;
    return flSize + 1;
}
static size_t minGain(size_t srcSize, U32 strat) { U32 const minlog = (strat >= 8) ? strat - 1 : 6; return (srcSize >> minlog) + 2; }
static int allBytesIdentical(const BYTE* src, size_t n) { for (size_t i = 1; i < n; i++) if (src[i] != src[0]) return 0; return 1; }

static size_t compressLiterals(BYTE* dst, const BYTE* src, size_t srcSize, const blockstate_t* prev, blockstate_t* next, U32 strategy, int suspectUncompressible) {
    size_t const lhSize = 3 + (srcSize >= 1 KB) + (srcSize >= 16 KB);
    BYTE* const ostart = dst; U32 singleStream = srcSize < 256; int hType = set_compressed; size_t cLitSize;
    next->huf = prev->huf; next->huf_repeat = prev->huf_repeat;
    {   int const shift = (9 - (int)strategy) < 3 ? (9 - (int)strategy) : 3;
        size_t const mintc = (prev->huf_repeat == HUF_repeat_valid) ? 6 : (size_t)8 << shift;
        if (srcSize < mintc) return noCompressLiterals(dst, src, srcSize); }
    {   int repeat = prev->huf_repeat;
        int const flags = ((strategy < 4 && srcSize <= 1024) ? HUF_flags_preferRepeat : 0) | (suspectUncompressible ? HUF_flags_suspectUncompressible : 0);
        if (repeat == HUF_repeat_valid && lhSize == 3) singleStream = 1;
        cLitSize = HUF_compress_repeat(ostart + lhSize, src, srcSize, (int)singleStream, &next->huf, &repeat, flags);
        if (repeat != HUF_repeat_none) hType = set_repeat;
    }
    {   size_t const mg = minGain(srcSize, strategy);
        if ((cLitSize == 0) || (cLitSize >= srcSize - mg)) { next->huf = prev->huf; next->huf_repeat = prev->huf_repeat; return noCompressLiterals(dst, src, srcSize); } }
    if (cLitSize == 1) {
        if ((srcSize >= 8) || allBytesIdentical(src, srcSize)) { next->huf = prev->huf; next->huf_repeat = prev->huf_repeat; return compressRleLiteralsBlock(dst, src, srcSize); }
    }
    if (hType == set_compressed) next->huf_repeat = HUF_repeat_check;
    switch (lhSize) {
        case 3: { U32 const lhc = hType + ((U32)(!singleStream) << 2) + ((U32)srcSize << 4) + ((U32)cLitSize << 14); wr24(ostart, lhc); break; }
        case 4: { U32 const lhc = hType + (2 << 2) + ((U32)srcSize << 4) + ((U32)cLitSize << 18); wr32(ostart, lhc); break; }
        default: { U32 const lhc = hType + (3 << 2) + ((U32)srcSize << 4) + ((U32)cLitSize << 22); wr32(ostart, lhc); ostart[4] = (BYTE)(cLitSize >> 10); break; }
    }
    return lhSize + cLitSize;
}

static int selectEncodingType(int* repeatMode, size_t mostFrequent, size_t nbSeq, U32 defaultNormLog, int isDefaultAllowed, U32 strategy) {
    if (mostFrequent == nbSeq) {
        *repeatMode = FSE_repeat_none;
        if (isDefaultAllowed && nbSeq <= 2) return set_basic;
        return set_rle;
    }
    /* strategy < ZSTD_lazy: simple heuristics */
    if (isDefaultAllowed) {
        size_t const staticFse_nbSeq_max = 1000;
        size_t const mult = 10 - strategy;
        size_t const dynamicFse_nbSeq_min = (((size_t)1 << defaultNormLog) * mult) >> 3;
        if ((*repeatMode == FSE_repeat_valid) && (nbSeq < staticFse_nbSeq_max)) return set_repeat;
        if ((nbSeq < dynamicFse_nbSeq_min) || (mostFrequent < (nbSeq >> (defaultNormLog - 1)))) { *repeatMode = FSE_repeat_none; return set_basic; }
    }
    *repeatMode = FSE_repeat_check;
    return set_compressed;
}

/* returns bytes written for the table description, or (size_t)-1 */
static size_t buildCTable(BYTE* op, fse_ctable* next, U32 FSELog, int type, unsigned* cnt, U32 max, const BYTE* codeTable, size_t nbSeq,
                          const short* defaultNorm, U32 defaultNormLog, U32 defaultMax, const fse_ctable* prev) {
    switch (type) {
        case set_rle: FSE_buildCTable_rle(next, (BYTE)max); *op = codeTable[0]; return 1;
        case set_repeat: *next = *prev; return 0;
        case set_basic: FSE_buildCTable(next, defaultNorm, defaultMax, defaultNormLog); return 0;
        default: {
            short norm[MaxML + 1]; size_t nbSeq_1 = nbSeq;
            const U32 tableLog = FSE_optimalTableLog_internal(FSELog, nbSeq, max, 2);
            if (cnt[codeTable[nbSeq - 1]] > 1) { cnt[codeTable[nbSeq - 1]]--; nbSeq_1--; }
            if (FSE_normalizeCount(norm, tableLog, cnt, nbSeq_1, max, nbSeq_1 >= 2048) < 0) return (size_t)-1;
            {   size_t const NCountSize = FSE_writeNCount(op, norm, max, tableLog);
                FSE_buildCTable(next, norm, max, tableLog);
                return NCountSize; }
        }
    }
}

static size_t histCount(unsigned* cnt, unsigned* maxSymbolValuePtr, const BYTE* src, size_t n) {
    unsigned maxSymbolValue = *maxSymbolValuePtr; size_t largest = 0;
    memset(cnt, 0, (maxSymbolValue + 1) * sizeof *cnt);
    for (size_t i = 0; i < n; i++) cnt[src[i]]++;
    while (!cnt[maxSymbolValue]) maxSymbolValue--;
    *maxSymbolValuePtr = maxSymbolValue;
    for (unsigned s = 0; s <= maxSymbolValue; s++) if (cnt[s] > largest) largest = cnt[s];
    return largest;
}

/* ZSTD_entropyCompressSeqStore_internal; returns 0 when the caller must emit a raw block */
static size_t entropyCompressSeqStore(seqstore_t* ss, const blockstate_t* prev, blockstate_t* next, U32 strategy, BYTE* dst, size_t srcSize) {
    BYTE* const ostart = dst; BYTE* op = ostart;
    size_t const nbSeq = ss->nbSeq; unsigned cnt[MaxML + 2]; size_t lastCountSize = 0;
    {   unsigned const suspectUncompressible = (nbSeq == 0) || (ss->litSize / nbSeq >= 20);
        op += compressLiterals(op, ss->lit, ss->litSize, prev, next, strategy, (int)suspectUncompressible); }
    if (nbSeq < 128) *op++ = (BYTE)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (BYTE)((nbSeq >> 8) + 0x80); op[1] = (BYTE)nbSeq; op += 2; }
    else { op[0] = 0xFF; wr16(op + 1, (U32)(nbSeq - 0x7F00)); op += 3; }
    if (nbSeq == 0) {
        next->ll = prev->ll; next->of = prev->of; next->ml = prev->ml;
        next->ll_repeat = prev->ll_repeat; next->of_repeat = prev->of_repeat; next->ml_repeat = prev->ml_repeat;
        goto check;
    }
    {   BYTE* const seqHead = op++; U32 LLtype, Offtype, MLtype;
        for (size_t u = 0; u < nbSeq; u++) {
            ss->llCode[u] = (BYTE)LLcode(ss->seqs[u].litLength);
            ss->ofCode[u] = (BYTE)highbit32(ss->seqs[u].offBase);
            ss->mlCode[u] = (BYTE)MLcode(ss->seqs[u].mlBase);
        }
        {   unsigned max = MaxLL; size_t const mostFrequent = histCount(cnt, &max, ss->llCode, nbSeq);
            next->ll_repeat = prev->ll_repeat;
            LLtype = (U32)selectEncodingType(&next->ll_repeat, mostFrequent, nbSeq, 6, 1, strategy);
            {   size_t const countSize = buildCTable(op, &next->ll, LLFSELog, (int)LLtype, cnt, max, ss->llCode, nbSeq, LL_defaultNorm, 6, MaxLL, &prev->ll);
                if (countSize == (size_t)-1) return 0;
                if (LLtype == set_compressed) lastCountSize = countSize;
                op += countSize; } }
        {   unsigned max = MaxOff; size_t const mostFrequent = histCount(cnt, &max, ss->ofCode, nbSeq);
            int const defaultPolicy = (max <= DefaultMaxOff);
            next->of_repeat = prev->of_repeat;
            Offtype = (U32)selectEncodingType(&next->of_repeat, mostFrequent, nbSeq, 5, defaultPolicy, strategy);
            {   size_t const countSize = buildCTable(op, &next->of, OffFSELog, (int)Offtype, cnt, max, ss->ofCode, nbSeq, OF_defaultNorm, 5, DefaultMaxOff, &prev->of);
                if (countSize == (size_t)-1) return 0;
                if (Offtype == set_compressed) lastCountSize = countSize;
                op += countSize; } }
        {   unsigned max = MaxML; size_t const mostFrequent = histCount(cnt, &max, ss->mlCode, nbSeq);
            next->ml_repeat = prev->ml_repeat;
            MLtype = (U32)selectEncodingType(&next->ml_repeat, mostFrequent, nbSeq, 6, 1, strategy);
            {   size_t const countSize = buildCTable(op, &next->ml, MLFSELog, (int)MLtype, cnt, max, ss->mlCode, nbSeq, ML_defaultNorm, 6, MaxML, &prev->ml);
                if (countSize == (size_t)-1) return 0;
                if (MLtype == set_compressed) lastCountSize = countSize;
                op += countSize; } }
        *seqHead = (BYTE)((LLtype << 6) + (Offtype << 4) + (MLtype << 2));
    }
    {   bitw_t b; fse_cstate stML, stOF, stLL; size_t n;
        bw_init(&b, op);
        FSE_initCState2(&stML, &next->ml, ss->mlCode[nbSeq - 1]);
        FSE_initCState2(&stOF, &next->of, ss->ofCode[nbSeq - 1]);
        FSE_initCState2(&stLL, &next->ll, ss->llCode[nbSeq - 1]);
        bw_add(&b, ss->seqs[nbSeq - 1].litLength, LL_bits[ss->llCode[nbSeq - 1]]);
        bw_add(&b, ss->seqs[nbSeq - 1].mlBase, ML_bits[ss->mlCode[nbSeq - 1]]);
        bw_add(&b, ss->seqs[nbSeq - 1].offBase, ss->ofCode[nbSeq - 1]);
        for (n = nbSeq - 2; n < nbSeq; n--) {
            BYTE const llCode = ss->llCode[n], ofCode = ss->ofCode[n], mlCode = ss->mlCode[n];
            FSE_encodeSymbol(&b, &stOF, ofCode);
            FSE_encodeSymbol(&b, &stML, mlCode);
            FSE_encodeSymbol(&b, &stLL, llCode);
            bw_add(&b, ss->seqs[n].litLength, LL_bits[llCode]);
            bw_add(&b, ss->seqs[n].mlBase, ML_bits[mlCode]);
            bw_add(&b, ss->seqs[n].offBase, ofCode);
        }
        FSE_flushCState(&b, &stML); FSE_flushCState(&b, &stOF); FSE_flushCState(&b, &stLL);
        {   size_t const bitstreamSize = bw_close(&b);
            op += bitstreamSize;
            if (lastCountSize && (lastCountSize + bitstreamSize) < 4) return 0; }
    }
check:
    {   size_t const cSize = (size_t)(op - ostart);
        size_t const maxCSize = srcSize - minGain(srcSize, strategy);
        if (cSize >= maxCSize) return 0;
        return cSize; }
}

/* ------------------------------------------------------------------------------------------------
 * 1.5.7 pre-block splitter (ZSTD_splitBlock_byChunks, level 0 for dfast: byte histogram, 1 sample / 43 B)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { unsigned events[1024]; size_t nbEvents; } Fingerprint;
static void recordFingerprint43(Fingerprint* fp, const BYTE* p, size_t srcSize) {
    size_t const limit = srcSize - 2 + 1; size_t n;
    memset(fp->events, 0, sizeof(unsigned) * 256);
    fp->nbEvents = 0;
    for (n = 0; n < limit; n += 43) fp->events[p[n]]++;
    fp->nbEvents += limit / 43;
}
static U64 abs64(S64 v) { return (U64)(v < 0 ? -v : v); }
static int compareFingerprints(const Fingerprint* ref, const Fingerprint* newfp, int penalty) {
    U64 p50 = (U64)ref->nbEvents * (U64)newfp->nbEvents; U64 deviation = 0; size_t n;
    for (n = 0; n < 256; n++)
        deviation += abs64((S64)ref->events[n] * (S64)newfp->nbEvents - (S64)newfp->events[n] * (S64)ref->nbEvents);
    {   U64 threshold = p50 * (U64)(14 + penalty) / 16;
        return deviation >= threshold; }
}
static size_t splitBlock_byChunks0(const BYTE* p, size_t blockSize) {
    Fingerprint past, cur; int penalty = 3; size_t pos;
    memset(&past, 0, sizeof past);
    recordFingerprint43(&past, p, 8 KB);
    for (pos = 8 KB; pos <= blockSize - 8 KB; pos += 8 KB) {
        recordFingerprint43(&cur, p + pos, 8 KB);
        if (compareFingerprints(&past, &cur, penalty)) return pos;
        for (size_t n = 0; n < 256; n++) past.events[n] += cur.events[n];
        past.nbEvents += cur.nbEvents;
        if (penalty > 0) penalty--;
    }
    return blockSize;
}
static size_t optimalBlockSize(const BYTE* src, size_t srcSize, size_t blockSizeMax, S64 savings, int profile) {
    if (profile == 0) return srcSize < blockSizeMax ? srcSize : blockSizeMax;          /* <= 1.5.6 */
    if (srcSize < 128 KB || blockSizeMax < 128 KB) return srcSize < blockSizeMax ? srcSize : blockSizeMax;
    if (savings < 3) return 128 KB;
    return splitBlock_byChunks0(src, blockSizeMax);                                      /* splitLevels[dfast] = 1 -> byChunks(level 0) */
}

/* ------------------------------------------------------------------------------------------------
 * frame
 * ---------------------------------------------------------------------------------------------- */
static int isRLE(const BYTE* src, size_t n) { for (size_t i = 1; i < n; i++) if (src[i] != src[0]) return 0; return 1; }

size_t orc_l3_compress_bound(size_t n) { return n + (n >> 8) + (n < (128 KB) ? (((128 KB) - n) >> 11) : 0); }

/* profile: 0 = libzstd 1.5.6 block loop, 1 = 1.5.7 (pre-block splitter).  Returns the frame size, 0 on failure. */
size_t orc_l3_compress(const BYTE* src, size_t srcSize, BYTE* dst, size_t dstCap, int profile) {
    cparams_t cp; BYTE* op = dst;
    if (dstCap < orc_l3_compress_bound(srcSize) + 18) return 0;
    orc_l3_cparams(srcSize, &cp);
    {   /* ZSTD_writeFrameHeader: content size known, no checksum, no dictID */
        U32 const windowSize = 1u << cp.windowLog;
        U32 const singleSegment = windowSize >= srcSize;
        U32 const fcsCode = (srcSize >= 256) + (srcSize >= 65536 + 256) + (srcSize >= 0xFFFFFFFFU);
        wr32(op, 0xFD2FB528u); op += 4;
        *op++ = (BYTE)((singleSegment << 5) + (fcsCode << 6));
        if (!singleSegment) *op++ = (BYTE)((cp.windowLog - 10) << 3);
        switch (fcsCode) {
            case 0: if (singleSegment) *op++ = (BYTE)srcSize; break;
            case 1: wr16(op, (U32)(srcSize - 256)); op += 2; break;
            case 2: wr32(op, (U32)srcSize); op += 4; break;
            default: wr32(op, (U32)srcSize); wr32(op + 4, (U32)(srcSize >> 32)); op += 8; break;
        }
    }
    {   matchstate_t ms; seqstore_t ss; blockstate_t bs[2]; int cur = 0;
        size_t const blockSizeMax = (1u << cp.windowLog) < ZSTD_BLOCKSIZE_MAX ? (1u << cp.windowLog) : ZSTD_BLOCKSIZE_MAX;
        const BYTE* ip = src; size_t remaining = srcSize; S64 savings = 0; int isFirstBlock = 1; U32 blockIndex = 0;
        ms.cp = cp; ms.base = src - 2; ms.dictLimit = 2;
        ms.hashLong = (U32*)calloc((size_t)1 << cp.hashLog, 4);
        ms.hashSmall = (U32*)calloc((size_t)1 << cp.chainLog, 4);
        ss.seqs = (seq_t*)malloc(sizeof(seq_t) * (ZSTD_BLOCKSIZE_MAX / 3 + 8));
        ss.lit = (BYTE*)malloc(ZSTD_BLOCKSIZE_MAX + 32);
        ss.llCode = (BYTE*)malloc(ZSTD_BLOCKSIZE_MAX / 3 + 8); ss.mlCode = (BYTE*)malloc(ZSTD_BLOCKSIZE_MAX / 3 + 8); ss.ofCode = (BYTE*)malloc(ZSTD_BLOCKSIZE_MAX / 3 + 8);
        memset(bs, 0, sizeof bs);
        bs[0].rep[0] = 1; bs[0].rep[1] = 4; bs[0].rep[2] = 8;
        if (srcSize == 0) { wr24(op, 1); op += 3; }                         /* one empty raw last block */
        while (remaining) {
            size_t const blockSize = optimalBlockSize(ip, remaining, blockSizeMax, savings, profile);
            U32 const lastBlock = blockSize == remaining;
            const blockstate_t* prev = &bs[cur]; blockstate_t* next = &bs[cur ^ 1];
            size_t cSize;
            {   /* ZSTD_window_enforceMaxDist(&ms->window, ip, maxDist, ...): since 1.5.0 the window is slid to the block's START
                 * (1.4.x passed ip + blockSize); match candidates are still bounded from the block's END through
                 * ZSTD_getLowestPrefixIndex(ms, endIndex, windowLog) in the block compressor.  Pinned by differential cases
                 * whose repcodes / matches lie between 2 MiB - blockSize and 2 MiB back (tests/golden/fuzz_regress). */
                U32 const blockEndIdx = (U32)(ip - ms.base), maxDist = 1u << cp.windowLog;
                if (blockEndIdx > maxDist) { U32 const newLow = blockEndIdx - maxDist; if (ms.dictLimit < newLow) ms.dictLimit = newLow; }
            }
            if (blockSize < 1 + 1 + 3 + 1 + 1) cSize = 0;               /* MIN_CBLOCK_SIZE + blockHeader + 1 + 1: don't even try */
            else {
                ss.nbSeq = 0; ss.litSize = 0;
                memcpy(next->rep, prev->rep, sizeof next->rep);
                {   size_t const lastLLSize = compressBlock_doubleFast(&ms, &ss, next->rep, ip, blockSize);
                    memcpy(ss.lit + ss.litSize, ip + blockSize - lastLLSize, lastLLSize); ss.litSize += lastLLSize; }
                if (g_tap) g_tap(g_tap_ctx, blockIndex, ss.seqs, ss.nbSeq, ss.litSize, blockSize);
                cSize = entropyCompressSeqStore(&ss, prev, next, cp.strategy, op + 3, blockSize);
                if (!isFirstBlock && ss.nbSeq < 4 && ss.litSize < 10 && isRLE(ip, blockSize)) { cSize = 1; op[3] = ip[0]; }
                if (cSize > 1) cur ^= 1;                                     /* confirm repcodes + entropy tables */
                if (bs[cur].of_repeat == FSE_repeat_valid) bs[cur].of_repeat = FSE_repeat_check;
            }
            if (cSize == 0) {                                                /* ZSTD_noCompressBlock */
                wr24(op, lastBlock + (0u << 1) + (U32)(blockSize << 3));
                memcpy(op + 3, ip, blockSize);
                cSize = 3 + blockSize;
            } else {
                U32 const h = cSize == 1 ? lastBlock + (1u << 1) + (U32)(blockSize << 3) : lastBlock + (2u << 1) + (U32)(cSize << 3);
                wr24(op, h);
                cSize += 3;
            }
            savings += (S64)blockSize - (S64)cSize;
            ip += blockSize; remaining -= blockSize; op += cSize; isFirstBlock = 0; blockIndex++;
        }
        free(ms.hashLong); free(ms.hashSmall); free(ss.seqs); free(ss.lit); free(ss.llCode); free(ss.mlCode); free(ss.ofCode);
    }
    return (size_t)(op - dst);
}
