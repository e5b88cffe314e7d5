// Double-fast match finder, FOUR chunks per wavefront: one chunk per 16-lane group (= one DPP row).  Included by zstd_enc.hip.
//
// match_block (zstd_enc.hip) gives a chunk a whole wave: a step's control flow is scalar, its ballots and lane reads are wave-wide, and
// of the 64 lanes 7 (first step) to 62 carry positions.  The chip then holds 24 dependency chains per CU and spends ~400 scalar
// instructions per sequence on the lane predicates of ONE chain (PMC).  Here the same serial algorithm - same tables, same
// entries, same insertions in the same order, hence the same bytes - runs as four independent chains in one instruction stream:
//   * everything that was wave-uniform (ip, anchor, offsets, step, K, the ring window ...) is GROUP-uniform and lives in vector registers;
//     control flow diverges per group through the exec mask, ballots are the group's 16 bits of the wave's ballot, lane reads go through
//     ds_bpermute within the row;
//   * lane roles inside a group as in match_block: lane 0 = the complementary insertion at curr + 2, lanes 1, 2 = those at ip - 2 / ip - 1,
//     lanes 3.. = K consecutive search positions (K <= 11), lane 3 + K the look-ahead of the "long match at +1" rule, lane 15 fetches the
//     bytes of the immediate-repcode check;
//   * a candidate is verified and extended by its group in one round trip: 16 lanes x 4 bytes = the same 64-byte span (8 behind, 56 ahead);
//   * every group has its own LDS ring of the chunk around ip and its own collision scoreboard.
// A group whose block is finished waits (masked out) for the wave's other groups; the entropy stage then runs chunk by chunk with all
// 64 lanes (zstd_compress_body4).
#pragma once

#ifndef Q_RING
#define Q_RING 2048u          /* LDS source window per chunk (bytes) */
#endif
#define Q_RWM (Q_RING / 4 - 1)
#ifndef Q_FILL
#define Q_FILL (Q_RING / 2)   /* refill granule */
#endif
#define Q_SAFE 256u           /* the parser wants [ip, ip + Q_SAFE) resident when a step starts */
#ifndef Q_SCR
#define Q_SCR 512u            /* slots of the intra-step hash-collision detector (per table, per chunk) */
#endif
#define Q_GROUPS 4u
#define Q_KMAX 11u            /* lanes 3..13 search, lane 3 + K (<= 14) looks ahead, lane 15 serves the immediate repcode */
#ifndef Q_K0
#define Q_K0 4u
#endif
#ifndef Q_K1
#define Q_K1 11u
#endif

#ifdef HIPEMU
#define QBALLOT(p) hipemu_group_ballot16((p) ? 1 : 0)
#define QREAD(v, j) hipemu_group_xchg16((uint32_t)(v), (int)(j))
#define QSYNC() hipemu::group_barrier()
#else
// the group's 16 bits of the wave ballot (lanes of other groups that sit in another branch contribute zeros: they are masked off)
#define QBALLOT(p) ((uint32_t)(__ballot(p) >> (lane & 48u)) & 0xFFFFu)
#define QREAD(v, j) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane & 48u) | (uint32_t)(j)) << 2), (int)(v)))
#define QSYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)   /* compiler-level only: a wave's LDS / memory instructions issue in order */
#endif

// what one group needs to parse one block of its chunk, and what it hands back (LDS, written / read by the whole wave around the call)
struct QArg {
    const uint8_t* src; uint32_t* hashLong; uint32_t* hashSmall; zs_seq* seqs;
    uint32_t srcSize, blockStart, blockSize, dictLimit;
    uint32_t rep[3];            // in: repcode history at block start; out: at block end
    uint32_t active;            // 0: this group has no block to parse in this round
    uint32_t nbSeq, litSize, lastLL, anchor;        // out (MfState)
};

struct QWin { uint32_t lo, hi; };     // group-uniform

__device__ static inline uint64_t qring8(const uint32_t* ring, uint32_t p) {
    uint64_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(ring) + (p & (Q_RING - 1)), 8); return v;
}
__device__ static inline uint32_t qring4(const uint32_t* ring, uint32_t p) {
    uint32_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(ring) + (p & (Q_RING - 1)), 4); return v;
}
__device__ static inline uint32_t qring1(const uint32_t* ring, uint32_t p) { return reinterpret_cast<const uint8_t*>(ring)[p & (Q_RING - 1)]; }

// Append chunk bytes [w.hi, w.hi + Q_FILL) to the group's ring: 16 lanes x 16 bytes per row of 256 (pieces that start beyond the chunk
// re-read the last valid piece: nothing outside the caller's buffer granule is touched).
__device__ __forceinline__ static void qwin_append(gbytes_t src, uint32_t lastPiece, uint32_t* ring, QWin& w, uint32_t gl) {
    uint4 v[Q_FILL / 256];
    QSYNC();
#pragma unroll
    for (uint32_t k = 0; k < Q_FILL / 256; k++) {
        uint32_t pp = w.hi + k * 256 + gl * 16;
        if (pp > lastPiece) pp = lastPiece;
        v[k] = ld128a(src + pp);
    }
#pragma unroll
    for (uint32_t k = 0; k < Q_FILL / 256; k++)
        *reinterpret_cast<uint4*>(&ring[((w.hi + k * 256 + gl * 16) >> 2) & Q_RWM]) = v[k];
    if ((w.hi & (Q_RING - 1)) == 0 && gl == 0) *reinterpret_cast<uint4*>(&ring[Q_RING / 4]) = v[0];    // the 16-byte mirror behind the ring
    w.hi += Q_FILL;
    if (w.hi - w.lo > Q_RING) w.lo = w.hi - Q_RING;
    QSYNC();
}
__device__ __forceinline__ static void qwin_ensure(gbytes_t src, uint32_t srcCeil, uint32_t lastPiece, uint32_t* ring, QWin& w, uint32_t ip, uint32_t gl) {
    if (ip < w.lo || ip > w.hi + Q_RING / 2) {                        // far jump: restart the ring behind ip
        QSYNC();
        const uint32_t base = ip > Q_FILL ? (ip - Q_FILL) & ~(Q_FILL - 1) : 0;
        w.lo = w.hi = base;
    }
    while (ip + Q_SAFE > w.hi && w.hi < srcCeil) qwin_append(src, lastPiece, ring, w, gl);
}

// four bytes of the chunk at p: from the ring when resident, else from global memory - whole when all four are wanted, byte by byte
// when some of them lie outside what may be read (chunk start / end)
__device__ __forceinline__ static uint32_t qbytes4(gbytes_t src, const uint32_t* ring, bool resident, uint32_t p, uint32_t want) {
    if (resident) return qring4(ring, p);
    if (want == 0xFu) return gld32(src + p);
    uint32_t v = 0;
    for (uint32_t k = 0; k < 4; k++) if ((want >> k) & 1u) v |= (uint32_t)src[p + k] << (8 * k);
    return v;
}
// bit k = byte k of x equals byte k of y
__device__ static inline uint32_t qeq_nibble(uint32_t x, uint32_t y) {
    const uint32_t d = x ^ y;
    return ((d & 0xFFu) == 0 ? 1u : 0u) | ((d & 0xFF00u) == 0 ? 2u : 0u) | ((d & 0xFF0000u) == 0 ? 4u : 0u) | ((d & 0xFF000000u) == 0 ? 8u : 0u);
}
// The group compares the 64 bytes [pa - nb, pa - nb + 64) with [pb - nb, ...) (pb < pa): lane j owns bytes 4j .. 4j + 3 of the span.
// Returns the lane's nibble: bit k = its byte k is wanted (`want`) and equal.
__device__ __forceinline__ static uint32_t qeq_span(gbytes_t src, const uint32_t* ring, const QWin w, uint32_t pa, uint32_t pb, uint32_t nb, uint32_t want, uint32_t gl) {
    const bool aR = pa >= w.lo + nb && pa + (64 - nb) <= w.hi;
    const bool bR = pb >= w.lo + nb && pb + (64 - nb) <= w.hi;
    const uint32_t ia = pa + 4 * gl - nb, ib = pb + 4 * gl - nb;
    uint32_t x = 0, y = 0;
    if (want) { y = qbytes4(src, ring, bR, ib, want); x = qbytes4(src, ring, aR, ia, want); }
    return qeq_nibble(x, y) & want;
}
__device__ static inline uint32_t qcto(uint32_t m) { return (uint32_t)__ffs((int)~m) - 1u; }      // trailing ones (m != 0xFFFFFFFF)
// equal bytes from span lane `from` on, given the lanes' nibbles e; up to 4 * (16 - from)
__device__ __forceinline__ static uint32_t qspan_fwd(uint32_t e, uint32_t from, uint32_t lane) {
    const uint32_t t = qcto(QBALLOT(e == 0xFu) >> from);               // whole lanes in front of the first lane with a differing (or unwanted) byte
    if (t == 16 - from) return 4 * t;
    return 4 * t + qcto(QREAD(e, from + t));
}

// number of equal bytes of chunk[a..] and chunk[b..] (b < a), not reading a-side bytes at or beyond iend: the continuation of a match
// beyond the 64 bytes the first comparison covers (16 lanes x 8 bytes per pass)
__device__ static uint32_t qcount(gbytes_t src, const uint32_t* ring, const QWin w, uint32_t a, uint32_t b, uint32_t iend, uint32_t gl, uint32_t lane) {
    uint32_t total = 0;
    for (;;) {
        const uint32_t off = total + 8 * gl;
        uint32_t n = 8;
        if (a + total + 8 * 16 <= iend) {                              // every lane compares 8 whole bytes
            uint64_t x;
            if (a + total >= w.lo && a + total + 8 * 16 <= w.hi && b + total >= w.lo) x = qring8(ring, a + off) ^ qring8(ring, b + off);
            else x = gld64(src + a + off) ^ gld64(src + b + off);
            n = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
        } else {
            const uint32_t avail = (a + off < iend) ? iend - (a + off) : 0;
            if (avail >= 8) {
                const uint64_t x = gld64(src + a + off) ^ gld64(src + b + off);
                n = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
            } else {
                n = 0;
                while (n < avail && src[a + off + n] == src[b + off + n]) n++;
            }
        }
        const uint32_t m = QBALLOT(n < 8);
        if (m) {
            const uint32_t fl = (uint32_t)__ffs((int)m) - 1;
            return total + 8 * fl + QREAD(n, fl);
        }
        total += 8 * 16;
    }
}
// backward extension: while (ip > anchor && match > low && chunk[ip - 1] == chunk[match - 1]), 16 bytes per pass
__device__ static uint32_t qcount_back(gbytes_t src, const uint32_t* ring, const QWin w, uint32_t ip, uint32_t match, uint32_t anchor, uint32_t low,
                                       uint32_t gl, uint32_t lane) {
    uint32_t lim = ip - anchor;
    if (match - low < lim) lim = match - low;
    if (lim == 0) return 0;
    uint32_t done = 0;
    for (;;) {
        const uint32_t i = done + gl;
        bool ok = i < lim;
        const uint32_t j = ok ? i : done;
        if (ip <= w.hi && ip - done >= w.lo + 16 && match - done >= w.lo + 16) ok = ok && qring1(ring, ip - 1 - j) == qring1(ring, match - 1 - j);
        else ok = ok && src[ip - 1 - j] == src[match - 1 - j];
        const uint32_t m = QBALLOT(!ok);
        if (m) return done + (uint32_t)__ffs((int)m) - 1;
        done += 16;
    }
}

// ---------------------------------------------------------------------------------------------------
// ZSTD_compressBlock_doubleFast_noDict_generic for up to four blocks at once (one per 16-lane group).  Line by line the group form of
// match_block: every wave-uniform value there is group-uniform here.
// ---------------------------------------------------------------------------------------------------
__device__ ZS_NOINLINE static void match_block4(QArg* __restrict__ qa, uint32_t* __restrict__ rings, uint8_t* __restrict__ scrs, const uint32_t lane, const uint32_t sched) {
    const uint32_t g = lane >> 4, gl = lane & 15u;
    QArg& A = qa[g];
    if (!A.active) return;
    uint32_t* const ring = rings + g * (Q_RING / 4 + 4);
    uint8_t* const scr = scrs + g * (2 * Q_SCR);
    uint32_t kFirst = (sched & 0xFF) ? (sched & 0xFF) : Q_K0, kSecond = ((sched >> 8) & 0xFF) ? ((sched >> 8) & 0xFF) : Q_K1;
    if (kFirst > Q_KMAX) kFirst = Q_KMAX;
    if (kSecond > Q_KMAX) kSecond = Q_KMAX;
    const gbytes_t gsrc = (gbytes_t)A.src;
    const gwords_t gL = (gwords_t)A.hashLong, gS = (gwords_t)A.hashSmall;
    ZS_GLOBAL zs_seq* const gseqs = (ZS_GLOBAL zs_seq*)A.seqs;
    const uint32_t srcSize = A.srcSize;
    const zs_cparams cp = zs_level3_cparams(srcSize);
    uint32_t nbSeq = 0, litSize = 0;
    const uint32_t iend = A.blockStart + A.blockSize, blockSize = A.blockSize, dictLimit = A.dictLimit, maxDist = 1u << cp.windowLog;
    const uint32_t plowIdx = (iend + 2 - dictLimit > maxDist) ? iend + 2 - maxDist : dictLimit;
    const uint32_t hBitsL = cp.hashLog, hBitsS = cp.chainLog, mls = cp.minMatch;
    const uint32_t srcCeil = (srcSize + Q_FILL - 1) & ~(Q_FILL - 1), lastPiece = (srcSize - 1) & ~15u;
    const uint32_t idxBits = 32u - (uint32_t)__clz((int)(srcSize + 2)), tagBits = 32u - idxBits, idxMask = (uint32_t)((1ull << idxBits) - 1);
    uint32_t ip = A.blockStart, anchor = ip;
    uint32_t off1 = A.rep[0], off2 = A.rep[1], sav1 = 0, sav2 = 0;
    if (ip + 2 == plowIdx) ip++;
    {   const uint32_t cur = ip + 2, windowLow = (cur - dictLimit > maxDist) ? cur - maxDist : dictLimit, maxRep = cur - windowLow;
        if (off2 > maxRep) { sav2 = off2; off2 = 0; }
        if (off1 > maxRep) { sav1 = off1; off1 = 0; }
    }
    QWin w; w.lo = w.hi = 0;
#define QSTORE_SEQ(ll_, lp_, ob_, ml_) do { if (gl == 0) zs_put_seq(&gseqs[nbSeq], (ob_), (ll_), (ml_) - 3, (lp_)); \
                                            litSize += (ll_); nbSeq++; } while (0)
#define QCOMP_INSERT() do { if (gl == 0) { if (!shadowL0) gL[hl] = eL; if (!shadowS0) gS[hs] = eS; } \
                            if (gl == 1) gL[hl] = eL; \
                            if (gl == 2) gS[hs] = eS; } while (0)
    if (blockSize >= 8) {
        const uint32_t ilimit = iend - 8;
        bool afterMatch = false;          // the immediate-repcode check (offset_2 at ip) of the match just stored is still due
        bool comp = false;                // ... and so are its complementary insertions (X = curr + 2, ip - 2, ip - 1)
        bool runStart = true;
        uint32_t X = 0, step = 1, nextStep = 0, width = kFirst;
        for (;;) {                                                    // one iteration per step of this group
            if (runStart) { step = 1; nextStep = ip + 256; width = kFirst; runStart = false; }
            uint32_t K = 0;
            const bool tail = ip + step > ilimit;
            if (tail) {
                if (!(ip <= ilimit && (comp || afterMatch))) break;
            } else {
                if (step == 1) {
                    K = ilimit - ip;
                    const uint32_t K1 = nextStep > ip + 1 ? nextStep - ip : 1;
                    if (K1 < K) K = K1;
                } else {
                    uint32_t K1 = 1;
                    if (nextStep > ip + step) K1 = (nextStep - ip - 1) / step + 1;
                    K = (ilimit - step - ip) / step + 1;
                    if (K1 < K) K = K1;
                }
                if (width < K) K = width;
            }
            if (ip + Q_SAFE > w.hi || ip < w.lo) qwin_ensure(gsrc, srcCeil, lastPiece, ring, w, ip, gl);
            // ---- positions, hashes, table entries ----
            const uint32_t pos = gl == 0 ? X : gl < 3 ? ip + gl - 3 : ip + (gl - 3) * step;
            const bool compL = comp && gl < 2, compS = comp && (gl == 0 || gl == 2);
            bool searching = gl >= 3 && gl < 3 + K;
            const bool lane3 = gl == 3;                               // ip itself: searched (K > 0) or only checked for the immediate repcode
            const bool mayUse = compL || compS || (gl >= 3 && gl <= 3 + K);
            const uint32_t spos = mayUse ? pos : ip;                  // an address every lane may read
            const bool posWin = ip + K * step + 8 <= w.hi && (!comp || (X >= w.lo && ip >= w.lo + 2));
            uint64_t d8;
            if (posWin) d8 = qring8(ring, spos); else d8 = gld64(gsrc + spos);
            const uint32_t hl = hash8(d8, hBitsL), hs = hashS(d8, hBitsS, mls);
            const uint32_t tL = tag8(d8, hBitsL, tagBits), tS = tag4((uint32_t)d8, tagBits);
            const uint32_t eL = ((tL << 1) << (idxBits - 1)) | (pos + 2), eS = ((tS << 1) << (idxBits - 1)) | (pos + 2);
            // ---- repcode pre-check: with the bytes at pos + 1 - off1 in the ring the first repcode hit is known before any probe ----
            const bool r1Near = K > 0 && off1 > 0 && posWin && ip + 1 >= w.lo + off1;
            uint32_t r1 = 0;
            if (r1Near) {
                r1 = qring4(ring, searching ? pos + 1 - off1 : ip);
                const uint32_t rb = QBALLOT(searching && r1 == (uint32_t)(d8 >> 8));
                if (rb) { const uint32_t fr = (uint32_t)__ffs((int)rb) - 1; K = fr - 2; searching = gl >= 3 && gl <= fr; }
            }
            // ---- two lanes, one bucket: find the first lane with an earlier partner and stop in front of it ----
            bool shadowL0 = false, shadowS0 = false;                  // lane 0's insertion is overwritten by lane 1's / lane 2's
            if (comp) {
                shadowL0 = QREAD(hl, 0) == QREAD(hl, 1);
                shadowS0 = QREAD(hs, 0) == QREAD(hs, 2);
            }
            bool flagLook = false;
            if (K > 0) {
                const bool partL = compL || (gl >= 3 && gl <= 3 + K), partS = compS || searching;
                const uint32_t sl = hl & (Q_SCR - 1), ss = Q_SCR + (hs & (Q_SCR - 1));
                QSYNC();
                if (partL) scr[sl] = (uint8_t)gl;
                if (partS) scr[ss] = (uint8_t)gl;
                QSYNC();
                uint32_t rL = partL ? scr[sl] : gl, rS = partS ? scr[ss] : gl;
                while (QBALLOT(gl < rL || gl < rS)) {                 // converge on the lowest lane id of every shared slot
                    QSYNC();
                    if (gl < rL) scr[sl] = (uint8_t)gl;
                    if (gl < rS) scr[ss] = (uint8_t)gl;
                    QSYNC();
                    rL = partL ? scr[sl] : gl; rS = partS ? scr[ss] : gl;
                }
                const uint32_t fb = QBALLOT(gl >= 3 && (rL < gl || rS < gl));
                if (fb) {
                    const uint32_t t = (uint32_t)__ffs((int)fb) - 1;
                    if (t == 3) {
                        // ip itself shares a slot with a complementary insertion: make those first, then search
                        QCOMP_INSERT();
                        comp = false;
                        continue;
                    }
                    if (t <= 3 + K) { K = t - 3; searching = searching && gl < t; flagLook = true; }      // lanes 3 .. t - 1 search, lane t looks ahead
                }
            }
            const uint32_t look = 3 + K;
            // ---- probes (K + 1 long, K short), the far bytes of the repcode checks ----
            const bool probeL = K > 0 && gl >= 3 && gl <= look, probeS = searching;
            const uint32_t hl3 = QREAD(hl, 3), hs3 = QREAD(hs, 3);
            uint32_t cL = 0, cS = 0;
            if (K > 0) {
                cL = gL[probeL ? hl : hl3];
                cS = gS[probeS ? hs : hs3];
            }
            const bool r2Near = afterMatch && posWin && ip >= w.lo + off2;
            const bool needFar = (K > 0 && off1 > 0 && !r1Near) || (afterMatch && !r2Near);
            uint32_t rfar = 0;
            if (needFar) {
                uint32_t fa = (searching && off1 > 0) ? pos + 1 - off1 : ip;
                if (gl == 15 && afterMatch) fa = ip - off2;
                rfar = gld32(gsrc + fa);
            }
            if (K > 0 && off1 > 0 && !r1Near) r1 = rfar;
            // ---- the immediate repcode of the previous match (offset_2 at ip) ----
            if (afterMatch) {
                afterMatch = false;
                const uint32_t r2 = r2Near ? qring4(ring, ip - off2) : QREAD(rfar, 15);
                const uint32_t d0 = QREAD((uint32_t)d8, 3);
                if (r2 == d0) {
                    const uint32_t a = ip + 4;
                    const uint32_t rem = iend - a;                     // a <= iend: ip <= ilimit
                    const uint32_t want = rem >= 4 * gl + 4 ? 0xFu : rem > 4 * gl ? (1u << (rem - 4 * gl)) - 1 : 0u;
                    uint32_t n = qspan_fwd(qeq_span(gsrc, ring, w, a, a - off2, 0, want, gl), 0, lane);
                    if (n == 64) n += qcount(gsrc, ring, w, a + 64, a + 64 - off2, iend, gl, lane);
                    const uint32_t rlen = 4 + n;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    if (comp) { QCOMP_INSERT(); comp = false; }
                    QSYNC();                                          // (emulator) the insertion at ip comes after the complementary ones
                    if (lane3) { gS[hs] = eS; gL[hl] = eL; }
                    QSTORE_SEQ(0, ip, 1, rlen);
                    ip += rlen; anchor = ip;
                    afterMatch = ip <= ilimit && off2 > 0;
                    runStart = true;
                    continue;
                }
            }
            // ---- the look-ahead lane sees the insertion an earlier lane of this step makes into its bucket ----
            if (flagLook) {
                const uint32_t hk = QREAD(hl, look);
                const uint32_t em = QBALLOT((compL || searching) && hl == hk && !(gl == 0 && shadowL0));
                if (em) {
                    const uint32_t e = 31u - (uint32_t)__clz((int)em); const uint32_t ee = QREAD(eL, e);
                    if (gl == look) cL = ee;
                }
            }
            // ---- events ----
            const uint32_t iL = cL & idxMask, iS = cS & idxMask;
            bool vL = probeL && iL >= plowIdx && ((cL ^ eL) & ~idxMask) == 0;        // in the window and same tag
            bool vS = probeS && iS >= plowIdx && ((cS ^ eS) & ~idxMask) == 0;
            const bool repOK = searching && off1 > 0 && r1 == (uint32_t)(d8 >> 8);
            int f = -1;
            uint32_t start = 0, mlen = 0, offBase = 0;
            bool isRep = false;
            for (;;) {
                const uint32_t ev = !searching ? 0u : repOK ? 1u : vL ? 2u : vS ? 3u : 0u;
                const uint32_t bm = QBALLOT(ev != 0);
                if (!bm) { f = -1; break; }
                f = __ffs((int)bm) - 1;
                const uint32_t evf = QREAD(ev, f);
                const uint32_t posf = ip + ((uint32_t)f - 3) * step;
                if (evf == 1) {                                       // repcode at posf + 1
                    start = posf + 1;
                    const uint32_t a = start + 4;
                    const uint32_t rem = iend > a ? iend - a : 0;
                    const uint32_t want = rem >= 4 * gl + 4 ? 0xFu : rem > 4 * gl ? (1u << (rem - 4 * gl)) - 1 : 0u;
                    uint32_t n = qspan_fwd(qeq_span(gsrc, ring, w, a, a - off1, 0, want, gl), 0, lane);
                    if (n == 64) n += qcount(gsrc, ring, w, a + 64, a + 64 - off1, iend, gl, lane);
                    mlen = 4 + n; offBase = 1; isRep = true;
                    break;
                }
                const uint32_t lowPos = plowIdx - 2;
                // the 64-byte span of a table candidate: 8 bytes behind the position (backward extension, lanes 0, 1), 56 ahead (lanes 2..15)
#define QSPAN_WANT(p_, lim_, want_) do { \
                    if (gl < 2) { const uint32_t far_ = 8 - 4 * gl; /* distance of the lane's byte 0 behind p_ */ \
                                  (want_) = (lim_) >= far_ ? 0xFu : (lim_) + 4 > far_ ? (0xFu << (far_ - (lim_))) & 0xFu : 0u; } \
                    else { const uint32_t o_ = 4 * gl - 8, rem_ = iend - (p_); /* p_ <= ilimit */ \
                           (want_) = rem_ >= o_ + 4 ? 0xFu : rem_ > o_ ? (1u << (rem_ - o_)) - 1 : 0u; } } while (0)
                if (evf == 2) {                                       // long match at posf
                    uint32_t mpos = QREAD(iL, f) - 2;
                    uint32_t lim = posf - anchor; if (mpos - lowPos < lim) lim = mpos - lowPos;
                    uint32_t want; QSPAN_WANT(posf, lim, want);
                    const uint32_t e = qeq_span(gsrc, ring, w, posf, mpos, 8, want, gl);
                    const uint32_t full = QBALLOT(e == 0xFu);
                    if ((full & 0xCu) != 0xCu) { if (gl == (uint32_t)f) vL = false; continue; }     // a tag's false positive
                    uint32_t fwd = qspan_fwd(e, 2, lane);
                    if (fwd == 56) fwd += qcount(gsrc, ring, w, posf + 56, mpos + 56, iend, gl, lane);
                    const uint32_t e1 = QREAD(e, 1), e0 = QREAD(e, 0);
                    uint32_t back = e1 == 0xFu ? 4 + ((uint32_t)__clz((int)~(e0 << 28))) : (uint32_t)__clz((int)~(e1 << 28));
                    if (back == 8 && lim > 8) back += qcount_back(gsrc, ring, w, posf - 8, mpos - 8, anchor, lowPos, gl, lane);
                    start = posf - back; mpos -= back; mlen = fwd + back;
                    offBase = start - mpos + 3;
                    break;
                }
                {                                                     // short match at posf; a strictly longer long match at +1 wins
                    uint32_t mpos = QREAD(iS, f) - 2;
                    uint32_t lim = posf - anchor; if (mpos - lowPos < lim) lim = mpos - lowPos;
                    uint32_t want; QSPAN_WANT(posf, lim, want);
                    uint32_t e = qeq_span(gsrc, ring, w, posf, mpos, 8, want, gl);
                    const uint32_t full = QBALLOT(e == 0xFu);
                    if ((full & 0x4u) != 0x4u) { if (gl == (uint32_t)f) vS = false; continue; }
                    uint32_t fwd = qspan_fwd(e, 2, lane);
                    if (fwd == 56) fwd += qcount(gsrc, ring, w, posf + 56, mpos + 56, iend, gl, lane);
                    uint32_t sp = posf;
                    if (QREAD((uint32_t)vL, f + 1)) {
                        const uint32_t p1 = posf + step, m1 = QREAD(iL, f + 1) - 2;
                        uint32_t lim1 = p1 - anchor; if (m1 - lowPos < lim1) lim1 = m1 - lowPos;
                        uint32_t want1; QSPAN_WANT(p1, lim1, want1);
                        const uint32_t ee = qeq_span(gsrc, ring, w, p1, m1, 8, want1, gl);
                        const uint32_t full1 = QBALLOT(ee == 0xFu);
                        if ((full1 & 0xCu) == 0xCu) {
                            uint32_t f1 = qspan_fwd(ee, 2, lane);
                            if (f1 == 56) f1 += qcount(gsrc, ring, w, p1 + 56, m1 + 56, iend, gl, lane);
                            if (f1 > fwd) { sp = p1; mpos = m1; fwd = f1; e = ee; lim = lim1; }
                        }
                    }
                    const uint32_t e1 = QREAD(e, 1), e0 = QREAD(e, 0);
                    uint32_t back = e1 == 0xFu ? 4 + ((uint32_t)__clz((int)~(e0 << 28))) : (uint32_t)__clz((int)~(e1 << 28));
                    if (back == 8 && lim > 8) back += qcount_back(gsrc, ring, w, sp - 8, mpos - 8, anchor, lowPos, gl, lane);
                    start = sp - back; mpos -= back; mlen = fwd + back;
                    offBase = start - mpos + 3;
                    break;
                }
            }
            // ---- commit: the visited positions insert themselves, then the pending complementary insertions ----
            const uint32_t lastIns = f >= 0 ? (uint32_t)f : 2 + K;
            if (gl >= 3 && gl <= lastIns) { gL[hl] = eL; gS[hs] = eS; }
            if (comp) { QCOMP_INSERT(); comp = false; }
            if (f < 0) {
                if (tail) break;
                const bool inc = ip + K * step >= nextStep;
                ip += K * step;
                if (inc) { step++; nextStep += 256; }
                width = width < kSecond ? kSecond : (width * 2 > Q_KMAX ? Q_KMAX : width * 2);
                continue;
            }
            if (!isRep) {
                off2 = off1; off1 = offBase - 3;
                QSYNC();                                              // (emulator) ... after the insertions of the visited positions
                if (step < 4 && gl == (uint32_t)f + 1) gL[hl] = eL;              // hashLong[hl1] = ip1
            }
            QSTORE_SEQ(start - anchor, anchor, offBase, mlen);
            X = ip + ((uint32_t)f - 3) * step + 2;                    // curr + 2
            ip = start + mlen; anchor = ip;
            comp = ip <= ilimit;
            afterMatch = comp && off2 > 0;
            runStart = true;
        }
    }
    sav2 = (sav1 != 0 && off1 != 0) ? sav1 : sav2;
    if (gl == 0) {
        A.rep[0] = off1 ? off1 : sav1;
        A.rep[1] = off2 ? off2 : sav2;
        A.nbSeq = nbSeq; A.lastLL = iend - anchor; A.anchor = anchor;
        A.litSize = litSize + (iend - anchor);
    }
#undef QSTORE_SEQ
#undef QCOMP_INSERT
#undef QSPAN_WANT
}
