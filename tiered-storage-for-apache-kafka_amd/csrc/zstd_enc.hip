// Zstandard level-3 frame compressor — gfx950, one 64-lane wavefront per chunk, up to 24 chunks resident per CU.
//
// Replaces zstd-jni's  new ZstdCompressCtx(); setPledgedSrcSize(n); setContentSize(true); compress(chunk)
//   core/src/main/java/io/aiven/kafka/tieredstorage/transform/CompressionChunkEnumeration.java:50-63
// and must emit, byte for byte, the frame libzstd emits for the same chunk (one-shot compression call, level 3:
// strategy dfast, windowLog <= 21, 128 KiB blocks, Huffman literals + FSE sequences; profile 1.5.7 adds that
// release's pre-block splitter).  The serial statement of the algorithm, pinned against the real library, is
// oracle/zstd_l3.c; this file is its wave-parallel form (DESIGN.md §5 has the measurements behind each choice):
//
//  * Match finding is an inherently serial greedy parse (every decision updates two hash tables and the repcode
//    history), so parallelism inside a chunk is SPECULATION: the wave evaluates K consecutive search positions at once
//    (K = 4 after a match, then 32, then 59 while nothing matches) - each lane hashes its position and probes both tables
//    (global memory: 512 KiB + 256 KiB per chunk cannot shrink into LDS without changing the output); a step whose lanes
//    share a bucket is cut in front of the second one (LDS scoreboard), a ballot picks the first lane with a match -
//    everything before it is exactly what the serial loop would have done - its table inserts are committed and the match
//    is verified and extended with one wave-wide 64-byte compare.
//  * A dependent global round trip costs a wave 1300-2000 cycles here, so the parser never reads global memory for bytes
//    near ip (LDS ring of the chunk around ip), never reads a candidate that cannot match (tags in the table entries),
//    verifies a far candidate and runs both extensions in one round trip, and keeps its state in SGPRs.
//  * Chunks are independent (fresh context per chunk in the reference); the kernel is latency bound per chunk and
//    HBM-random-access bound in aggregate, so it is shaped for residency: 80 VGPRs, 6.5 KiB LDS (parse-stage and
//    entropy-stage LDS alias), and callers keep several batches in flight.
//  * The entropy stage of each block: histograms, Huffman bit packing, literal gathering and the FSE sequence bit stream
//    run on all lanes (the three FSE state machines on three lanes, then prefix-summed bit packing); only the table
//    constructions (Huffman tree, FSE normalisation: a few thousand dependent steps per block) stay on lane 0.
//  * The wave also runs the stages either side of the compression when the batch asks for them (tsx_chain_fuse): CRC32C of
//    the source chunk before parsing it (crc_dev.h), AES-256-GCM over the finished frame (gcm_dev.h) - one launch per
//    batch; separate CRC / GCM launches starve for LDS on a chip full of compressor waves.
// This is byte-stream work: no MFMA.  Algorithmic traffic per chunk: N bytes read + transformed bytes written.
#include "zstd_common.h"
#ifdef ZS_DBG
#include <stdio.h>
#endif
#include "gcm_dev.h"
#include "crc_dev.h"

#define LANES 64
// "this value is the same in every lane": results of out-of-line calls and LDS broadcasts are divergent to the compiler;
// pinning the parser's state to SGPRs turns its control flow into scalar branches instead of exec-mask juggling.
#define UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(x)))
// A function that is not inlined into the kernel receives generic pointers (flat_load / flat_store: both wait counters, no
// scalar base).  The parser says what it knows: its tables, the chunk and the sequence array are global memory, and their base
// addresses are the same in every lane.
#ifdef HIPEMU
#define ZS_GLOBAL
#else
#define ZS_GLOBAL __attribute__((address_space(1)))
#endif
typedef const ZS_GLOBAL uint8_t* gbytes_t;
typedef ZS_GLOBAL uint32_t* gwords_t;
struct __attribute__((packed)) zs_u64u { uint64_t v; };
struct __attribute__((packed)) zs_u32u { uint32_t v; };
__device__ static inline uint64_t gld64(gbytes_t p) { return reinterpret_cast<const ZS_GLOBAL zs_u64u*>(p)->v; }
__device__ static inline uint32_t gld32(gbytes_t p) { return reinterpret_cast<const ZS_GLOBAL zs_u32u*>(p)->v; }
__device__ static inline uint4 ld128a(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }     // 16-byte aligned
#ifdef HIPEMU
struct alignas(16) zs_u32x4 { uint32_t x, y, z, w; };
#else
typedef uint32_t zs_u32x4 __attribute__((ext_vector_type(4)));
__device__ static inline uint4 ld128a(gbytes_t p) { const zs_u32x4 t = *reinterpret_cast<const ZS_GLOBAL zs_u32x4*>(p); return make_uint4(t.x, t.y, t.z, t.w); }
#endif
#ifdef HIPEMU
__device__ static inline void zs_put_seq(zs_seq* p, uint32_t offBase, uint32_t litLength, uint32_t mlBase, uint32_t litPos) { zs_seq q; q.offBase = offBase; q.litLength = litLength; q.mlBase = mlBase; q.litPos = litPos; *p = q; }
#else
__device__ static inline void zs_put_seq(ZS_GLOBAL zs_seq* p, uint32_t offBase, uint32_t litLength, uint32_t mlBase, uint32_t litPos) {
    zs_u32x4 t; t.x = offBase; t.y = litLength; t.z = mlBase; t.w = litPos;                    // field order of zs_seq
    *reinterpret_cast<ZS_GLOBAL zs_u32x4*>(p) = t;
}
#endif
template <class T> __device__ static inline T* uni_ptr(T* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = UNI((uint32_t)a), hi = UNI((uint32_t)(a >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}
#define ZS_RING 4096u         /* LDS source window of the parser (bytes) */
#define ZS_RWM (ZS_RING / 4 - 1)
#define ZS_FILL 2048u         /* refill granule */
#define ZS_SAFE 384u          /* the parser may touch [ip, ip + ZS_SAFE) between two refill checks */
#ifndef ZS_WAVES_PER_SIMD
#define ZS_WAVES_PER_SIMD 6   /* occupancy target: 80 VGPRs, <= 6826 B of LDS -> 24 chunks per CU */
#endif
#define ZS_SCR 1024u          /* slots of the intra-step hash-collision detector (per table) */
// cold, register-hungry scalar stages are kept out of line so the speculative match loop keeps its occupancy
#define ZS_NOINLINE __attribute__((noinline))

// ---- optional phase profile (make prof): lap timer, lane 0 attributes the cycles since the previous PT() to bucket k ----
#ifdef TSX_PROF2
// light-weight lap timers for the parser: scalar accumulators, no LDS traffic, no forced drains
#define LT_DECL unsigned long long lt_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lt_last_ = (unsigned long long)clock64();
#define LT(k) do { const unsigned long long n_ = (unsigned long long)clock64(); lt_[k] += n_ - lt_last_; lt_last_ = n_; } while (0)
#define LT_USE(v) do { asm volatile("" :: "s"(__builtin_amdgcn_readfirstlane((uint32_t)(v))) : "memory"); } while (0)
#define LT_FLUSH() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) g_prof[i_] += lt_[i_]; } while (0)
#else
#define LT_DECL
#define LT(k) do {} while (0)
#define LT_USE(v) do {} while (0)
#define LT_FLUSH() do {} while (0)
#endif
#ifdef TSX_PROF
__shared__ unsigned long long g_prof[24];
#ifdef TSX_PROF2
#define PT(k) do {} while (0)
#define PCNT(k, v) do {} while (0)
#define PTW(k) do {} while (0)
#else
#define PT(k) do { const unsigned long long now_ = (unsigned long long)clock64(); if (threadIdx.x == 0) { g_prof[k] += now_ - g_prof[23]; g_prof[23] = now_; } } while (0)
#define PCNT(k, v) do { if (threadIdx.x == 0) g_prof[k] += (v); } while (0)
#define PTW(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); PT(k); } while (0)   /* drain, then lap: stage latency */
#endif
__shared__ unsigned long long g_prof_take;
static unsigned long long* g_prof_out = nullptr;                      // device buffer: 24 u64 per chunk
extern "C" void tsx_debug_set_prof(void* dev_ptr) { g_prof_out = (unsigned long long*)dev_ptr; }
#else
#define PT(k) do {} while (0)
#define PCNT(k, v) do {} while (0)
#define PTW(k) do {} while (0)
#endif

// ---- format tables ------------------------------------------------------------------------------------
__device__ static const uint8_t kLLbits[36] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16};
__device__ static const uint8_t kMLbits[53] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16};
__device__ static const short kLLdefaultNorm[36] = {4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1};
__device__ static const short kOFdefaultNorm[29] = {1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1};
__device__ static const short kMLdefaultNorm[53] = {1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1};
__device__ static const uint8_t kLLcode[64] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
    22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24};
__device__ static const uint8_t kMLcode[128] = {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
    32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
    42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42};
__device__ static const uint32_t kRtb[8] = {0, 473195, 504333, 520860, 550000, 700000, 750000, 830000};

__device__ static inline uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }
__device__ static inline uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ static inline uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ static inline uint4 ld128(const uint8_t* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ static inline uint32_t LLcode(uint32_t ll) { return ll > 63 ? hb32(ll) + 19 : kLLcode[ll]; }
__device__ static inline uint32_t MLcode(uint32_t ml) { return ml > 127 ? hb32(ml) + 36 : kMLcode[ml]; }

__device__ static inline uint32_t hash8(uint64_t u, uint32_t h) { return (uint32_t)((u * 0xCF1BBCDCB7A56463ULL) >> (64 - h)); }
__device__ static inline uint32_t hashS(uint64_t u, uint32_t h, uint32_t mls) {
    if (mls == 5) return (uint32_t)(((u << 24) * 889523592379ULL) >> (64 - h));
    return ((uint32_t)u * 2654435761U) >> (32 - h);                       // mls == 4
}

// ---- tagged table entries ------------------------------------------------------------------------------------
// libzstd's tables hold indices only, so every probe costs a read of the candidate's bytes - here a random HBM line per
// probe, nearly all of them for candidates that do not match.  The tables are private to the kernel, so an entry also
// carries, in the bits above the index, a tag hashed from exactly the bytes the serial code compares (8 for the long
// table, 4 for the short one): equal bytes imply equal tags, so a probe whose tag differs is rejected without touching
// the candidate and the parse is unchanged.  idxBits = bits of (chunk size + 2); 9 tag bits for a 4 MiB chunk.
__device__ static inline uint32_t tag8(uint64_t u, uint32_t hBitsL, uint32_t tagBits) {       // bits right below the long index
    return tagBits ? (uint32_t)(((u * 0xCF1BBCDCB7A56463ULL) << hBitsL) >> (64 - tagBits)) : 0u;
}
__device__ static inline uint32_t tag4(uint32_t u, uint32_t tagBits) { return tagBits ? (u * 0x85EBCA6Bu) >> (32 - tagBits) : 0u; }

// ---- LDS state of one chunk (one wave per workgroup) --------------------------------------------------------
struct HufTable { uint16_t val[256]; uint8_t nb[256]; uint32_t tableLog, maxSym; };
struct FseTable { uint16_t state[512]; uint32_t dnb[56]; int32_t dfs[56]; uint32_t tableLog; };
struct NodeElt { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; };

struct EncLds {
    int hufRepeat[2];           // 0 none, 1 check
    uint32_t scal[16];          // lane-0 -> wave broadcast slots
    union alignas(16) {
        struct {                // entropy stage of a block
            // The two Huffman tables of the literal stage ([cur] = table of the previous compressed-literals block, [cur ^ 1] =
            // candidate) are dead while the sequences are coded and the LL table is dead while the literals are: they share their
            // bytes, and between two blocks the Huffman tables wait in the chunk's workspace (ZS_WS_HUFSAVE).  That is what brings
            // the wave's LDS under 160 KiB / 24: six chunks per SIMD instead of five.
            union { FseTable ll; HufTable huf[2]; };
            FseTable of;        // (the literal stage borrows it to FSE-code the Huffman weights)
            union {
                FseTable ml;            // sequence stage
                uint32_t hist[256];     // literal stage (and the pre-splitter): byte histogram, dead before ml is built
            };
            // second histogram (pre-splitter / sampling of the literals); sequence-code histograms; and, while the description of a new
            // Huffman table is written (huf_writeCTable: the sampling is over, the sequence stage has not begun), the table's weights
            // (bytes 0 .. 255) and their histogram (words 64 .. 79).  With those two arrays inside hist2 the wave's LDS is 6384 bytes: the
            // hardware allocates LDS in 1280-byte granules on gfx950, so 6704 bytes occupied 7680 and a CU held 21 chunks - not the 24 its
            // registers allow (measured: at most 5376 = 21 x 256 waves of a 6144-wave launch were ever resident at once).
            uint32_t hist2[256];
            uint8_t tableSymbol[512];
            uint16_t cumul[64];
            short norm[64];
        };
        struct {                // parse stage of a block (re-primed per block): source window + collision scoreboard
            uint32_t ring[ZS_RING / 4 + 4];     // + 16-byte mirror of the first bytes
            uint8_t scr[2 * ZS_SCR];
        } p;
        struct {                // GCM tail over the finished frame (gcm_encrypt_wave)
            tsx_gf128 tab[256];
            uint32_t t0[256];
        } g;
        uint32_t crcTab[4 * 256];       // CRC32C head over the source chunk (crc32c_wave)
    };
};

// ---------------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------------
// Cross-lane memory hand-off inside ONE wave (lane A's store observed by lane B's later load).  The hardware issues a
// wave's vector-memory / LDS instructions in order and keeps same-address order, so nothing is needed there; the fiber
// emulator (tests/emu) does not run lanes in lockstep and needs a rendezvous.
#ifdef HIPEMU
#define WAVE_MEM_SYNC() __threadfence_block()
#define VM_DRAIN() do {} while (0)
#define LOADED64(x) do {} while (0)
#else
#define WAVE_MEM_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)   /* compiler-level only */
#define VM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// "this value is consumed here": pins the wait for a global load inside the branch that issued it, so that the join with
// an LDS-sourced alternative does not inherit a vmcnt(0) (which would also drain every store still in flight).
#define LOADED64(x) do { uint32_t lo_ = (uint32_t)(x), hi_ = (uint32_t)((x) >> 32); asm volatile("" : "+v"(lo_), "+v"(hi_)); (x) = ((uint64_t)hi_ << 32) | lo_; } while (0)
#endif

// ---- the source window ------------------------------------------------------------------------------------
// Under 8 resident waves per CU every dependent global round trip costs a wave 800-2000 cycles (tools/ubench/lat.hip),
// and the serial parse needs the bytes around ip at every step: position hashing, repcode checks, match extension,
// complementary insertions.  So the parser keeps chunk bytes [lo, hi) (the last ~4 KiB and the next ~4 KiB) in an LDS
// ring, refilled 4 KiB at a time with coalesced 16-byte loads; only the hash tables and candidates older than the ring
// are read from global memory.  The ring aliases the entropy stage's scratch (EncLds) and is re-primed per block.
struct Win { uint32_t lo, hi; };      // wave-uniform

// The ring is ZS_RING bytes plus a 16-byte mirror of its first bytes, so an unaligned 8-byte read never has to wrap
// (gfx950 LDS reads need no alignment: one ds_read_b64 / ds_read_b32 each).
__device__ static inline uint64_t ring8(const uint32_t* ring, uint32_t p) {
    uint64_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(ring) + (p & (ZS_RING - 1)), 8); return v;
}
__device__ static inline uint32_t ring4(const uint32_t* ring, uint32_t p) {
    uint32_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(ring) + (p & (ZS_RING - 1)), 4); return v;
}
__device__ static inline uint32_t ring1(const uint32_t* ring, uint32_t p) { return reinterpret_cast<const uint8_t*>(ring)[p & (ZS_RING - 1)]; }

// Append chunk bytes [w.hi, w.hi + ZS_FILL) to the ring (16-byte pieces; pieces that start beyond the chunk are skipped
// by re-reading the last valid piece, so nothing outside the caller's buffer granule is touched).
template <class SP> __device__ __forceinline__ static void win_append(SP src, uint32_t lastPiece, uint32_t* ring, Win& w, uint32_t lane) {
    uint4 v[ZS_FILL / 1024];
    WAVE_MEM_SYNC();                                                  // (emulator) no lane may still be reading the slots replaced here
#pragma unroll
    for (uint32_t k = 0; k < ZS_FILL / 1024; k++) {
        uint32_t pp = w.hi + k * 1024 + lane * 16;
        if (pp > lastPiece) pp = lastPiece;
        v[k] = ld128a(src + pp);
    }
#pragma unroll
    for (uint32_t k = 0; k < ZS_FILL / 1024; k++)
        *reinterpret_cast<uint4*>(&ring[((w.hi + k * 1024 + lane * 16) >> 2) & ZS_RWM]) = v[k];
    if ((w.hi & (ZS_RING - 1)) == 0 && lane == 0) *reinterpret_cast<uint4*>(&ring[ZS_RING / 4]) = v[0];    // the mirror
    w.hi += ZS_FILL;
    if (w.hi - w.lo > ZS_RING) w.lo = w.hi - ZS_RING;
    WAVE_MEM_SYNC();
}
// make [ip, ip + ZS_SAFE) resident (or everything up to the end of the chunk)
template <class SP> __device__ __forceinline__ static void win_ensure(SP src, uint32_t srcCeil, uint32_t lastPiece, uint32_t* ring, Win& w,
                                                  uint32_t ip, uint32_t lane) {
    if (ip < w.lo || ip > w.hi + ZS_RING / 2) {                      // far jump: restart the ring behind ip
        WAVE_MEM_SYNC();
        const uint32_t base = ip > ZS_FILL ? (ip - ZS_FILL) & ~(ZS_FILL - 1) : 0;
        w.lo = w.hi = base;
    }
    while (ip + ZS_SAFE > w.hi && w.hi < srcCeil) win_append(src, lastPiece, ring, w, lane);
}

// number of equal bytes of src[a..] and src[b..] (b < a), not reading a-side bytes at or beyond iend: the continuation of a
// match beyond the 64 bytes the step's first comparison covers (rare; 64 lanes x 8 bytes per pass)
__device__ static uint32_t wave_count(const uint8_t* __restrict__ src, const uint32_t* ring, const Win w, uint32_t a, uint32_t b, uint32_t iend, uint32_t lane) {
    uint32_t total = 0;
    for (;;) {
        const uint32_t off = total + 8 * lane;
        uint32_t n = 8;
        if (a + total + 8 * LANES <= iend) {                         // every lane compares 8 whole bytes
            uint64_t x;
            if (a + total >= w.lo && a + total + 8 * LANES <= w.hi && b + total >= w.lo) x = ring8(ring, a + off) ^ ring8(ring, b + off);
            else { PCNT(21, 1); x = ld64(src + a + off) ^ ld64(src + b + off); }
            n = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
        } else {
            const uint32_t avail = (a + off < iend) ? iend - (a + off) : 0;
            if (avail >= 8) {
                uint64_t x = ld64(src + a + off) ^ ld64(src + b + off);
                n = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
            } else {
                n = 0;
                while (n < avail && src[a + off + n] == src[b + off + n]) n++;
            }
        }
        const unsigned long long m = __ballot(n < 8);
        if (m) {
            const int fl = __ffsll((long long)m) - 1;
            return total + 8 * (uint32_t)fl + __builtin_amdgcn_readlane(n, fl);
        }
        total += 8 * LANES;
    }
}

// backward extension: while (ip > anchor && match > low && src[ip-1] == src[match-1])
__device__ static uint32_t wave_count_back(const uint8_t* __restrict__ src, const uint32_t* ring, const Win w, uint32_t ip, uint32_t match, uint32_t anchor,
                                           uint32_t low, uint32_t lane) {
    uint32_t lim = ip - anchor;
    if (match - low < lim) lim = match - low;
    if (lim == 0) return 0;
    uint32_t done = 0;
    for (;;) {
        const uint32_t i = done + lane;
        bool ok = i < lim;
        const uint32_t j = ok ? i : done;
        if (ip <= w.hi && ip - done >= w.lo + LANES && match - done >= w.lo + LANES) ok = ok && ring1(ring, ip - 1 - j) == ring1(ring, match - 1 - j);
        else ok = ok && src[ip - 1 - i] == src[match - 1 - i];
        const unsigned long long m = __ballot(!ok);
        if (m) return done + (uint32_t)(__ffsll((long long)m) - 1);
        done += LANES;
    }
}

struct MfState { uint32_t nbSeq, litSize, lastLL, anchor; };

// The parser keeps the literals where they are: a sequence records where its literal run starts in the chunk and the
// entropy stage gathers them with all lanes (gather_literals) instead of copying on the serial critical path.

// ---------------------------------------------------------------------------------------------------
// double-fast match finder for one block (ZSTD_compressBlock_doubleFast_noDict_generic, speculative form).
// All "positions" are offsets within the chunk; table values are libzstd's indices = position + 2; rep[] is updated as the serial
// code does.  Everything that is the same for all lanes (ip, anchor, offsets, step...) is derived from ballots / readlanes so it
// lives in SGPRs and the control flow is scalar.  On log-like data 54 % of all sequences start at the FIRST position searched after
// the previous match and 69 % within two (tools/stats/parse_stats.c), and only a step's first event is ever used, hence:
//   * lane roles: lane 0 = the complementary insertion at curr + 2, lanes 1.. = consecutive positions from ip - 2 (lanes 1, 2 are
//     the complementary insertions at ip - 2 / ip - 1, lanes 3.. the K search positions, lane 3 + K the look-ahead for the "long
//     match at +1" rule, lane 63 fetches the bytes of the immediate-repcode check): the complementary insertions of the previous
//     match share the hash computation, the collision check and the store instructions of the next step;
//   * K starts at ZS_K0 (4) after a match and widens (ZS_K1 = 32, then 59) only while nothing is found;
//   * two lanes of a step that touch the same bucket are not patched up but avoided: a byte scoreboard in LDS (converging to the
//     lowest lane id per slot) finds the first lane with an earlier partner and the step is cut in front of it - before the table
//     loads are issued, so a cut costs no memory traffic; only the look-ahead lane is resolved exactly (one ballot);
//   * only the step's first (potential) event is verified: the wave compares the 64 bytes around that ONE candidate with the ring
//     (8 behind, 56 ahead: verification, forward and backward extension in one round trip, one byte per lane, one ballot); a
//     tag's false positive (1/512) is struck out and the next event of the same step taken.
// ---------------------------------------------------------------------------------------------------
#ifndef ZS_K0
#define ZS_K0 4u              /* search positions of the first step after a match */
#endif
#ifndef ZS_K1
#define ZS_K1 32u             /* ... of the second step; doubling from there */
#endif
#define ZS_KMAX 59u           /* lanes 3..61 search, 62 looks ahead, 63 serves the immediate repcode */
// the rare continuations (matches longer than the 64 bytes the first comparison covers) stay out of line
__device__ ZS_NOINLINE static uint32_t count_more(const uint8_t* __restrict__ src, const uint32_t* ring, const Win w, uint32_t a, uint32_t b, uint32_t iend, uint32_t lane) {
    return wave_count(src, ring, w, a, b, iend, lane);
}
__device__ ZS_NOINLINE static uint32_t count_more_back(const uint8_t* __restrict__ src, const uint32_t* ring, const Win w, uint32_t ip, uint32_t match, uint32_t anchor,
                                                       uint32_t low, uint32_t lane) {
    return wave_count_back(src, ring, w, ip, match, anchor, low, lane);
}

// bit i = lane i is valid and chunk byte pa + i - nb equals byte pb + i - nb (pb < pa).  Bytes come from the ring when the whole
// 64-byte span is resident, else from global memory (valid lanes only touch [0, srcSize)).
__device__ __forceinline__ static unsigned long long eq_mask(const gbytes_t src, const uint32_t* ring, const Win w, uint32_t pa, uint32_t pb,
                                                             uint32_t nb, bool valid, uint32_t lane) {
    const bool aR = pa >= w.lo + nb && pa + (64 - nb) <= w.hi;
    const bool bR = pb >= w.lo + nb && pb + (64 - nb) <= w.hi;
    const uint32_t ia = valid ? pa + lane - nb : pa, ib = valid ? pb + lane - nb : pb;
    uint32_t x, y;
    if (bR) y = ring1(ring, ib); else y = src[ib];
    if (aR) x = ring1(ring, ia); else x = src[ia];
    return __ballot(valid && x == y);
}
__device__ static inline uint32_t cto64(unsigned long long m) { return m == ~0ull ? 64u : (uint32_t)__ffsll((long long)~m) - 1; }


__device__ ZS_NOINLINE static void match_block(const uint8_t* __restrict__ src, const uint32_t srcSize_, const uint32_t blockStart,
                                                    const uint32_t blockSize_, uint32_t* __restrict__ hashLong, uint32_t* __restrict__ hashSmall,
                                                    const zs_cparams cp, const uint32_t dictLimitIn, uint32_t* rep, zs_seq* __restrict__ seqs,
                                                    MfState& ms, uint32_t* ring, uint8_t* scr, const uint32_t lane, const uint32_t sched) {
    // Speculation schedule (never changes the output, only what a search run costs): positions of the first step after a match, of the
    // second step; doubling from there.  sched = K0 | K1 << 8, 0 in a field = the compile-time default (4, 32).  Measured with 18-step
    // runs, three batches in flight / one at a time (profiles/r02_sweep_k_schedule.txt): (2,16) 17.6-17.9 GiB/s / 762 ms, (3,24) 18.5-18.6 /
    // 729-735, (4,32) 18.8 / 719, (4,48) 18.8 / 725, (6,32) 18.75 / 723: the dependent round trips a wider step saves are worth more than
    // the table lines it wastes, on a full chip too.  (Sweeps of 6 steps had said the opposite - their start-up transient dominates.)
    // Round 3 tried to bound a step by a PREDICTOR as well: a 256-byte recency filter over the 4-grams of the bytes already parsed says
    // for every search lane whether the tail of its 8-byte window has been seen lately (on log-like content the first event of a run sits
    // where the window turns from novel bytes into recurring ones).  On the exact parse it cuts the speculative reads from 15.8 to 5.3
    // per sequence at fewer steps (tools/stats/step_sim.c, profiles/r03_step_sim_K.txt) - and on the device it LOSES: 18.6-18.8 GiB/s
    // against 19.2-19.4 in flight, 9.6 against 10.3 one batch at a time (three alternating rounds, profiles/r03_gram_predictor_ab.txt),
    // like round 2's event-position predictor: ~16 instructions and two LDS round trips per step on the serial path cost more than the
    // table lines they save.  The code is in the history (commit "Parser: 4-gram recency predictor"), not in the kernel.
    const uint32_t kFirst = (UNI(sched) & 0xFF) ? (UNI(sched) & 0xFF) : ZS_K0, kSecond = ((UNI(sched) >> 8) & 0xFF) ? ((UNI(sched) >> 8) & 0xFF) : ZS_K1;
    const gbytes_t gsrc = (gbytes_t)uni_ptr(src);
    const gwords_t gL = (gwords_t)uni_ptr(hashLong), gS = (gwords_t)uni_ptr(hashSmall);
    ZS_GLOBAL zs_seq* const gseqs = (ZS_GLOBAL zs_seq*)uni_ptr(seqs);
    const uint32_t srcSize = UNI(srcSize_);
    uint32_t nbSeq = 0, litSize = 0;
    static_assert(sizeof(zs_seq) == 16, "zs_put_seq writes the four fields as one 16-byte store");
    const uint32_t iend = UNI(blockStart + blockSize_), blockSize = UNI(blockSize_), dictLimit = UNI(dictLimitIn), maxDist = 1u << UNI(cp.windowLog);
    const uint32_t plowIdx = (iend + 2 - dictLimit > maxDist) ? iend + 2 - maxDist : dictLimit;
    const uint32_t hBitsL = UNI(cp.hashLog), hBitsS = UNI(cp.chainLog), mls = UNI(cp.minMatch);
    const uint32_t srcCeil = (srcSize + ZS_FILL - 1) & ~(ZS_FILL - 1), lastPiece = (srcSize - 1) & ~15u;
    const uint32_t idxBits = 32u - (uint32_t)__clz((int)(srcSize + 2)), tagBits = 32u - idxBits, idxMask = (uint32_t)((1ull << idxBits) - 1);
    uint32_t ip = UNI(blockStart), anchor = ip;
    uint32_t off1 = UNI(rep[0]), off2 = UNI(rep[1]), sav1 = 0, sav2 = 0;
    if (ip + 2 == plowIdx) ip++;
    {   const uint32_t cur = ip + 2, windowLow = (cur - dictLimit > maxDist) ? cur - maxDist : dictLimit, maxRep = cur - windowLow;
        if (off2 > maxRep) { sav2 = off2; off2 = 0; }
        if (off1 > maxRep) { sav1 = off1; off1 = 0; }
    }
    Win w; w.lo = w.hi = 0;
#define STORE_SEQ(ll_, lp_, ob_, ml_) do { if (lane == 0) zs_put_seq(&gseqs[nbSeq], (ob_), (ll_), (ml_) - 3, (lp_)); \
                                           litSize += (ll_); nbSeq++; } while (0)
    if (blockSize >= 8) {
        const uint32_t ilimit = iend - 8;
        bool afterMatch = false;          // the immediate-repcode check (offset_2 at ip) of the match just stored is still due
        bool comp = false;                // ... and so are its complementary insertions (X = curr + 2, ip - 2, ip - 1)
        bool runStart = true;
        uint32_t X = 0, step = 1, nextStep = 0, width = kFirst;
        for (;;) {                                                    // one iteration per wave step
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
            if (ip + ZS_SAFE > w.hi || ip < w.lo) win_ensure(gsrc, srcCeil, lastPiece, ring, w, ip, lane);
            // ---- positions, hashes, table entries ----
            const uint32_t pos = lane == 0 ? X : lane < 3 ? ip + lane - 3 : ip + (lane - 3) * step;
            const bool compL = comp && lane < 2, compS = comp && (lane == 0 || lane == 2);
            bool searching = lane >= 3 && lane < 3 + K;
            const bool lane3 = lane == 3;                             // ip itself: searched (K > 0) or only checked for the immediate repcode
            const bool mayUse = compL || compS || (lane >= 3 && lane <= 3 + K);
            const uint32_t spos = mayUse ? pos : ip;                  // an address every lane may read
            const bool posWin = ip + K * step + 8 <= w.hi && (!comp || (X >= w.lo && ip >= w.lo + 2));
            uint64_t d8;
            if (posWin) d8 = ring8(ring, spos); else { d8 = gld64(gsrc + spos); LOADED64(d8); }
            const uint32_t hl = hash8(d8, hBitsL), hs = hashS(d8, hBitsS, mls);
            const uint32_t tL = tag8(d8, hBitsL, tagBits), tS = tag4((uint32_t)d8, tagBits);
            const uint32_t eL = ((tL << 1) << (idxBits - 1)) | (pos + 2), eS = ((tS << 1) << (idxBits - 1)) | (pos + 2);
            // ---- repcode pre-check: with the bytes at pos + 1 - off1 in the ring the first repcode hit is known before any probe ----
            const bool r1Near = K > 0 && off1 > 0 && posWin && ip + 1 >= w.lo + off1;
            uint32_t r1 = 0;
            if (r1Near) {
                r1 = ring4(ring, searching ? pos + 1 - off1 : ip);
                const unsigned long long rb = __ballot(searching && r1 == (uint32_t)(d8 >> 8));
                if (rb) { const uint32_t fr = (uint32_t)__ffsll((long long)rb) - 1; K = fr - 2; searching = lane >= 3 && lane <= fr; }
            }
            // ---- two lanes, one bucket: find the first lane with an earlier partner and stop in front of it ----
            bool shadowL0 = false, shadowS0 = false;                  // lane 0's insertion is overwritten by lane 1's / lane 2's
            if (comp) {
                shadowL0 = __builtin_amdgcn_readlane(hl, 0) == __builtin_amdgcn_readlane(hl, 1);
                shadowS0 = __builtin_amdgcn_readlane(hs, 0) == __builtin_amdgcn_readlane(hs, 2);
            }
            bool flagLook = false;
            if (K > 0) {
                const bool partL = compL || (lane >= 3 && lane <= 3 + K), partS = compS || searching;
                const uint32_t sl = hl & (ZS_SCR - 1), ss = ZS_SCR + (hs & (ZS_SCR - 1));
                WAVE_MEM_SYNC();
                if (partL) scr[sl] = (uint8_t)lane;
                if (partS) scr[ss] = (uint8_t)lane;
                WAVE_MEM_SYNC();
                uint32_t rL = partL ? scr[sl] : lane, rS = partS ? scr[ss] : lane;
                while (__any(lane < rL || lane < rS)) {               // converge on the lowest lane id of every shared slot
                    WAVE_MEM_SYNC();
                    if (lane < rL) scr[sl] = (uint8_t)lane;
                    if (lane < rS) scr[ss] = (uint8_t)lane;
                    WAVE_MEM_SYNC();
                    rL = partL ? scr[sl] : lane; rS = partS ? scr[ss] : lane;
                }
                const unsigned long long fb = __ballot(lane >= 3 && (rL < lane || rS < lane));
                if (fb) {
                    const uint32_t t = (uint32_t)__ffsll((long long)fb) - 1;
                    if (t == 3) {
                        // ip itself shares a slot with a complementary insertion: make those first, then search
                        if (lane == 0) { if (!shadowL0) gL[hl] = eL; if (!shadowS0) gS[hs] = eS; }
                        if (lane == 1) gL[hl] = eL;
                        if (lane == 2) gS[hs] = eS;
                        comp = false;
                        PCNT(19, 1);
                        continue;
                    }
                    if (t <= 3 + K) {                                 // lanes 3 .. t - 1 search, lane t looks ahead
                        if (t < 3 + K) PCNT(19, 1);
                        K = t - 3; searching = searching && lane < t; flagLook = true;
                    }
                }
            }
            const uint32_t look = 3 + K;
            // ---- probes (K + 1 long, K short), the far bytes of the repcode checks ----
            const bool probeL = K > 0 && lane >= 3 && lane <= look, probeS = searching;
            const uint32_t hl3 = __builtin_amdgcn_readlane(hl, 3), hs3 = __builtin_amdgcn_readlane(hs, 3);
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
                if (lane == 63 && afterMatch) fa = ip - off2;
                rfar = gld32(gsrc + fa);
            }
            if (K > 0 && off1 > 0 && !r1Near) r1 = rfar;
            PCNT(12, 1); PCNT(15, K);
            // ---- the immediate repcode of the previous match (offset_2 at ip) ----
            if (afterMatch) {
                afterMatch = false;
                const uint32_t r2 = r2Near ? ring4(ring, ip - off2) : (uint32_t)__builtin_amdgcn_readlane(rfar, 63);
                const uint32_t d0 = __builtin_amdgcn_readlane((uint32_t)d8, 3);
                if (UNI(r2) == d0) {
                    const uint32_t a = ip + 4;
                    uint32_t n = cto64(eq_mask(gsrc, ring, w, a, a - off2, 0, a + lane < iend, lane));
                    if (n == 64) n += UNI(count_more(src, ring, w, a + 64, a + 64 - off2, iend, lane));
                    const uint32_t rlen = 4 + n;
                    const uint32_t t = off2; off2 = off1; off1 = t;
                    if (comp) {
                        if (lane == 0) { if (!shadowL0) gL[hl] = eL; if (!shadowS0) gS[hs] = eS; }
                        if (lane == 1) gL[hl] = eL;
                        if (lane == 2) gS[hs] = eS;
                        comp = false;
                    }
                    WAVE_MEM_SYNC();                                  // (emulator) the insertion at ip comes after the complementary ones
                    if (lane3) { gS[hs] = eS; gL[hl] = eL; }
                    STORE_SEQ(0, ip, 1, rlen);
                    ip += rlen; anchor = ip;
                    PCNT(13, 1);
                    afterMatch = ip <= ilimit && off2 > 0;
                    runStart = true;
                    continue;
                }
            }
            // ---- the look-ahead lane sees the insertion an earlier lane of this step makes into its bucket ----
            if (flagLook) {
                const uint32_t hk = __builtin_amdgcn_readlane(hl, look);
                const unsigned long long em = __ballot((compL || searching) && hl == hk && !(lane == 0 && shadowL0));
                if (em) {
                    const uint32_t e = 63u - (uint32_t)__clzll((long long)em); const uint32_t ee = __builtin_amdgcn_readlane(eL, e);
                    if (lane == look) cL = ee;
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
                const unsigned long long bm = __ballot(ev != 0);
                if (!bm) { f = -1; break; }
                f = __ffsll((long long)bm) - 1;
                const uint32_t evf = __builtin_amdgcn_readlane(ev, f);
                const uint32_t posf = ip + ((uint32_t)f - 3) * step;
                if (evf == 1) {                                       // repcode at posf + 1
                    start = posf + 1;
                    const uint32_t a = start + 4;
                    uint32_t n = cto64(eq_mask(gsrc, ring, w, a, a - off1, 0, a + lane < iend, lane));
                    if (n == 64) n += UNI(count_more(src, ring, w, a + 64, a + 64 - off1, iend, lane));
                    mlen = 4 + n; offBase = 1; isRep = true;
                    break;
                }
                const uint32_t lowPos = plowIdx - 2;
                if (evf == 2) {                                       // long match at posf
                    uint32_t mpos = __builtin_amdgcn_readlane(iL, f) - 2;
                    uint32_t lim = posf - anchor; if (mpos - lowPos < lim) lim = mpos - lowPos;
                    const bool valid = lane < 8 ? (8 - lane) <= lim : posf + (lane - 8) < iend;
                    const unsigned long long m = eq_mask(gsrc, ring, w, posf, mpos, 8, valid, lane);
                    if (((m >> 8) & 0xFF) != 0xFF) { if (lane == (uint32_t)f) vL = false; PCNT(21, 1); continue; }     // a tag's false positive
                    uint32_t fwd = cto64(m >> 8);
                    if (fwd == 56) fwd += UNI(count_more(src, ring, w, posf + 56, mpos + 56, iend, lane));
                    uint32_t back = (uint32_t)__clz((int)~(((uint32_t)m & 0xFF) << 24));
                    if (back == 8 && lim > 8) back += UNI(count_more_back(src, ring, w, posf - 8, mpos - 8, anchor, lowPos, lane));
                    start = posf - back; mpos -= back; mlen = fwd + back;
                    offBase = start - mpos + 3;
                    break;
                }
                {                                                     // short match at posf; a strictly longer long match at +1 wins
                    uint32_t mpos = __builtin_amdgcn_readlane(iS, f) - 2;
                    uint32_t lim = posf - anchor; if (mpos - lowPos < lim) lim = mpos - lowPos;
                    const bool valid = lane < 8 ? (8 - lane) <= lim : posf + (lane - 8) < iend;
                    unsigned long long m = eq_mask(gsrc, ring, w, posf, mpos, 8, valid, lane);
                    if (((m >> 8) & 0xF) != 0xF) { if (lane == (uint32_t)f) vS = false; PCNT(21, 1); continue; }
                    uint32_t fwd = cto64(m >> 8);
                    if (fwd == 56) fwd += UNI(count_more(src, ring, w, posf + 56, mpos + 56, iend, lane));
                    uint32_t sp = posf;
                    if (__builtin_amdgcn_readlane((uint32_t)vL, f + 1)) {
                        const uint32_t p1 = posf + step, m1 = __builtin_amdgcn_readlane(iL, f + 1) - 2;
                        uint32_t lim1 = p1 - anchor; if (m1 - lowPos < lim1) lim1 = m1 - lowPos;
                        const bool valid1 = lane < 8 ? (8 - lane) <= lim1 : p1 + (lane - 8) < iend;
                        const unsigned long long mm = eq_mask(gsrc, ring, w, p1, m1, 8, valid1, lane);
                        if (((mm >> 8) & 0xFF) == 0xFF) {
                            uint32_t f1 = cto64(mm >> 8);
                            if (f1 == 56) f1 += UNI(count_more(src, ring, w, p1 + 56, m1 + 56, iend, lane));
                            if (f1 > fwd) { sp = p1; mpos = m1; fwd = f1; m = mm; lim = lim1; }
                        }
                    }
                    uint32_t back = (uint32_t)__clz((int)~(((uint32_t)m & 0xFF) << 24));
                    if (back == 8 && lim > 8) back += UNI(count_more_back(src, ring, w, sp - 8, mpos - 8, anchor, lowPos, lane));
                    start = sp - back; mpos -= back; mlen = fwd + back;
                    offBase = start - mpos + 3;
                    break;
                }
            }
            // ---- commit: the visited positions insert themselves, then the pending complementary insertions ----
            const uint32_t lastIns = f >= 0 ? (uint32_t)f : 2 + K;
            if (lane >= 3 && lane <= lastIns) { gL[hl] = eL; gS[hs] = eS; }
            if (comp) {
                if (lane == 0) { if (!shadowL0) gL[hl] = eL; if (!shadowS0) gS[hs] = eS; }
                if (lane == 1) gL[hl] = eL;
                if (lane == 2) gS[hs] = eS;
                comp = false;
            }
            if (f < 0) {
                if (tail) break;
                const bool inc = ip + K * step >= nextStep;
                ip += K * step;
                if (inc) { step++; nextStep += 256; }
                width = width < kSecond ? kSecond : (width * 2 > ZS_KMAX ? ZS_KMAX : width * 2);
                continue;
            }
            if (!isRep) {
                off2 = off1; off1 = offBase - 3;
                WAVE_MEM_SYNC();                                      // (emulator) ... after the insertions of the visited positions
                if (step < 4 && lane == (uint32_t)f + 1) gL[hl] = eL;              // hashLong[hl1] = ip1
            }
            STORE_SEQ(start - anchor, anchor, offBase, mlen);
            X = ip + ((uint32_t)f - 3) * step + 2;                    // curr + 2
            ip = UNI(start + mlen); anchor = ip;
            off1 = UNI(off1); off2 = UNI(off2);
            PCNT(13, 1);
            comp = ip <= ilimit;
            afterMatch = comp && off2 > 0;
            runStart = true;
        }
    }
    sav2 = (sav1 != 0 && off1 != 0) ? sav1 : sav2;
    rep[0] = off1 ? off1 : sav1;
    rep[1] = off2 ? off2 : sav2;
    ms.nbSeq = nbSeq; ms.lastLL = iend - anchor; ms.anchor = anchor;
    ms.litSize = litSize + ms.lastLL;
#undef STORE_SEQ
    PT(4);
}

// Per-lane copy of a short run with up to 32 bytes of loads in flight before the first store (a byte loop would pay one
// memory round trip per byte).
__device__ static inline void copy_run(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n) {
    uint32_t k = 0;
    while (n - k >= 32) {
        const uint64_t a = ld64(src + k), b = ld64(src + k + 8), c = ld64(src + k + 16), d = ld64(src + k + 24);
        __builtin_memcpy(dst + k, &a, 8); __builtin_memcpy(dst + k + 8, &b, 8); __builtin_memcpy(dst + k + 16, &c, 8); __builtin_memcpy(dst + k + 24, &d, 8);
        k += 32;
    }
    const uint32_t r = n - k, nq = r >> 3;
    uint64_t q0 = 0, q1 = 0, q2 = 0; uint32_t w = 0; uint16_t h = 0; uint8_t b1 = 0;
    if (nq > 0) q0 = ld64(src + k);
    if (nq > 1) q1 = ld64(src + k + 8);
    if (nq > 2) q2 = ld64(src + k + 16);
    const uint32_t t = k + 8 * nq;
    if (r & 4) __builtin_memcpy(&w, src + t, 4);
    if (r & 2) __builtin_memcpy(&h, src + t + (r & 4), 2);
    if (r & 1) b1 = src[t + (r & 6)];
    if (nq > 0) __builtin_memcpy(dst + k, &q0, 8);
    if (nq > 1) __builtin_memcpy(dst + k + 8, &q1, 8);
    if (nq > 2) __builtin_memcpy(dst + k + 16, &q2, 8);
    if (r & 4) __builtin_memcpy(dst + t, &w, 4);
    if (r & 2) __builtin_memcpy(dst + t + (r & 4), &h, 2);
    if (r & 1) dst[t + (r & 6)] = b1;
}

// Literals of a parsed block, gathered by all lanes: sequence u's run is src[litPos, litPos + litLength) and lands at the
// running sum of the earlier runs; the tail after the last match follows.
__device__ ZS_NOINLINE static void gather_literals(uint8_t* __restrict__ lit, const uint8_t* __restrict__ src, const zs_seq* __restrict__ seqs,
                                                   uint32_t nbSeq, uint32_t tailPos, uint32_t tailLen, uint32_t lane) {
    uint32_t base = 0;
    for (uint32_t g = 0; g < nbSeq; g += LANES) {
        const uint32_t u = g + lane;
        uint32_t ll = 0, lp = 0;
        if (u < nbSeq) { const zs_seq q = seqs[u]; ll = q.litLength; lp = q.litPos; }
        uint32_t incl = ll;
        for (int o = 1; o < LANES; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += t; }
        const uint32_t dst = base + incl - ll;
        copy_run(lit + dst, src + lp, ll);
        base += __shfl(incl, LANES - 1);
    }
    for (uint32_t i = lane; i < tailLen; i += LANES) lit[base + i] = src[tailPos + i];
}

// ---------------------------------------------------------------------------------------------------
// lane-0 serial pieces (FSE / Huffman table construction), work arrays in LDS
// ---------------------------------------------------------------------------------------------------
struct BitW { uint8_t* start; uint8_t* p; uint8_t* end; uint64_t acc; uint32_t n; bool overflow; };
__device__ static inline void bw_init(BitW& b, uint8_t* dst, uint8_t* end) { b.start = b.p = dst; b.end = end; b.acc = 0; b.n = 0; b.overflow = false; }
__device__ static inline void bw_add(BitW& b, uint64_t v, uint32_t nb) {
    if (!nb) return;
    b.acc |= (v & ((1ull << nb) - 1)) << b.n;
    b.n += nb;
    while (b.n >= 8) { if (b.p < b.end) *b.p++ = (uint8_t)b.acc; else b.overflow = true; b.acc >>= 8; b.n -= 8; }
}
__device__ static inline uint32_t bw_close(BitW& b) {
    bw_add(b, 1, 1);
    if (b.n) { if (b.p < b.end) *b.p++ = (uint8_t)b.acc; else b.overflow = true; b.n = 0; }
    return (uint32_t)(b.p - b.start);
}

__device__ static uint32_t fse_minTableLog(uint32_t srcSize, uint32_t maxSym) {
    uint32_t a = hb32(srcSize) + 1, b = hb32(maxSym) + 2;
    return a < b ? a : b;
}
__device__ static uint32_t fse_optimalTableLog(uint32_t maxTableLog, uint32_t srcSize, uint32_t maxSym, uint32_t minus) {
    uint32_t maxBitsSrc = hb32(srcSize - 1) - minus, tableLog = maxTableLog, minBits = fse_minTableLog(srcSize, maxSym);
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

__device__ ZS_NOINLINE static int fse_normalizeM2(short* norm, uint32_t tableLog, const uint32_t* cnt, uint32_t total, uint32_t maxSym, short lowProbCount) {
    const short NOT_YET = -2;
    uint32_t s, distributed = 0, toDist;
    const uint32_t lowThreshold = total >> tableLog;
    uint32_t lowOne = (uint32_t)(((uint64_t)total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (cnt[s] == 0) { norm[s] = 0; continue; }
        if (cnt[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= cnt[s]; continue; }
        if (cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; continue; }
        norm[s] = NOT_YET;
    }
    toDist = (1u << tableLog) - distributed;
    if (toDist == 0) return 0;
    if ((total / toDist) > lowOne) {
        lowOne = (uint32_t)(((uint64_t)total * 3) / (toDist * 2));
        for (s = 0; s <= maxSym; s++)
            if (norm[s] == NOT_YET && cnt[s] <= lowOne) { norm[s] = 1; distributed++; total -= cnt[s]; }
        toDist = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        uint32_t maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++) if (cnt[s] > maxC) { maxV = s; maxC = cnt[s]; }
        norm[maxV] += (short)toDist;
        return 0;
    }
    if (total == 0) {
        for (s = 0; toDist > 0; s = (s + 1) % (maxSym + 1)) if (norm[s] > 0) { toDist--; norm[s]++; }
        return 0;
    }
    {   const uint64_t vStepLog = 62 - tableLog, mid = (1ULL << (vStepLog - 1)) - 1;
        const uint64_t rStep = ((((uint64_t)1 << vStepLog) * toDist) + mid) / total;
        uint64_t tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == NOT_YET) {
                const uint64_t end = tmpTotal + (cnt[s] * rStep);
                const uint32_t weight = (uint32_t)(end >> vStepLog) - (uint32_t)(tmpTotal >> vStepLog);
                if (weight < 1) return -1;
                norm[s] = (short)weight;
                tmpTotal = end;
            }
        }
    }
    return 0;
}

__device__ ZS_NOINLINE static int fse_normalizeCount(short* norm, uint32_t tableLog, const uint32_t* cnt, uint32_t total, uint32_t maxSym, bool useLowProb) {
    const short lowProbCount = useLowProb ? -1 : 1;
    const uint64_t scale = 62 - tableLog, step = ((uint64_t)1 << 62) / total, vStep = 1ULL << (scale - 20);
    int still = 1 << tableLog;
    uint32_t s, largest = 0; short largestP = 0;
    const uint32_t lowThreshold = total >> tableLog;
    if (tableLog < fse_minTableLog(total, maxSym)) return -1;
    for (s = 0; s <= maxSym; s++) {
        if (cnt[s] == total) return 0;
        if (cnt[s] == 0) { norm[s] = 0; continue; }
        if (cnt[s] <= lowThreshold) { norm[s] = lowProbCount; still--; }
        else {
            short proba = (short)((cnt[s] * step) >> scale);
            if (proba < 8) { const uint64_t restToBeat = vStep * kRtb[proba]; proba += (cnt[s] * step) - ((uint64_t)proba << scale) > restToBeat; }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) { if (fse_normalizeM2(norm, tableLog, cnt, total, maxSym, lowProbCount) < 0) return -1; }
    else norm[largest] += (short)still;
    return (int)tableLog;
}

__device__ ZS_NOINLINE static uint32_t fse_writeNCount(uint8_t* out0, const short* norm, uint32_t maxSym, uint32_t tableLog) {
    uint8_t* out = out0;
    int nbBits, remaining, threshold; const int tableSize = 1 << tableLog;
    uint32_t bitStream = 0; int bitCount = 0; uint32_t symbol = 0; const uint32_t alphabetSize = maxSym + 1; int previousIs0 = 0;
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    remaining = tableSize + 1; threshold = tableSize; nbBits = (int)tableLog + 1;
    while (symbol < alphabetSize && remaining > 1) {
        if (previousIs0) {
            uint32_t start = symbol;
            while (symbol < alphabetSize && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) { start += 24; bitStream += 0xFFFFU << bitCount; out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int c = norm[symbol++];
            const int mx = (2 * threshold - 1) - remaining;
            remaining -= c < 0 ? -c : c;
            c++;
            if (c >= threshold) c += mx;
            bitStream += (uint32_t)c << bitCount;
            bitCount += nbBits;
            bitCount -= (c < mx);
            previousIs0 = (c == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (uint32_t)(out - out0);
}

__device__ ZS_NOINLINE static void fse_buildCTable(FseTable& ct, const short* norm, uint32_t maxSym, uint32_t tableLog, uint16_t* cumul, uint8_t* tableSymbol) {
    const uint32_t tableSize = 1u << tableLog, tableMask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t highThreshold = tableSize - 1, u;
    ct.tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = (uint16_t)(cumul[u - 1] + 1); tableSymbol[highThreshold--] = (uint8_t)(u - 1); }
        else cumul[u] = (uint16_t)(cumul[u - 1] + (uint32_t)norm[u - 1]);
    }
    cumul[maxSym + 1] = (uint16_t)(tableSize + 1);
    {   uint32_t position = 0;
        for (uint32_t symbol = 0; symbol <= maxSym; symbol++) {
            const int freq = norm[symbol];
            for (int n = 0; n < freq; n++) {
                tableSymbol[position] = (uint8_t)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    for (u = 0; u < tableSize; u++) { const uint8_t s = tableSymbol[u]; ct.state[cumul[s]++] = (uint16_t)(tableSize + u); }
    {   uint32_t total = 0;
        for (uint32_t s = 0; s <= maxSym; s++) {
            const int nv = norm[s];
            if (nv == 0) { ct.dnb[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct.dfs[s] = 0; }
            else if (nv == -1 || nv == 1) { ct.dnb[s] = (tableLog << 16) - (1u << tableLog); ct.dfs[s] = (int)(total - 1); total++; }
            else {
                const uint32_t maxBitsOut = tableLog - hb32((uint32_t)nv - 1), minStatePlus = (uint32_t)nv << maxBitsOut;
                ct.dnb[s] = (maxBitsOut << 16) - minStatePlus; ct.dfs[s] = (int)(total - (uint32_t)nv); total += (uint32_t)nv;
            }
        }
    }
}
__device__ static inline uint32_t fse_init2(const FseTable& ct, uint32_t symbol) {
    const uint32_t dnb = ct.dnb[symbol];
    const uint32_t nbBitsOut = (dnb + (1u << 15)) >> 16;
    const uint32_t value = (nbBitsOut << 16) - dnb;
    return ct.state[(value >> nbBitsOut) + ct.dfs[symbol]];
}
__device__ static inline void fse_encode(BitW& b, const FseTable& ct, uint32_t& value, uint32_t symbol) {
    const uint32_t nbBitsOut = (value + ct.dnb[symbol]) >> 16;
    bw_add(b, value, nbBitsOut);
    value = ct.state[(value >> nbBitsOut) + ct.dfs[symbol]];
}

// ---- Huffman table construction (HUF_buildCTable_wksp), on the wave -----------------------------------------------
// Until round 6 this ran on lane 0 alone with its work arrays (a 514-node array of structs) in the chunk's GLOBAL workspace: ~2000
// dependent global round trips per table, 1.6 M cycles per block on content K and 3.0 M on content B, whose 4 MiB chunks are 164 blocks
// under profile 1.5.7 - 14 % of a B chunk's time (profiles/r06_compressor_wave_laps_K_and_B.txt).  Now:
//  * the nodes are separate arrays in what is dead in LDS while a table is built - leaves (sorted by count, index = rank) and internal
//    nodes apart, `parent` and `nbBits` in ONE 16-bit array - 3.5 KB in `of` and in the tail of `ml` .. `norm`;
//  * HUF_sort is a counting sort on all lanes: rank histogram by LDS atomics, suffix sums by a wave scan, a symbol's slot = start of
//    its rank + the symbols before it with the same rank (the serial loop's stable order); only the quick sort INSIDE the log2 ranks
//    (counts >= 165: its order among equal counts is the algorithm's own) stays serial;
//  * the two-queue merge of HUF_buildTree is serial by nature (255 steps on lane 0); the depths follow by pointer jumping on all
//    lanes (8 rounds) instead of a 511-step walk; HUF_setMaxHeight (rare) stays serial;
//  * HUF_buildCTableFromTree: per-length counts by LDS atomics, a symbol's code = first code of its length + the symbols before it
//    with the same length.
// No array here is indexed dynamically in registers: a private array that is becomes scratch memory, i.e. global round trips.
#define RANK_TABLE 192
#define RANK_LOG_BEGIN 158
#define RANK_CUTOFF 165
struct HufWork {
    uint32_t* lcount;            // [-1 .. 255] leaves' counts in sorted order (lcount[-1] = the sentinel of HUF_buildTree)
    uint8_t* lbyte;              // [256] ... and their symbols
    uint32_t* icount;            // [256] internal nodes 256 .. 511 (tree build)
    uint16_t* par;               // [512] parent index, then nbBits (leaves 0 .. 255, internal nodes 256 .. 511)
    uint32_t* rank;              // [192] HUF_sort: symbols per rank, then the first slot of each rank      (aliases icount)
    uint8_t* idx8;               // [256] HUF_sort: rank of every symbol                                   (aliases icount)
    uint8_t* qstack;             // [192] explicit stack of the quick sort: 64 frames of (low, high, kind)
    uint16_t* dpt;               // [256] depths of the internal nodes while they are computed              (aliases icount, after the tree)
    uint32_t* rankLast;          // [14]  HUF_setMaxHeight                                                 (aliases icount, after the tree)
    uint32_t* nbPerRank;         // [16]  symbols per code length                                          (aliases icount, after the tree)
    uint16_t* valStart;          // [16]  first code of each length                                        (aliases icount, after the tree)
};
#define HUF_WORK_A_BYTES 1472u   /* icount 1024 (rank 768 + idx8 256 / dpt 512 + rankLast + nbPerRank + valStart), lbyte 256, qstack 192 */
#define HUF_WORK_B_BYTES 2052u   /* lcount 1028, par 1024 */
__device__ static inline HufWork huf_work(uint8_t* regA, uint8_t* regB) {
    HufWork W;
    W.icount = reinterpret_cast<uint32_t*>(regA); W.rank = W.icount; W.idx8 = regA + 768; W.lbyte = regA + 1024; W.qstack = regA + 1280;
    W.dpt = reinterpret_cast<uint16_t*>(regA); W.rankLast = reinterpret_cast<uint32_t*>(regA + 512); W.nbPerRank = reinterpret_cast<uint32_t*>(regA + 576);
    W.valStart = reinterpret_cast<uint16_t*>(regA + 640);
    W.lcount = reinterpret_cast<uint32_t*>(regB) + 1; W.par = reinterpret_cast<uint16_t*>(regB + 1028);
    return W;
}
__device__ static inline uint32_t huf_getIndex(uint32_t c) { return c < RANK_CUTOFF ? c : hb32(c) + RANK_LOG_BEGIN; }
__device__ static inline void huf_swap(const HufWork& W, int a, int b) {
    const uint32_t c = W.lcount[a]; const uint8_t y = W.lbyte[a];
    W.lcount[a] = W.lcount[b]; W.lbyte[a] = W.lbyte[b]; W.lcount[b] = c; W.lbyte[b] = y;
}
__device__ static void huf_insertionSort(const HufWork& W, int base, int low, int high) {
    const int size = high - low + 1;
    const int h = base + low;
    for (int i = 1; i < size; ++i) {
        const uint32_t keyC = W.lcount[h + i]; const uint8_t keyB = W.lbyte[h + i]; int j = i - 1;
        while (j >= 0 && W.lcount[h + j] < keyC) { W.lcount[h + j + 1] = W.lcount[h + j]; W.lbyte[h + j + 1] = W.lbyte[h + j]; j--; }
        W.lcount[h + j + 1] = keyC; W.lbyte[h + j + 1] = keyB;
    }
}
__device__ static int huf_partition(const HufWork& W, int base, int low, int high) {
    const uint32_t pivot = W.lcount[base + high]; int i = low - 1;
    for (int j = low; j < high; j++) if (W.lcount[base + j] > pivot) { i++; huf_swap(W, base + i, base + j); }
    huf_swap(W, base + i + 1, base + high);
    return i + 1;
}
// HUF_simpleQuickSort with its recursion made explicit (lane 0).  A CALL frame applies the insertion-sort threshold on
// entry; a CONTinuation frame is the rest of the caller's `while (low < high)` loop, which partitions without
// re-checking the threshold.  The two sides of a partition are disjoint, so their processing order is free.  Frames live in LDS
// (low, high < 256: a byte each; high may be low - 1 = -1: stored + 1).
__device__ ZS_NOINLINE static void huf_quickSort(const HufWork W, int base, int low0, int high0) {
    uint8_t* const st = W.qstack;
    int sp = 0;
#define QPUSH(lo_, hi_, call_) do { st[3 * sp] = (uint8_t)(lo_); st[3 * sp + 1] = (uint8_t)((hi_) + 1); st[3 * sp + 2] = (uint8_t)(call_); sp++; } while (0)
    QPUSH(low0, high0, 1);
    while (sp) {
        --sp;
        const int low = st[3 * sp], high = (int)st[3 * sp + 1] - 1; const bool call = st[3 * sp + 2] != 0;
        if (call && high - low < 8) { huf_insertionSort(W, base, low, high); continue; }
        if (!(low < high)) continue;
        const int idx = huf_partition(W, base, low, high);
        if (idx - low < high - idx) { QPUSH(idx + 1, high, 0); QPUSH(low, idx - 1, 1); }
        else { QPUSH(low, idx - 1, 0); QPUSH(idx + 1, high, 1); }
    }
#undef QPUSH
}

// All lanes call it; returns the table's depth (the same in every lane).  `bc`: three broadcast words in LDS.
__device__ ZS_NOINLINE static uint32_t huf_buildCTable(HufTable& ct, const uint32_t* cnt, const uint32_t maxSym, uint32_t maxNbBits, uint8_t* regA, uint8_t* regB, uint32_t* bc, const uint32_t lane) {
    const HufWork W = huf_work(regA, regB);
    // (huffNode[n] of the serial code: n < 256 a leaf - count lcount[n], symbol lbyte[n] -, else an internal node - count icount[n - 256];
    //  parent / nbBits of either kind par[n])
    const uint32_t n1 = maxSym + 1;
    for (uint32_t i = lane; i < RANK_TABLE; i += LANES) W.rank[i] = 0;
    for (uint32_t i = lane; i < 256; i += LANES) reinterpret_cast<uint32_t*>(W.par)[i] = 0;       // nbBits of the symbols that do not occur stays 0
    __syncthreads();
    // ---- HUF_sort ----
    uint32_t myC[4], myI[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t n = lane + 64u * (uint32_t)k;
        myC[k] = 0; myI[k] = 0xFFFFu;
        if (n < n1) { myC[k] = cnt[n]; myI[k] = huf_getIndex(myC[k]); W.idx8[n] = (uint8_t)myI[k]; atomicAdd(&W.rank[myI[k]], 1u); }
    }
    __syncthreads();
    {   // rank[k] <- symbols in ranks ABOVE k = the first slot of rank k (the serial code's rankPosition[k + 1].base)
        const uint32_t a = W.rank[3 * lane], b = W.rank[3 * lane + 1], c = W.rank[3 * lane + 2];
        uint32_t t = a + b + c;
        for (int o = 1; o < LANES; o <<= 1) { const uint32_t u = __shfl_down(t, o); if (lane + (uint32_t)o < LANES) t += u; }
        const uint32_t above = t - (a + b + c);
        __syncthreads();
        W.rank[3 * lane + 2] = above; W.rank[3 * lane + 1] = above + c; W.rank[3 * lane] = above + c + b;
    }
    __syncthreads();
    {   uint32_t before[4] = {0, 0, 0, 0};
        for (uint32_t m = 0; m < n1; m++) {
            const uint32_t v = W.idx8[m];
#pragma unroll
            for (int k = 0; k < 4; k++) before[k] += (v == myI[k]) & (m < lane + 64u * (uint32_t)k);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t n = lane + 64u * (uint32_t)k;
            if (n < n1) { const uint32_t pos = W.rank[myI[k]] + before[k]; W.lcount[pos] = myC[k]; W.lbyte[pos] = (uint8_t)n; }
        }
    }
    __syncthreads();
    if (lane == 0) {
        // (the serial code's bucket n holds the symbols of rank n - 1)
        for (int n = RANK_CUTOFF; n < RANK_TABLE - 1; ++n) {
            const int bucketStart = (int)W.rank[n - 1], bucketSize = (int)W.rank[n - 2] - bucketStart;
            if (bucketSize > 1) huf_quickSort(W, bucketStart, 0, bucketSize - 1);
        }
        // ---- HUF_buildTree (the rank arrays are dead from here: icount takes their place) ----
        int nonNullRank = (int)maxSym;
        const int STARTNODE = 256;
        int lowS, lowN, nodeNb = STARTNODE, n, nodeRoot;
        while (W.lcount[nonNullRank] == 0) nonNullRank--;
        lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
        W.icount[nodeNb - 256] = W.lcount[lowS] + W.lcount[lowS - 1];
        W.par[lowS] = W.par[lowS - 1] = (uint16_t)nodeNb;
        nodeNb++; lowS -= 2;
        for (n = nodeNb; n <= nodeRoot; n++) W.icount[n - 256] = 1u << 30;
        W.lcount[-1] = 1u << 31;
        while (nodeNb <= nodeRoot) {
            uint32_t ca, cb; int a, b;
            if (W.lcount[lowS] < W.icount[lowN - 256]) { a = lowS--; ca = W.lcount[a]; } else { a = lowN++; ca = W.icount[a - 256]; }
            if (W.lcount[lowS] < W.icount[lowN - 256]) { b = lowS--; cb = W.lcount[b]; } else { b = lowN++; cb = W.icount[b - 256]; }
            W.icount[nodeNb - 256] = ca + cb;
            W.par[a] = W.par[b] = (uint16_t)nodeNb;
            nodeNb++;
        }
        bc[0] = (uint32_t)nonNullRank; bc[1] = (uint32_t)nodeRoot;
    }
    __syncthreads();
    const uint32_t nonNullRank = bc[0], nodeRoot = bc[1], nIntern = nodeRoot - 255u;
    __syncthreads();
    {   // depths of the internal nodes by pointer jumping: par[256 + i] = an ancestor (relative index), dpt[i] = the distance to it; the root points at itself
        uint32_t j[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = lane + 64u * (uint32_t)k;
            if (i < nIntern) { const bool root = i + 256u == nodeRoot; W.dpt[i] = root ? 0 : 1; if (root) W.par[256 + i] = (uint16_t)i; else W.par[256 + i] = (uint16_t)(W.par[256 + i] - 256u); }
        }
        __syncthreads();
        for (int round = 0; round < 8; round++) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = lane + 64u * (uint32_t)k;
                j[k] = 0; d[k] = 0;
                if (i < nIntern) { const uint32_t a = W.par[256 + i]; d[k] = (uint32_t)W.dpt[i] + W.dpt[a]; j[k] = W.par[256 + a]; }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = lane + 64u * (uint32_t)k;
                if (i < nIntern) { W.dpt[i] = (uint16_t)d[k]; W.par[256 + i] = (uint16_t)j[k]; }
            }
            __syncthreads();
        }
        // leaves: one below their parent
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t n = lane + 64u * (uint32_t)k;
            if (n <= nonNullRank) W.par[n] = (uint16_t)(W.dpt[W.par[n] - 256u] + 1u);
        }
    }
    __syncthreads();
    uint16_t* const nbBits = W.par;                                    // leaves only from here
    if (lane == 0) {
        // ---- HUF_setMaxHeight ----
        const uint32_t lastNonNull = nonNullRank, targetNbBits = maxNbBits;
        const uint32_t largestBits = nbBits[lastNonNull];
        if (largestBits <= targetNbBits) maxNbBits = largestBits;
        else {
            int totalCost = 0; const uint32_t baseCost = 1u << (largestBits - targetNbBits); int n = (int)lastNonNull;
            while (nbBits[n] > targetNbBits) { totalCost += baseCost - (1 << (largestBits - nbBits[n])); nbBits[n] = (uint16_t)targetNbBits; n--; }
            while (nbBits[n] == targetNbBits) --n;
            totalCost >>= (largestBits - targetNbBits);
            {   const uint32_t noSymbol = 0xF0F0F0F0; uint32_t* const rankLast = W.rankLast;
                for (int i = 0; i < ZS_HUF_TABLELOG_MAX + 2; i++) rankLast[i] = noSymbol;
                {   uint32_t currentNbBits = targetNbBits;
                    for (int pos = n; pos >= 0; pos--) {
                        if (nbBits[pos] >= currentNbBits) continue;
                        currentNbBits = nbBits[pos];
                        rankLast[targetNbBits - currentNbBits] = (uint32_t)pos;
                    }
                }
                while (totalCost > 0) {
                    uint32_t nBitsToDecrease = hb32((uint32_t)totalCost) + 1;
                    for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                        const uint32_t highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
                        if (highPos == noSymbol) continue;
                        if (lowPos == noSymbol) break;
                        if (W.lcount[highPos] <= 2 * W.lcount[lowPos]) break;
                    }
                    while (nBitsToDecrease <= ZS_HUF_TABLELOG_MAX && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
                    totalCost -= 1 << (nBitsToDecrease - 1);
                    nbBits[rankLast[nBitsToDecrease]]++;
                    if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                    if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
                    else {
                        rankLast[nBitsToDecrease]--;
                        if (nbBits[rankLast[nBitsToDecrease]] != targetNbBits - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
                    }
                }
                while (totalCost < 0) {
                    if (rankLast[1] == noSymbol) {
                        while (nbBits[n] == targetNbBits) n--;
                        nbBits[n + 1]--;
                        rankLast[1] = (uint32_t)(n + 1);
                        totalCost++;
                        continue;
                    }
                    nbBits[rankLast[1] + 1]--;
                    rankLast[1]++;
                    totalCost++;
                }
            }
            maxNbBits = targetNbBits;
        }
        bc[2] = maxNbBits;
    }
    // ---- HUF_buildCTableFromTree ----
    if (lane < 16) W.nbPerRank[lane] = 0;
    for (uint32_t i = lane; i < 256; i += LANES) { ct.val[i] = 0; ct.nb[i] = 0; }
    __syncthreads();
    maxNbBits = bc[2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t n = lane + 64u * (uint32_t)k;
        if (n <= nonNullRank) atomicAdd(&W.nbPerRank[nbBits[n]], 1u);
        if (n < n1) ct.nb[W.lbyte[n]] = (uint8_t)nbBits[n];
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t mn = 0;
        W.valStart[0] = 0;
        for (int n = (int)maxNbBits; n > 0; n--) { W.valStart[n] = (uint16_t)mn; mn += W.nbPerRank[n]; mn >>= 1; }
        ct.tableLog = maxNbBits; ct.maxSym = maxSym;
    }
    __syncthreads();
    {   // ct.val[n] = valPerRank[ct.nb[n]]++ in symbol order: the first code of the length + the symbols before n with the same length
        uint32_t before[4] = {0, 0, 0, 0}, myNb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t n = lane + 64u * (uint32_t)k; myNb[k] = n < n1 ? ct.nb[n] : 0xFFu; }
        for (uint32_t m = 0; m < n1; m++) {
            const uint32_t v = ct.nb[m];
#pragma unroll
            for (int k = 0; k < 4; k++) before[k] += (v == myNb[k]) & (m < lane + 64u * (uint32_t)k);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t n = lane + 64u * (uint32_t)k; if (n < n1) ct.val[n] = (uint16_t)(W.valStart[myNb[k]] + before[k]); }
    }
    __syncthreads();
    return maxNbBits;
}

// HUF_compressWeights + HUF_writeCTable; returns header size, 0xFFFFFFFF when the table cannot be described
__device__ ZS_NOINLINE static uint32_t huf_writeCTable(uint8_t* dst, const HufTable& ct, uint32_t maxSym, uint32_t huffLog, EncLds& L) {
    uint8_t* const hw = reinterpret_cast<uint8_t*>(L.hist2);            // 256 weights + one pad byte ...
    for (uint32_t n = 0; n < maxSym; n++) { const uint32_t nb = ct.nb[n]; hw[n] = nb ? (uint8_t)(huffLog + 1 - nb) : 0; }
    uint32_t hSize = 0;
    {   // HUF_compressWeights(dst + 1, hw, maxSym)
        uint8_t* op = dst + 1; const uint32_t wtSize = maxSym;
        uint32_t maxSV = ZS_HUF_TABLELOG_MAX; uint32_t* cnt = L.hist2 + 68;    // ... and behind them the 13 counters of the weights' histogram
        if (wtSize > 1) {
            for (int i = 0; i <= ZS_HUF_TABLELOG_MAX; i++) cnt[i] = 0;
            for (uint32_t i = 0; i < wtSize; i++) cnt[hw[i]]++;
            while (!cnt[maxSV]) maxSV--;
            uint32_t maxCount = 0;
            for (uint32_t i = 0; i <= maxSV; i++) if (cnt[i] > maxCount) maxCount = cnt[i];
            if (maxCount == wtSize) hSize = 1;
            else if (maxCount == 1) hSize = 0;
            else {
                const uint32_t tableLog = fse_optimalTableLog(6, wtSize, maxSV, 2);
                if (fse_normalizeCount(L.norm, tableLog, cnt, wtSize, maxSV, false) < 0) return 0xFFFFFFFFu;
                op += fse_writeNCount(op, L.norm, maxSV, tableLog);
                fse_buildCTable(L.of, L.norm, maxSV, tableLog, L.cumul, L.tableSymbol);     // L.of is free until the sequence stage
                // FSE_compress_usingCTable: two interleaved states, from the last weight to the first
                if (wtSize <= 2) hSize = 0;
                else {
                    BitW b; bw_init(b, op, op + 512);
                    const uint8_t* ip = hw + wtSize; uint32_t s1, s2;
                    if (wtSize & 1) { s1 = fse_init2(L.of, *--ip); s2 = fse_init2(L.of, *--ip); fse_encode(b, L.of, s1, *--ip); }
                    else { s2 = fse_init2(L.of, *--ip); s1 = fse_init2(L.of, *--ip); }
                    while (ip > hw) { fse_encode(b, L.of, s2, *--ip); if (ip > hw) fse_encode(b, L.of, s1, *--ip); }
                    bw_add(b, s2, tableLog); bw_add(b, s1, tableLog);
                    op += bw_close(b);
                    hSize = (uint32_t)(op - (dst + 1));
                }
            }
        }
    }
    if ((hSize > 1) & (hSize < maxSym / 2)) { dst[0] = (uint8_t)hSize; return hSize + 1; }
    if (maxSym > 128) return 0xFFFFFFFFu;
    dst[0] = (uint8_t)(128 + (maxSym - 1));
    hw[maxSym] = 0;
    for (uint32_t n = 0; n < maxSym; n += 2) dst[(n / 2) + 1] = (uint8_t)((hw[n] << 4) + hw[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}

// ---- wave-parallel pieces ---------------------------------------------------------------------------------
__device__ static inline uint32_t wave_sum(uint32_t v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ static inline uint32_t wave_max(uint32_t v) { for (int o = 32; o; o >>= 1) { uint32_t t = __shfl_xor(v, o); v = t > v ? t : v; } return v; }
__device__ static inline uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
    uint32_t s = v;
    for (int o = 1; o < LANES; o <<= 1) { uint32_t t = __shfl_up(s, o); if (lane >= (uint32_t)o) s += t; }
    return s - v;
}

__device__ ZS_NOINLINE static void wave_histogram(uint32_t* hist, const uint8_t* __restrict__ p, uint32_t n, uint32_t lane) {
    for (uint32_t i = lane; i < 256; i += LANES) hist[i] = 0;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += LANES) atomicAdd(&hist[p[i]], 1u);
    __syncthreads();
}

// One Huffman stream (HUF_compress1X_usingCTable): symbols are written from the LAST to the first, LSB-first,
// closed by a 1 bit.  tmp is 4-byte aligned scratch; returns the stream size in bytes.
__device__ ZS_NOINLINE static uint32_t wave_huf_encode(uint32_t* __restrict__ tmp, const uint8_t* __restrict__ src, uint32_t n, const HufTable& ct, uint32_t lane) {
    const uint32_t per = (n + LANES - 1) / LANES;
    const uint32_t r0 = lane * per < n ? lane * per : n, r1 = r0 + per < n ? r0 + per : n;     // reversed index range
    uint32_t bits = 0;
    for (uint32_t r = r0; r < r1; r++) bits += ct.nb[src[n - 1 - r]];
    const uint32_t startBit = wave_excl_scan(bits, lane);
    const uint32_t total = __shfl(startBit + bits, LANES - 1);
    const uint32_t words = (total + 1 + 31) / 32;
    for (uint32_t w = lane; w < words; w += LANES) tmp[w] = 0;
    __threadfence_block();
    __syncthreads();
    {   uint64_t acc = 0; uint32_t word = startBit >> 5, nacc = startBit & 31;
        for (uint32_t r = r0; r < r1; r++) {
            const uint8_t s = src[n - 1 - r];
            acc |= (uint64_t)ct.val[s] << nacc;
            nacc += ct.nb[s];
            if (nacc >= 32) { atomicOr(&tmp[word], (uint32_t)acc); acc >>= 32; nacc -= 32; word++; }
        }
        if (lane == LANES - 1) { acc |= 1ull << nacc; nacc++; }                                 // end mark
        if (nacc) atomicOr(&tmp[word], (uint32_t)acc);
        if (nacc > 32) atomicOr(&tmp[word + 1], (uint32_t)(acc >> 32));
    }
    __threadfence_block();
    __syncthreads();
    return (total + 1 + 7) / 8;
}

__device__ static inline void wave_copy(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint32_t n, uint32_t lane) {
    for (uint32_t i = lane; i < n; i += LANES) dst[i] = src[i];
}

// HUF_compress1X / 4X_usingCTable + the compressibility check of HUF_compressCTable_internal.
// Writes at op (inside blockout); returns the total size from ostart, 0 if not compressible.
__device__ ZS_NOINLINE static uint32_t wave_huf_compress(uint8_t* ostart, uint8_t* op, const uint8_t* __restrict__ lit, uint32_t n, bool single, const HufTable& ct,
                                             uint32_t* tmp, uint32_t lane) {
    if (single) {
        const uint32_t c = wave_huf_encode(tmp, lit, n, ct, lane);
        wave_copy(op, (const uint8_t*)tmp, c, lane);
        op += c;
    } else {
        if (n < 12) return 0;
        const uint32_t seg = (n + 3) / 4;
        uint8_t* const jump = op;
        op += 6;
        for (int i = 0; i < 4; i++) {
            const uint32_t len = i < 3 ? seg : n - 3 * seg;
            const uint32_t c = wave_huf_encode(tmp, lit + (uint32_t)i * seg, len, ct, lane);
            if (c == 0 || c > 65535) return 0;
            if (i < 3 && lane == 0) { jump[2 * i] = (uint8_t)c; jump[2 * i + 1] = (uint8_t)(c >> 8); }
            wave_copy(op, (const uint8_t*)tmp, c, lane);
            op += c;
            __syncthreads();
        }
    }
    const uint32_t tot = (uint32_t)(op - ostart);
    if (tot >= n - 1) return 0;
    return tot;
}

// ---------------------------------------------------------------------------------------------------
// literals section (ZSTD_compressLiterals).  Returns its size; updates L.huf / L.hufRepeat ("next" side).
// cur = index of the confirmed (previous) Huffman state; the candidate state is written at cur ^ 1.
// ---------------------------------------------------------------------------------------------------
__device__ static uint32_t write_raw_literals(uint8_t* dst, const uint8_t* lit, uint32_t n, uint32_t lane) {
    const uint32_t fl = 1 + (n > 31) + (n > 4095);
    WAVE_MEM_SYNC();                // the fallback overwrites what a Huffman attempt left at dst: other lanes' earlier stores come first
    if (lane == 0) {
        if (fl == 1) dst[0] = (uint8_t)(0 + (n << 3));
        else if (fl == 2) { const uint32_t v = 0 + (1 << 2) + (n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); }
        else { const uint32_t v = 0 + (3 << 2) + (n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); }
    }
    wave_copy(dst + fl, lit, n, lane);
    return fl + n;
}
__device__ static uint32_t write_rle_literals(uint8_t* dst, const uint8_t* lit, uint32_t n, uint32_t lane) {
    const uint32_t fl = 1 + (n > 31) + (n > 4095);
    WAVE_MEM_SYNC();
    if (lane == 0) {
        if (fl == 1) dst[0] = (uint8_t)(1 + (n << 3));
        else if (fl == 2) { const uint32_t v = 1 + (1 << 2) + (n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); }
        else { const uint32_t v = 1 + (3 << 2) + (n << 4); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); }
        dst[fl] = lit[0];
    }
    return fl + 1;
}

__device__ ZS_NOINLINE static uint32_t compress_literals(uint8_t* dst, const uint8_t* __restrict__ lit, uint32_t n, EncLds& L, int cur, bool suspectUncompressible,
                                             uint32_t* tmp, uint32_t lane) {
    const int nxt = cur ^ 1;
    const uint32_t lhSize = 3 + (n >= 1024) + (n >= 16384);
    bool single = n < 256;
    // "next" starts as a copy of "prev" (nothing to copy: we only switch `cur` when a new table is adopted)
    if (n < 64) return write_raw_literals(dst, lit, n, lane);              // ZSTD_minLiteralsToCompress (dfast, no valid repeat)
    const int prevRepeat = L.hufRepeat[cur];
    const bool preferRepeat = n <= 1024;                                   // strategy < lazy && srcSize <= 1024
    uint8_t* const ostart = dst + lhSize;
    // ---- HUF_compress_internal ----
    uint32_t cLit = 0; bool usedOld = false, newTable = false;
    bool decided = false;
    if (suspectUncompressible && n >= 40960) {                             // sample the first and last 4 KiB
        wave_histogram(L.hist2, lit, 4096, lane);
        uint32_t m = 0; for (uint32_t i = lane; i < 256; i += LANES) m = L.hist2[i] > m ? L.hist2[i] : m;
        uint32_t largestTotal = wave_max(m);
        __syncthreads();
        wave_histogram(L.hist2, lit + n - 4096, 4096, lane);
        m = 0; for (uint32_t i = lane; i < 256; i += LANES) m = L.hist2[i] > m ? L.hist2[i] : m;
        largestTotal += wave_max(m);
        if (largestTotal <= ((2 * 4096) >> 7) + 4) { cLit = 0; decided = true; }
    }
    uint32_t maxSym = 255, largest = 0;
    if (!decided) {
        PT(5);
        wave_histogram(L.hist, lit, n, lane);
        uint32_t m = 0, top = 0;
        for (uint32_t i = lane; i < 256; i += LANES) { const uint32_t c = L.hist[i]; if (c > m) m = c; if (c) top = i; }
        largest = wave_max(m); maxSym = wave_max(top);
        if (largest == n) { cLit = 1; decided = true; if (lane == 0) ostart[0] = lit[0]; }
        else if (largest <= (n >> 7) + 4) { cLit = 0; decided = true; }
    }
    if (!decided) {
        int repeat = prevRepeat;
        if (repeat == 1) {                                                 // HUF_validateCTable
            bool bad = L.huf[cur].maxSym < maxSym;
            for (uint32_t i = lane; i <= maxSym; i += LANES) bad |= (L.hist[i] != 0) & (L.huf[cur].nb[i] == 0);
            if (__any(bad)) repeat = 0;
        }
        if (preferRepeat && repeat != 0) {
            cLit = wave_huf_compress(ostart, ostart, lit, n, single, L.huf[cur], tmp, lane);
            usedOld = true;
        } else {
            // build the candidate table (lane 0), describe it, compare with reusing the old one
            PT(5);
            // the work arrays in what is dead right now (huf_buildCTable): `of` (free until the weights are FSE-coded, below), and everything
            // behind the counts in `hist` up to the end of `norm` (tail of `ml`, `hist2`, `tableSymbol`, `cumul`, `norm`)
            static_assert(sizeof(L.of) >= HUF_WORK_A_BYTES, "the first group of work arrays fits the OF table");
            static_assert(offsetof(EncLds, norm) + sizeof(((EncLds*)0)->norm) - (offsetof(EncLds, hist) + sizeof(((EncLds*)0)->hist)) >= HUF_WORK_B_BYTES && (offsetof(EncLds, hist) & 3) == 0,
                          "lcount[-1 .. 255] + par[512] fit behind the byte histogram");
            uint32_t huffLog = fse_optimalTableLog(ZS_LitHufLog, n, maxSym, 1);
            huffLog = UNI(huf_buildCTable(L.huf[nxt], L.hist, maxSym, huffLog, reinterpret_cast<uint8_t*>(&L.of), reinterpret_cast<uint8_t*>(L.hist) + sizeof(L.hist), &L.scal[4], lane));
            uint32_t oldBits = 0, newBits = 0;                             // (what reusing the old table / using the new one would cost: all lanes)
            if (repeat != 0) {
                for (uint32_t s_ = lane; s_ <= maxSym; s_ += LANES) { oldBits += L.huf[cur].nb[s_] * L.hist[s_]; newBits += L.huf[nxt].nb[s_] * L.hist[s_]; }
                oldBits = wave_sum(oldBits); newBits = wave_sum(newBits);
            }
            if (lane == 0) {
                const uint32_t hSize = huf_writeCTable(ostart, L.huf[nxt], maxSym, huffLog, L);
                uint32_t useOld = 0, fail = 0;
                if (hSize == 0xFFFFFFFFu) fail = 1;
                else {
                    if (repeat != 0) {
                        if ((oldBits >> 3) <= hSize + (newBits >> 3) || hSize + 12 >= n) useOld = 1;
                    }
                    if (!useOld && hSize + 12 >= n) fail = 1;
                }
                L.scal[0] = hSize; L.scal[1] = useOld; L.scal[2] = fail;
            }
            __syncthreads();
            const uint32_t hSize = L.scal[0]; const bool useOld = L.scal[1], fail = L.scal[2];
            __syncthreads();
            PT(6);
            if (fail) cLit = 0;
            else if (useOld) { cLit = wave_huf_compress(ostart, ostart, lit, n, single, L.huf[cur], tmp, lane); usedOld = true; }
            else { cLit = wave_huf_compress(ostart, ostart + hSize, lit, n, single, L.huf[nxt], tmp, lane); newTable = true; }
        }
    }
    // ---- back in ZSTD_compressLiterals ----
    PT(7);
    const uint32_t minGain = (n >> 6) + 2;
    if (cLit == 0 || cLit >= n - minGain) return write_raw_literals(dst, lit, n, lane);
    if (cLit == 1) return write_rle_literals(dst, lit, n, lane);           // n >= 64 here, so (srcSize >= 8) holds
    const uint32_t hType = (usedOld && !newTable) ? 3u : 2u;                // set_repeat : set_compressed
    if (newTable) L.scal[8] = 1;                                           // caller adopts huf[nxt] if the block is kept
    if (lane == 0) {
        if (lhSize == 3) { const uint32_t v = hType + ((uint32_t)(!single) << 2) + (n << 4) + (cLit << 14); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); }
        else if (lhSize == 4) { const uint32_t v = hType + (2 << 2) + (n << 4) + (cLit << 18); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); dst[3] = (uint8_t)(v >> 24); }
        else { const uint32_t v = hType + (3 << 2) + (n << 4) + (cLit << 22); dst[0] = (uint8_t)v; dst[1] = (uint8_t)(v >> 8); dst[2] = (uint8_t)(v >> 16); dst[3] = (uint8_t)(v >> 24); dst[4] = (uint8_t)(cLit >> 10); }
    }
    return lhSize + cLit;
}

// ---------------------------------------------------------------------------------------------------
// sequences section (ZSTD_buildSequencesStatistics + ZSTD_encodeSequences).  Returns bytes written at op,
// 0xFFFFFFFF if the block must be emitted raw.
// ---------------------------------------------------------------------------------------------------
__device__ static int select_encoding(uint32_t mostFrequent, uint32_t nbSeq, uint32_t defaultNormLog, bool defaultAllowed) {
    if (mostFrequent == nbSeq) return (defaultAllowed && nbSeq <= 2) ? 0 : 1;           // set_basic : set_rle
    if (defaultAllowed) {
        const uint32_t dynMin = ((1u << defaultNormLog) * 8u) >> 3;                      // mult = 10 - strategy(2)
        if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defaultNormLog - 1))) return 0;   // set_basic
    }
    return 2;                                                                            // set_compressed
}

// lane 0: one of LL / OF / ML.  Returns description size (0xFFFFFFFF on failure); *type receives the mode.
__device__ ZS_NOINLINE static uint32_t build_seq_table(uint8_t* op, FseTable& ct, uint32_t FSELog, uint32_t* cnt, uint32_t maxSymStart, const uint8_t* codes, uint32_t nbSeq,
                                           const short* defaultNorm, uint32_t defaultNormLog, uint32_t defaultMax, bool isOffsets, EncLds& L, uint32_t* type) {
    uint32_t max = maxSymStart;
    while (!cnt[max]) max--;
    uint32_t mostFrequent = 0;
    for (uint32_t s = 0; s <= max; s++) if (cnt[s] > mostFrequent) mostFrequent = cnt[s];
    const bool defaultAllowed = isOffsets ? (max <= ZS_DefaultMaxOff) : true;
    const int t = select_encoding(mostFrequent, nbSeq, defaultNormLog, defaultAllowed);
    *type = (uint32_t)t;
    if (t == 1) {                                                          // rle
        ct.tableLog = 0; ct.state[0] = 0; ct.state[1] = 0; ct.dnb[max] = 0; ct.dfs[max] = 0;
        *op = codes[0];
        return 1;
    }
    if (t == 0) {
        for (uint32_t s = 0; s <= defaultMax; s++) L.norm[s] = defaultNorm[s];
        fse_buildCTable(ct, L.norm, defaultMax, defaultNormLog, L.cumul, L.tableSymbol);
        return 0;
    }
    uint32_t nbSeq_1 = nbSeq;
    const uint32_t tableLog = fse_optimalTableLog(FSELog, nbSeq, max, 2);
    if (cnt[codes[nbSeq - 1]] > 1) { cnt[codes[nbSeq - 1]]--; nbSeq_1--; }
    if (fse_normalizeCount(L.norm, tableLog, cnt, nbSeq_1, max, nbSeq_1 >= 2048) < 0) return 0xFFFFFFFFu;
    const uint32_t sz = fse_writeNCount(op, L.norm, max, tableLog);
    fse_buildCTable(ct, L.norm, max, tableLog, L.cumul, L.tableSymbol);
    return sz;
}

__device__ ZS_NOINLINE static uint32_t compress_sequences(uint8_t* op0, uint8_t* oend, const zs_seq* __restrict__ seqs, uint32_t nbSeq, uint8_t* __restrict__ codes,
                                              EncLds& L, uint32_t* tmp, uint32_t tmpCap, uint32_t lane) {
    uint8_t* op = op0;
    uint8_t* const llC = codes; uint8_t* const ofC = codes + ZS_WS_CODE_STRIDE; uint8_t* const mlC = codes + 2 * ZS_WS_CODE_STRIDE;
    if (lane == 0) {
        if (nbSeq < 128) *op = (uint8_t)nbSeq;
        else if (nbSeq < 0x7F00) { op[0] = (uint8_t)((nbSeq >> 8) + 0x80); op[1] = (uint8_t)nbSeq; }
        else { op[0] = 0xFF; op[1] = (uint8_t)(nbSeq - 0x7F00); op[2] = (uint8_t)((nbSeq - 0x7F00) >> 8); }
    }
    op += nbSeq < 128 ? 1 : nbSeq < 0x7F00 ? 2 : 3;
    if (nbSeq == 0) return (uint32_t)(op - op0);
    // codes + the three histograms (all lanes); cnt layout: [0..35] LL, [64..95] OF, [128..180] ML inside hist2
    uint32_t* const cLL = L.hist2; uint32_t* const cOF = L.hist2 + 64; uint32_t* const cML = L.hist2 + 128;
    for (uint32_t i = lane; i < 192; i += LANES) L.hist2[i] = 0;
    __syncthreads();
    for (uint32_t u = lane; u < nbSeq; u += LANES) {
        const zs_seq q = seqs[u];
        const uint32_t a = LLcode(q.litLength), b = hb32(q.offBase), c = MLcode(q.mlBase);
        llC[u] = (uint8_t)a; ofC[u] = (uint8_t)b; mlC[u] = (uint8_t)c;
        atomicAdd(&cLL[a], 1u); atomicAdd(&cOF[b], 1u); atomicAdd(&cML[c], 1u);
    }
    __threadfence_block();
    __syncthreads();
    PT(8);
    if (lane == 0) {
        uint8_t* const seqHead = op; uint8_t* q = op + 1;
        uint32_t tLL, tOF, tML, lastCountSize = 0, fail = 0;
        uint32_t s1 = build_seq_table(q, L.ll, ZS_LLFSELog, cLL, ZS_MaxLL, llC, nbSeq, kLLdefaultNorm, 6, ZS_MaxLL, false, L, &tLL);
        if (s1 == 0xFFFFFFFFu) fail = 1; else { if (tLL == 2) lastCountSize = s1; q += s1; }
        uint32_t s2 = fail ? 0 : build_seq_table(q, L.of, ZS_OffFSELog, cOF, ZS_MaxOff, ofC, nbSeq, kOFdefaultNorm, 5, ZS_DefaultMaxOff, true, L, &tOF);
        if (s2 == 0xFFFFFFFFu) fail = 1; else if (!fail) { if (tOF == 2) lastCountSize = s2; q += s2; }
        uint32_t s3 = fail ? 0 : build_seq_table(q, L.ml, ZS_MLFSELog, cML, ZS_MaxML, mlC, nbSeq, kMLdefaultNorm, 6, ZS_MaxML, false, L, &tML);
        if (s3 == 0xFFFFFFFFu) fail = 1; else if (!fail) { if (tML == 2) lastCountSize = s3; q += s3; }
        // lane 0 hands over: where the bit stream starts, the count size rule, failure
        L.scal[3] = fail ? 0xFFFFFFFFu : (uint32_t)(q - op0);
        L.scal[4] = lastCountSize;
        if (!fail) *seqHead = (uint8_t)((tLL << 6) + (tOF << 4) + (tML << 2));
        PT(9);
    }
    __threadfence_block();
    __syncthreads();
    const uint32_t qoff = L.scal[3], lastCountSize = L.scal[4];
    __syncthreads();
    if (qoff == 0xFFFFFFFFu) return 0xFFFFFFFFu;
    // ---- ZSTD_encodeSequences, wave-parallel ----
    // (A) the three FSE state machines are independent chains: lanes 0 / 1 / 2 walk LL / OF / ML from the last sequence to
    //     the first and record, per sequence, the bits each transition emits (value | nbBits << 12).
    uint16_t* const stb = (uint16_t*)(codes + 3 * ZS_WS_CODE_STRIDE);
    uint32_t finalState = 0, finalLog = 0;
    if (lane < 3) {
        const FseTable& ct = lane == 0 ? L.ll : lane == 1 ? L.of : L.ml;
        const uint8_t* __restrict__ cd = codes + lane * ZS_WS_CODE_STRIDE;
        uint16_t* __restrict__ o = stb + lane * ZS_WS_CODE_STRIDE;
        uint32_t st = fse_init2(ct, cd[nbSeq - 1]);
        o[nbSeq - 1] = 0;
        if (nbSeq >= 2) {                                           // codes are read 8 at a time, one group ahead of their use
            int32_t n = (int32_t)nbSeq - 2;
            uint32_t g = (uint32_t)n & ~7u;
            uint64_t cur = *reinterpret_cast<const uint64_t*>(cd + g);
            for (;;) {
                const uint64_t nxt = g >= 8 ? *reinterpret_cast<const uint64_t*>(cd + g - 8) : 0;
                const int top = n & 7;
                // the per-symbol constants do not depend on the state: fetch all eight before walking the dependent chain,
                // which then costs one LDS lookup (the next state) per symbol
                uint32_t dn[8]; int32_t df[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const uint32_t sym = (uint32_t)(cur >> (8 * j)) & 0xFF; dn[j] = ct.dnb[sym < 56 ? sym : 0]; df[j] = ct.dfs[sym < 56 ? sym : 0]; }
                uint32_t ob[4] = {0, 0, 0, 0};
#pragma unroll
                for (int j = 7; j >= 0; j--) {
                    if (j <= top) {
                        const uint32_t nb = (st + dn[j]) >> 16;
                        ob[j >> 1] |= ((st & ((1u << nb) - 1)) | (nb << 12)) << (16 * (j & 1));
                        st = ct.state[(st >> nb) + df[j]];
                    }
                }
                if (top == 7) *reinterpret_cast<uint4*>(o + g) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
                else {
#pragma unroll
                    for (int j = 0; j < 8; j++) if (j <= top) o[g + j] = (uint16_t)(ob[j >> 1] >> (16 * (j & 1)));
                }
                if (g == 0) break;
                g -= 8; n = (int32_t)g + 7; cur = nxt;
            }
        }
        finalState = st & ((1u << ct.tableLog) - 1); finalLog = ct.tableLog;
    }
    __threadfence_block();
    __syncthreads();
    PT(10);
    // (B) every lane packs a contiguous run of sequences (in emission order: last sequence first)
    const uint32_t fLL = __shfl(finalState, 0), fOF = __shfl(finalState, 1), fML = __shfl(finalState, 2);
    const uint32_t gLL = __shfl(finalLog, 0), gOF = __shfl(finalLog, 1), gML = __shfl(finalLog, 2);
    const uint32_t per = (nbSeq + LANES - 1) / LANES;
    const uint32_t r0 = lane * per < nbSeq ? lane * per : nbSeq, r1 = r0 + per < nbSeq ? r0 + per : nbSeq;
    uint32_t bits = 0;
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t n = nbSeq - 1 - r;
        bits += (stb[n] >> 12) + (stb[ZS_WS_CODE_STRIDE + n] >> 12) + (stb[2 * ZS_WS_CODE_STRIDE + n] >> 12) + kLLbits[llC[n]] + kMLbits[mlC[n]] + ofC[n];
    }
    if (lane == LANES - 1) bits += gML + gOF + gLL + 1;             // final states + end mark
    const uint32_t startBit = wave_excl_scan(bits, lane);
    const uint32_t totalBits = __shfl(startBit + bits, LANES - 1);
    const uint32_t bitstreamSize = (totalBits + 7) / 8;
    uint8_t* const q = op0 + qoff;
    if (q + bitstreamSize > oend || bitstreamSize > tmpCap) return 0xFFFFFFFFu;       // does not fit: the block goes out raw
    {   const uint32_t words = (totalBits + 31) / 32 + 1;
        for (uint32_t wd = lane; wd < words; wd += LANES) tmp[wd] = 0;
    }
    __threadfence_block();
    __syncthreads();
    {   uint64_t acc = 0; uint32_t word = startBit >> 5, nacc = startBit & 31;
#define SEQ_PUT(v, nb) do { acc |= (uint64_t)(v) << nacc; nacc += (nb); if (nacc >= 32) { atomicOr(&tmp[word], (uint32_t)acc); acc >>= 32; nacc -= 32; word++; } } while (0)
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t n = nbSeq - 1 - r;
            const zs_seq sq = seqs[n];
            const uint32_t lc = llC[n], oc = ofC[n], mc = mlC[n];
            const uint32_t sLL = stb[n], sOF = stb[ZS_WS_CODE_STRIDE + n], sML = stb[2 * ZS_WS_CODE_STRIDE + n];
            const uint32_t nOF = sOF >> 12, nML = sML >> 12, nLL = sLL >> 12;
            const uint32_t v1 = (sOF & 0xFFF) | ((sML & 0xFFF) << nOF) | ((sLL & 0xFFF) << (nOF + nML));
            SEQ_PUT(v1, nOF + nML + nLL);
            const uint32_t bl = kLLbits[lc], bm = kMLbits[mc];
            const uint64_t v2 = (uint64_t)(sq.litLength & ((1u << bl) - 1)) | ((uint64_t)(sq.mlBase & ((1u << bm) - 1)) << bl);
            SEQ_PUT(v2, bl + bm);
            SEQ_PUT(sq.offBase & (uint32_t)((1ull << oc) - 1), oc);
        }
        if (lane == LANES - 1) {
            SEQ_PUT(fML, gML); SEQ_PUT(fOF, gOF); SEQ_PUT(fLL, gLL);
            SEQ_PUT(1u, 1u);
        }
#undef SEQ_PUT
        if (nacc) atomicOr(&tmp[word], (uint32_t)acc);
    }
    __threadfence_block();
    __syncthreads();
    wave_copy(q, (const uint8_t*)tmp, bitstreamSize, lane);
    __threadfence_block();
    __syncthreads();
    PT(10);
    if (lastCountSize && (lastCountSize + bitstreamSize) < 4) return 0xFFFFFFFFu;
    return qoff + bitstreamSize;
}

// ---------------------------------------------------------------------------------------------------
// libzstd 1.5.7 pre-block splitter (ZSTD_splitBlock_byChunks level 0): byte histogram of every 43rd byte of
// each 8 KiB chunk, compared with the accumulated past.
// ---------------------------------------------------------------------------------------------------
__device__ ZS_NOINLINE static uint32_t split_block_1_5_7(const uint8_t* __restrict__ p, EncLds& L, uint32_t lane) {
    uint32_t* past = L.hist; uint32_t* cur = L.hist2;
    uint32_t pastN = 0; int penalty = 3;
    for (uint32_t i = lane; i < 256; i += LANES) past[i] = 0;
    __syncthreads();
    for (uint32_t n = lane * 43; n < 8191; n += LANES * 43) atomicAdd(&past[p[n]], 1u);
    pastN = 8191 / 43;
    __syncthreads();
    for (uint32_t pos = 8192; pos <= ZS_BLOCK_MAX - 8192; pos += 8192) {
        for (uint32_t i = lane; i < 256; i += LANES) cur[i] = 0;
        __syncthreads();
        for (uint32_t n = lane * 43; n < 8191; n += LANES * 43) atomicAdd(&cur[p[pos + n]], 1u);
        const uint32_t curN = 8191 / 43;
        __syncthreads();
        uint64_t dev = 0;
        for (uint32_t i = lane; i < 256; i += LANES) {
            const int64_t d = (int64_t)past[i] * (int64_t)curN - (int64_t)cur[i] * (int64_t)pastN;
            dev += (uint64_t)(d < 0 ? -d : d);
        }
        for (int o = 32; o; o >>= 1) dev += __shfl_xor(dev, o);
        const uint64_t p50 = (uint64_t)pastN * (uint64_t)curN;
        const uint64_t threshold = p50 * (uint64_t)(14 + penalty) / 16;
        if (dev >= threshold) return pos;
        for (uint32_t i = lane; i < 256; i += LANES) past[i] += cur[i];
        pastN += curN;
        if (penalty > 0) penalty--;
        __syncthreads();
    }
    return ZS_BLOCK_MAX;
}

__device__ static bool wave_is_rle(const uint8_t* __restrict__ p, uint32_t n, uint32_t lane) {
    const uint8_t b0 = p[0];
    bool bad = false;
    for (uint32_t i = lane; i < n && !bad; i += LANES) bad = p[i] != b0;
    return !__any(bad);
}

// ---------------------------------------------------------------------------------------------------
// the kernel: one wave per chunk
// ---------------------------------------------------------------------------------------------------
// The frame is complete: publish its size and, when the batch is also encrypted, run GCM over it in this same wave.  A separate
// GCM launch would need 40 KiB of LDS per workgroup on CUs whose LDS and VGPRs are held by compressor waves of the batches in
// flight, and sat hundreds of ms in the queue for 20 ms of work; here it costs the wave a few ms of its second-long life.
__device__ static ZS_NOINLINE void finish_frame(tsx_chunk_desc* __restrict__ descs, uint32_t chunk, const uint8_t* frame, uint32_t flen,
                                    uint32_t* __restrict__ zlen, int32_t* __restrict__ status, const tsx_chain_fuse fuse, uint8_t* keyLocal, EncLds& L, uint32_t lane) {
    if (lane == 0) zlen[chunk] = flen;
    if (!fuse.key && !fuse.out) return;                                 // the frame stays in the staging buffer (stages as separate launches)
    __threadfence_block();
    __syncthreads();
    const uint64_t dstOff = descs[chunk].dst_off;
    if (!fuse.key) {
        // compression without encryption: the frame goes to the caller's slot as it is (16 bytes per lane: slots are 16-byte aligned)
        if (flen > descs[chunk].dst_cap) {
            if (lane == 0) { status[chunk] = TSX_E_DST_TOO_SMALL; descs[chunk].dst_len = 0; if (fuse.self_status) descs[chunk].status = TSX_E_DST_TOO_SMALL; }
            return;
        }
        uint8_t* const o = fuse.out + dstOff;
        const uint32_t q = flen >> 4;
        for (uint32_t i = lane; i < q; i += LANES) reinterpret_cast<uint4*>(o)[i] = reinterpret_cast<const uint4*>(frame)[i];
        for (uint32_t i = (q << 4) + lane; i < flen; i += LANES) o[i] = frame[i];
        if (lane == 0) { descs[chunk].dst_len = flen; if (fuse.self_status) descs[chunk].status = TSX_OK; }
        return;
    }
    if ((uint64_t)flen + 28 > descs[chunk].dst_cap) {
        if (lane == 0) { status[chunk] = TSX_E_DST_TOO_SMALL; descs[chunk].dst_len = 0; if (fuse.self_status) descs[chunk].status = TSX_E_DST_TOO_SMALL; }
        return;
    }
    const tsx_gcm_key* key = fuse.key;
    if (fuse.key_on_host) {
        // the key schedule waits in the caller's pinned memory: 21 KB over PCIe once per chunk, into this chunk's workspace
        static_assert(sizeof(tsx_gcm_key) <= ZS_WS_KEYCOPY_BYTES && sizeof(tsx_gcm_key) % 16 == 0, "key copy fits its workspace region");
        const uint4* s_ = reinterpret_cast<const uint4*>(fuse.key); uint4* d_ = reinterpret_cast<uint4*>(keyLocal);
        for (uint32_t i = lane; i < sizeof(tsx_gcm_key) / 16; i += LANES) d_[i] = s_[i];
        __threadfence_block();
        __syncthreads();
        key = reinterpret_cast<const tsx_gcm_key*>(keyLocal);
    }
    uint8_t iv[12];
    { const uint8_t* p_ = descs[chunk].iv; for (int i = 0; i < 12; i++) iv[i] = p_[i]; }
    gcm_encrypt_wave(fuse.aes, key, iv, frame, flen, fuse.out + dstOff, L.g.t0, L.g.tab, lane);
    if (fuse.key_on_host) {
        __threadfence_block();
        __syncthreads();
        uint4 z; z.x = z.y = z.z = z.w = 0;
        uint4* d_ = reinterpret_cast<uint4*>(keyLocal);
        for (uint32_t i = lane; i < sizeof(tsx_gcm_key) / 16; i += LANES) d_[i] = z;      // the copy does not outlive the chunk
    }
    if (lane == 0) { descs[chunk].dst_len = flen + 28; if (fuse.self_status) descs[chunk].status = TSX_OK; }
}

// A guest wave's look at the host's yield word (tsx_svc_host.yield, pinned memory: one PCIe read)
#ifdef HIPEMU
__device__ static inline uint32_t zs_yield_asked(const uint32_t* p) { return hipemu_yield_probe(p); }
#else
__device__ static inline uint32_t zs_yield_asked(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#endif

__device__ static inline uint32_t svc_cu_key();                        // (which compute unit is this wave on right now?  below, with the service)

// One chunk, start to finish, in the calling wave: CRC32C head, frame, GCM tail (or the copy into the caller's slot), descriptor.
// Every argument is the same in all lanes (the service kernel hands them over in SGPRs).
// `yield` != nullptr (a guest wave): looked at before every block; raised -> true is returned with the chunk unfinished.  Nothing the
// caller can see has been written by then but the chunk's CRC32C and TSX_OK in status[] (both the same again next time); hash tables,
// frame and entropy state live in the chunk's own workspace and are set up afresh by whoever starts the chunk again.
// `reserved` != nullptr (every other wave): the bitmap of reserved compute units - before every block the wave asks the hardware where it
// IS, and hands the chunk back the same way when that is a reserved CU.  A wave does not move by itself, but the hardware's scheduler may
// save a queue's waves and restore them later, anywhere (compute wave save / restore, when another queue's dispatch has waited long
// enough): after that, compressor waves sat on the reserved CUs for the rest of their launch and a fetch found no room - the "kernel of
// a fetch that does not start, once in a few hundred fetches" of round 5 (profiles/r06_stuck_fetch_trace.txt).
__device__ __forceinline__ static bool zstd_compress_chunk(EncLds& L, const uint8_t* __restrict__ src_base, tsx_chunk_desc* __restrict__ descs,
                                                           uint8_t* __restrict__ mid, uint64_t mid_stride, uint32_t* __restrict__ zlen,
                                                           int32_t* __restrict__ status, uint8_t* __restrict__ work, uint32_t profile, uint32_t sched,
                                                           const tsx_chain_fuse fuse, const uint32_t chunk, const uint32_t* yield, const uint32_t* reserved
#ifdef TSX_PROF
                                                           , unsigned long long* __restrict__ prof_out
#endif
                                                           ) {
    const uint32_t lane = threadIdx.x;
#ifdef TSX_PROF
    if (lane == 0) { for (int i = 0; i < 24; i++) g_prof[i] = 0; g_prof[22] = g_prof[23] = (unsigned long long)clock64(); g_prof[2] = wall_clock64(); }
    __syncthreads();
#endif
    const uint8_t* __restrict__ src = src_base + descs[chunk].src_off;
    const uint32_t srcSize = descs[chunk].src_len;
    uint8_t* const frame = mid + (uint64_t)chunk * mid_stride;
    uint8_t* const ws = work + (size_t)chunk * ZS_WS_BYTES;
    uint32_t* const hashLong = (uint32_t*)(ws + ZS_WS_HASHLONG);
    uint32_t* const hashSmall = (uint32_t*)(ws + ZS_WS_HASHSMALL);
    zs_seq* const seqs = (zs_seq*)(ws + ZS_WS_SEQS);
    uint8_t* const lit = ws + ZS_WS_LIT;
    uint8_t* const codes = ws + ZS_WS_CODES;
    uint8_t* const blockout = ws + ZS_WS_BLOCKOUT;
    uint32_t* const huftmp = (uint32_t*)(blockout + (256u << 10));                  // 4-byte aligned stream scratch
    if (fuse.crc) {
        const uint32_t crc = crc32c_wave(fuse.crc, src, srcSize, L.crcTab, lane);
        if (lane == 0) descs[chunk].crc32c = crc;
        PT(16);
    }
    if (fuse.self_status) { if (lane == 0) status[chunk] = TSX_OK; }    // (finish_frame publishes the chunk's final status in its descriptor)
    else if (status[chunk] != TSX_OK) { if (lane == 0) { zlen[chunk] = 0; if (fuse.key) descs[chunk].dst_len = 0; } return false; }

    const zs_cparams cp = zs_level3_cparams(srcSize);
    {   // fresh tables (ZSTD_reset_matchState): zero hashLong[1 << hashLog] and hashSmall[1 << chainLog]
        uint4 z; z.x = z.y = z.z = z.w = 0;
        uint4* a = (uint4*)hashLong; uint4* b = (uint4*)hashSmall;
        for (uint32_t i = lane; i < (1u << cp.hashLog) / 4; i += LANES) a[i] = z;
        for (uint32_t i = lane; i < (1u << cp.chainLog) / 4; i += LANES) b[i] = z;
    }
    // ---- frame header (ZSTD_writeFrameHeader: content size known, no checksum, no dictID) ----
    uint32_t hdr = 0;
    {   const uint32_t windowSize = 1u << cp.windowLog;
        const uint32_t single = windowSize >= srcSize;
        const uint32_t fcs = (srcSize >= 256) + (srcSize >= 65536 + 256);
        uint8_t h[16]; uint32_t k = 0;
        h[k++] = 0x28; h[k++] = 0xB5; h[k++] = 0x2F; h[k++] = 0xFD;
        h[k++] = (uint8_t)((single << 5) + (fcs << 6));
        if (!single) h[k++] = (uint8_t)((cp.windowLog - 10) << 3);
        if (fcs == 0) { if (single) h[k++] = (uint8_t)srcSize; }
        else if (fcs == 1) { h[k++] = (uint8_t)(srcSize - 256); h[k++] = (uint8_t)((srcSize - 256) >> 8); }
        else { h[k++] = (uint8_t)srcSize; h[k++] = (uint8_t)(srcSize >> 8); h[k++] = (uint8_t)(srcSize >> 16); h[k++] = (uint8_t)(srcSize >> 24); }
        hdr = k;
        if (lane == 0) for (uint32_t i = 0; i < k; i++) frame[i] = h[i];
    }
    uint8_t* op = frame + hdr;
    if (srcSize == 0) {
        if (lane == 0) { op[0] = 1; op[1] = 0; op[2] = 0; }
        finish_frame(descs, chunk, frame, hdr + 3, zlen, status, fuse, ws + ZS_WS_KEYCOPY, L, lane);
        return false;
    }
    uint32_t* const hufSave = (uint32_t*)(ws + ZS_WS_HUFSAVE);
    static_assert(sizeof(L.huf) % 4 == 0 && sizeof(L.huf) <= 2048, "Huffman tables fit their place in the workspace");
    if (lane == 0) { L.hufRepeat[0] = 0; L.hufRepeat[1] = 0; L.huf[0].maxSym = 0; L.huf[1].maxSym = 0; }
    __syncthreads();
    for (uint32_t i = lane; i < sizeof(L.huf) / 4; i += LANES) hufSave[i] = reinterpret_cast<const uint32_t*>(&L.huf[0])[i];
    __threadfence_block();
    __syncthreads();
    PT(0);
    uint32_t repc[3] = {1, 4, 8};                                       // confirmed repcode history
    const uint32_t blockSizeMax = (1u << cp.windowLog) < ZS_BLOCK_MAX ? (1u << cp.windowLog) : ZS_BLOCK_MAX;
    uint32_t ipos = 0, remaining = srcSize, dictLimit = 2;
    int64_t savings = 0;
    int cur = 0;                 // index of the confirmed Huffman table (L.huf[cur]); a candidate is built in L.huf[cur ^ 1]
    bool first = true;
    while (remaining) {
        if (yield || reserved) {                                        // a guest: is the CU wanted back?  anybody else: am I (still) off the reserved CUs -
            uint32_t y = 0;                                             // and if I am not (restored there by the hardware's scheduler): is the CU wanted?
            if (lane == 0) {
                if (reserved) { const uint32_t k = svc_cu_key(); y = (reserved[k >> 5] >> (k & 31)) & 1u; if (y && yield) y = zs_yield_asked(yield); }
                else y = zs_yield_asked(yield);
            }
            if (UNI(y)) return true;
        }
        // ---- block size (ZSTD_optimalBlockSize) ----
        uint32_t blockSize = remaining < blockSizeMax ? remaining : blockSizeMax;
        if (profile == TSX_ZSTD_PROFILE_1_5_7 && remaining >= ZS_BLOCK_MAX && blockSizeMax >= ZS_BLOCK_MAX && savings >= 3)
            blockSize = UNI(split_block_1_5_7(src + ipos, L, lane));
        PT(1);
        const uint32_t lastBlock = blockSize == remaining;
        {   // ZSTD_window_enforceMaxDist(&ms->window, ip, maxDist, ...): libzstd >= 1.5.0 slides the window to the block's start
            const uint32_t blockEndIdx = ipos + 2, maxDist = 1u << cp.windowLog;
            if (blockEndIdx > maxDist && dictLimit < blockEndIdx - maxDist) dictLimit = blockEndIdx - maxDist;
        }
        uint32_t cSize = 0;                                             // 0 -> raw block
        if (blockSize >= 7) {
            uint32_t rep[3] = {repc[0], repc[1], repc[2]};
            MfState ms;
            match_block(src, srcSize, ipos, blockSize, hashLong, hashSmall, cp, dictLimit, rep, seqs, ms, L.p.ring, L.p.scr, lane, sched);
            __threadfence_block();
            __syncthreads();
            gather_literals(lit, src, seqs, ms.nbSeq, ms.anchor, ms.lastLL, lane);
            __threadfence_block();
            __syncthreads();
            PT(18); PCNT(20, 1);
            // ---- ZSTD_entropyCompressSeqStore ----
            if (lane == 0) L.scal[8] = 0;
            __syncthreads();
            const bool suspect = ms.nbSeq == 0 || (ms.litSize / ms.nbSeq >= 20);
            for (uint32_t i = lane; i < sizeof(L.huf) / 4; i += LANES) reinterpret_cast<uint32_t*>(&L.huf[0])[i] = hufSave[i];     // back from the workspace
            __threadfence_block();
            __syncthreads();
            uint32_t litBytes = UNI(compress_literals(blockout, lit, ms.litSize, L, cur, suspect, huftmp, lane));
            __threadfence_block();
            __syncthreads();
            for (uint32_t i = lane; i < sizeof(L.huf) / 4; i += LANES) hufSave[i] = reinterpret_cast<const uint32_t*>(&L.huf[0])[i];     // the LL table takes their place
            __threadfence_block();
            __syncthreads();
            uint32_t seqBytes = UNI(compress_sequences(blockout + litBytes, blockout + (255u << 10), seqs, ms.nbSeq, codes, L, huftmp, (ZS_BLOCKOUT_CAP - (256u << 10)) - 64, lane));
            const bool newHuf = UNI(L.scal[8]) != 0;
            if (seqBytes != 0xFFFFFFFFu) {
                cSize = litBytes + seqBytes;
                const uint32_t maxCSize = blockSize - ((blockSize >> 6) + 2);
                if (cSize >= maxCSize) cSize = 0;
            }
            if (!first && ms.nbSeq < 4 && ms.litSize < 10 && wave_is_rle(src + ipos, blockSize, lane)) cSize = 1;
            if (cSize > 1) {                                            // confirm repcodes + entropy tables
                repc[0] = rep[0]; repc[1] = rep[1]; repc[2] = rep[2];
                if (newHuf) { cur ^= 1; if (lane == 0) L.hufRepeat[cur] = 1; }      // HUF_repeat_check for the next block
                __syncthreads();
            }
        }
        // ---- emit the block ----
        if (cSize == 0) {
            if (lane == 0) { const uint32_t h = lastBlock + (0u << 1) + (blockSize << 3); op[0] = (uint8_t)h; op[1] = (uint8_t)(h >> 8); op[2] = (uint8_t)(h >> 16); }
            wave_copy(op + 3, src + ipos, blockSize, lane);
            cSize = 3 + blockSize;
        } else if (cSize == 1) {
            if (lane == 0) { const uint32_t h = lastBlock + (1u << 1) + (blockSize << 3); op[0] = (uint8_t)h; op[1] = (uint8_t)(h >> 8); op[2] = (uint8_t)(h >> 16); op[3] = src[ipos]; }
            cSize = 4;
        } else {
            if (lane == 0) { const uint32_t h = lastBlock + (2u << 1) + (cSize << 3); op[0] = (uint8_t)h; op[1] = (uint8_t)(h >> 8); op[2] = (uint8_t)(h >> 16); }
            wave_copy(op + 3, blockout, cSize, lane);
            cSize += 3;
        }
        PT(11);
        savings += (int64_t)blockSize - (int64_t)cSize;
        ipos += blockSize; remaining -= blockSize; op += cSize; first = false;
        __syncthreads();
    }
    finish_frame(descs, chunk, frame, (uint32_t)(op - frame), zlen, status, fuse, ws + ZS_WS_KEYCOPY, L, lane);
    PT(17);
#ifdef TSX_PROF
    if (lane == 0 && prof_out) {
        g_prof[14] = (unsigned long long)clock64() - g_prof[22];
        g_prof[3] = wall_clock64();                                    // [2], [3]: the chunk's begin and end on the 100 MHz wall clock; [19]: where it ran (CU key | guest << 16)
        g_prof[19] = svc_cu_key() | (yield ? 1u << 16 : 0u);
        for (int i = 0; i < 24; i++) prof_out[(size_t)chunk * 24 + i] = g_prof[i];
    }
#endif
    return false;
}


#ifdef TSX_PROF
#define ZS_PROF_PARAM , unsigned long long* __restrict__ prof_out
#define ZS_PROF_ARG , prof_out
#else
#define ZS_PROF_PARAM
#define ZS_PROF_ARG
#endif

// ---------------------------------------------------------------------------------------------------
// the compressor service (tsx_internal.h: tsx_svc_host / tsx_svc_dev): persistent waves, one device-wide ticket queue
// ---------------------------------------------------------------------------------------------------
// Words the host (or another wave) rewrites while this kernel lives are never read through a cache: member slots and ticket
// records are reused, and a persistent wave gets no kernel-boundary invalidate.
#ifdef HIPEMU
#define SVC_LD_SYS(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define SVC_LD_DEV(p) __atomic_load_n((p), __ATOMIC_ACQUIRE)
#define SVC_ST_DEV(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define SVC_ST_SYS(p, v) __atomic_store_n((p), (v), __ATOMIC_RELEASE)
#define SVC_ST_MIRROR(p, v) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
__device__ static inline uint64_t svc_now() { return hipemu_clock_100mhz(); }
__device__ static inline uint32_t svc_cu_key() { return hipemu_cu_key(); }
__device__ static inline void svc_nap(uint32_t) {}
__device__ static inline void svc_acquire_chunk() {}
__device__ static inline void svc_release_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }   // (the harness's __threadfence_system is a wave rendezvous: lane 0 is alone here)
__device__ static inline void svc_fence_device() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#else
#define SVC_LD_SYS(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#define SVC_LD_DEV(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SVC_ST_DEV(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SVC_ST_SYS(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM)
#define SVC_ST_MIRROR(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)   /* statistics mirrors: no ordering wanted */
__device__ static inline uint64_t svc_now() { return wall_clock64(); }                       // 100 MHz, the same on every CU
// Which compute unit is this wave on?  HW_ID[15:8] = CU_ID | SH_ID | SE_ID, XCC_ID[3:0] = the XCD: a 12-bit key, unique per CU
// (tsx_launch_cu_probe counts the keys of a launch that covers the chip; the front end checks the count against the CU count).
__device__ static inline uint32_t svc_cu_key() {
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);              // hwreg(HW_REG_HW_ID, 0, 32)
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);             // hwreg(HW_REG_XCC_ID, 0, 4)
    return ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
}
__device__ static inline void svc_nap(uint32_t n) { for (uint32_t i = 0; i < n; i++) __builtin_amdgcn_s_sleep(127); }     // ~3.5 us each
// What this wave reads next (a source chunk the copy engine has just written, descriptors the host has just rewritten) must not come
// from this CU's vector L1 or the scalar cache: both survive from the wave's previous chunk.
__device__ static inline void svc_acquire_chunk() {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ static inline void svc_release_system() { __threadfence_system(); }
__device__ static inline void svc_fence_device() { __threadfence(); }
#endif
static_assert(sizeof(tsx_zseg) == 128 && offsetof(tsx_zseg, src_base) == 16 && offsetof(tsx_zseg, fuse) == 72 && offsetof(tsx_zseg, done) == 112,
              "zstd_service_kernel reads a member entry as 16 eight-byte words, one per lane");
static_assert(sizeof(tsx_chain_fuse) == 40 && offsetof(tsx_chain_fuse, self_status) == 32, "layout of the fuse words");

__device__ static inline uint64_t svc_word(uint64_t w, int k) {            // word k of the member entry (lane k holds it), in every lane
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, k), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), k);
    return ((uint64_t)hi << 32) | lo;
}

// Lane 0 of an idle wave: the next ticket (1), or leave (2).  A wave leaves when the device is told to stop, when its launch has
// reached its age limit, or when the queue has been dry AND no wave has held a ticket for idle_exit_ticks - as long as anyone is
// still compressing, the idle waves stay (napping): the next member finds the whole supply of waves, not the stragglers' kernel.
__device__ static inline void svc_ret_lock(tsx_svc_dev* D) { while (atomicCAS(&D->ret_lock, 0u, 1u) != 0u) svc_nap(1); svc_fence_device(); }
__device__ static inline void svc_ret_unlock(tsx_svc_dev* D) { svc_fence_device(); atomicExch(&D->ret_lock, 0u); }
// A guest hands its chunk back (lane 0, after the wave's last access to the chunk's workspace).
__device__ ZS_NOINLINE static void svc_return_chunk(tsx_svc_dev* D, uint32_t member_gen, uint32_t chunk) {
    svc_ret_lock(D);
    const uint32_t n = SVC_LD_DEV(&D->ret_n);
    if (n < TSX_SVC_RETURNED_MAX) { SVC_ST_DEV(&D->ret[n].member_gen, member_gen); SVC_ST_DEV(&D->ret[n].chunk, chunk); SVC_ST_DEV(&D->ret_n, n + 1u); }
    svc_ret_unlock(D);
    atomicAdd(&D->stat_returned, 1u);
}
// (3): a chunk that a guest handed back, in *ticket / *chunk_out (member slot | generation, chunk index) - taken before any fresh ticket.
// `yield` != nullptr: this wave is a guest and leaves (2) as soon as the word is raised.
// `moved`: a wave of the launch itself that sits on a reserved CU (restored there) - a guest for as long as nobody wants the CU.
__device__ ZS_NOINLINE static uint32_t svc_take(const tsx_svc_host* H, tsx_svc_dev* D, const tsx_svc_launch a, const uint64_t t_start, const uint32_t* yield,
                                                const uint32_t key, uint32_t* ticket, uint32_t* chunk_out, const bool moved) {
    const uint64_t max_age = ((uint64_t)a.max_age_ticks_hi << 32) | a.max_age_ticks_lo;
    uint64_t quiet_since = 0, dry_since = 0;
    uint32_t nap = 1, looks = 0, turned_away = 1;
    for (;;) {
        const uint64_t now = svc_now();
        if (SVC_LD_DEV(&D->stop)) return 2;
        if (max_age && now - t_start > max_age) return 2;
        if (yield && ((a.guests & 4u) || moved) && (looks++ & 7u) == 0u && zs_yield_asked(yield)) return 2;     // (an idle guest: one PCIe read per eight looks, <= 2 ms apart)
        if (SVC_LD_DEV(&D->ret_n)) {
            atomicAdd(&D->busy, 1u);
            svc_ret_lock(D);
            const uint32_t n = SVC_LD_DEV(&D->ret_n);
            if (n) { *ticket = SVC_LD_DEV(&D->ret[n - 1u].member_gen); *chunk_out = SVC_LD_DEV(&D->ret[n - 1u].chunk); SVC_ST_DEV(&D->ret_n, n - 1u); }
            svc_ret_unlock(D);
            if (n) { atomicAdd(&D->cu_busy[key], 1u); return 3; }
            atomicSub(&D->busy, 1u);
        }
        if (SVC_LD_DEV(&D->draining)) return 2;                          // the launch is ending (below): no more tickets for it - the host starts the next one
        // one wave per poll_ticks asks the host - whether or not the mirror is dry: the host's stop word (pause, rotation, shutdown) must not wait
        // for a queue of thousands of tickets to be consumed first
        const uint32_t ps = SVC_LD_DEV(&D->poll_stamp);
        if ((uint32_t)now - ps >= a.poll_ticks && atomicCAS(&D->poll_stamp, ps, (uint32_t)now) == ps) {
            const uint32_t p = SVC_LD_SYS(&H->published);
            if (SVC_LD_SYS(&H->stop)) { SVC_ST_DEV(&D->stop, 1u); return 2; }
            uint32_t old = SVC_LD_DEV(&D->pub);
            while ((int32_t)(p - old) > 0) { const uint32_t prev = atomicCAS(&D->pub, old, p); if (prev == old) { atomicAdd(&D->avail, p - old); break; } old = prev; }   // (as many rights as tickets)
        }
        const uint32_t nx = SVC_LD_DEV(&D->next), pb = SVC_LD_DEV(&D->pub);
        if ((int32_t)(pb - nx) > 0) {
            // Tickets are waiting.  Two things decide whether this wave gets one:
            // - its CU's share (tsx_svc_dev.cu_busy): a partial load is spread evenly over the compressor's CUs - no CU runs more chunks than the
            //   outstanding ones (queued + in progress) divided by the CUs, rounded up, plus one.  The slot on the CU is reserved first (an atomic on
            //   the CU's own word), so that the waves of one CU cannot all pass the check at once;
            // - the semaphore tsx_svc_dev.avail: a right to one ticket is an atomic decrement that found it positive, and only then is `next`
            //   advanced - by a fetch-add that cannot fail.  (Until round 6 the ticket was a compare-and-swap on `next`: with 5000 idle waves going
            //   for a fresh batch's 2048 tickets almost every attempt lost - its value of `next` was stale by the time the atomic was served - and
            //   the tickets left at 20 per millisecond: the last chunk of a lone batch began 105 ms after the first, profiles/r06_ticket_storm.txt.)
            // A wave that is turned away looks again after a nap that doubles (3.5 - 56 us).
            const uint32_t limit = a.spread_cus ? ((pb - SVC_LD_DEV(&D->fin)) + a.spread_cus - 1u) / a.spread_cus + 1u : 0xFFFFFFFFu;     // (published - finished = outstanding)
            // Look before the read-modify-write, both times: a wave that decrements a spent semaphore holds it one lower until it has put the right
            // back, and with thousands of idle waves doing that around the clock a SMALL member's rights (7 tickets against ~170 waves inside that
            // window at any moment) never showed as positive to anybody - the host tests' 7-chunk batches stood still on the device (the CPU
            // harness runs one workgroup at a time and cannot see it; tests/test_zzzz_gpu_service.py::test_small_members_next_to_thousands_of_idle_waves).
            // Waves that only LOAD a spent semaphore leave it alone: it shows its true value as soon as the last loser has put its right back.
            if (SVC_LD_DEV(&D->cu_busy[key]) < limit && (int32_t)SVC_LD_DEV(&D->avail) > 0) {
                if (atomicAdd(&D->cu_busy[key], 1u) < limit) {
                    if ((int32_t)atomicSub(&D->avail, 1u) > 0) {
                        atomicAdd(&D->busy, 1u);                         // before `next` moves: whoever finds the queue dry finds busy != 0 (the idle exit looks at both)
                        svc_fence_device();
                        *ticket = atomicAdd(&D->next, 1u);
                        return 1;
                    }
                    atomicAdd(&D->avail, 1u);
                }
                atomicSub(&D->cu_busy[key], 1u);
            }
            svc_nap(turned_away); if (turned_away < 16u) turned_away *= 2u;
            continue;
        }
        // A guest does not wait for work.  With EVERY wave slot of the chip held and most of the waves idle, the busy ones crawl: a lone
        // 2048-chunk batch took 1.1 - 9.8 s instead of 1.1 s, whether the idle waves were guests, waves kept on the reserved CUs or ordinary
        // waves of a launch without any reservation; with as little as a third of one CU per shader engine free it is 1.1 s every time
        // (profiles/r06_full_chip_with_idle_waves.txt).  A chip that is full AND busy is fine (that is the saturated regime guests exist for).
        // So a guest that has found nothing to do for guest_idle_ticks (10 ms: the gap between two rounds of callers that resubmit at once is 2 - 3 ms)
        // leaves its slot; the next launch - which begins when work arrives after a dry spell - has guests again.
        // While more than half of the launch they help is busy the chip is not "mostly idle": the queue of callers that resubmit as their batches
        // complete runs dry for milliseconds at a time, a guest that leaves then is not replaced before all guests have left (one guest launch at a
        // time), and a saturated run went on with 291 of 768 guests (profiles/r06_ticket_storm.txt 6).  Then a guest waits 50 times as long.
        if (yield) {
            if (dry_since == 0) dry_since = now;
            else if (now - dry_since >= a.guest_idle_ticks && (SVC_LD_DEV(&D->busy) * 2u < a.main_waves || now - dry_since >= 50ull * a.guest_idle_ticks)) return 2;
        }
        if (a.guest_launch) { svc_nap(nap); if (nap < 64) nap *= 2; continue; }      // (when the launch they help ends is not for its guests to say)
        if (SVC_LD_DEV(&D->busy) != 0 || quiet_since == 0) quiet_since = now;
        if (now - quiet_since >= a.idle_exit_ticks && SVC_LD_DEV(&D->busy) == 0) { atomicExch(&D->draining, 1u); return 2; }
        svc_nap(nap);
        if (nap < (a.idle_nap_max ? a.idle_nap_max : 64u)) nap *= 2;
    }
}

static_assert(sizeof(EncLds) <= 5 * 1280, "five 1280-byte LDS granules per chunk: 25 chunks fit a CU's 160 KiB, the registers allow 24");
// A wave leaves: the last one of the launch tells the host (pinned memory) that the launch is over, and when it began and ended.
__device__ ZS_NOINLINE static void svc_wave_exit(tsx_svc_host* H, tsx_svc_dev* D, uint32_t launch_id, uint32_t guest_launch) {
    SVC_ST_MIRROR(&H->m_live, atomicSub(&D->live, 1u) - 1u);
    if (guest_launch) {                                                  // a guest launch counts, and reports its end, apart
        if (atomicAdd(&D->g_exited, 1u) + 1u != gridDim.x) return;
        SVC_ST_DEV(&D->g_exited, 0u);
        svc_release_system();
        SVC_ST_SYS(&H->g_ended_launch, launch_id);
        return;
    }
    if (atomicAdd(&D->exited, 1u) + 1u != gridDim.x) return;
    // the last wave: the other statistics words as they stand (tsx_svc_host.m_*)
    SVC_ST_MIRROR(&H->m_live_max, SVC_LD_DEV(&D->live_max)); SVC_ST_MIRROR(&H->m_wave_starts, SVC_LD_DEV(&D->stat_wave_starts));
    SVC_ST_MIRROR(&H->m_reserved_exits, SVC_LD_DEV(&D->stat_reserved_exits)); SVC_ST_MIRROR(&H->m_skipped, SVC_LD_DEV(&D->stat_skipped));
    SVC_ST_MIRROR(&H->m_yields, SVC_LD_DEV(&D->stat_yields)); SVC_ST_MIRROR(&H->m_returned, SVC_LD_DEV(&D->stat_returned));
    SVC_ST_MIRROR(&H->m_chunks, SVC_LD_DEV(&D->stat_chunks));
    const uint64_t now = svc_now();
    const uint64_t first = ((uint64_t)SVC_LD_DEV(&D->t_first_hi) << 32) | SVC_LD_DEV(&D->t_first_lo);
    SVC_ST_DEV(&D->entered, 0u); SVC_ST_DEV(&D->exited, 0u);            // the next launch counts from zero (it is only started once this one is seen ended)
    SVC_ST_DEV(&D->draining, 0u);
    for (uint32_t g = 0; g < 256u; g++) if (SVC_LD_DEV(&D->kept[g])) SVC_ST_DEV(&D->kept[g], 0u);
#ifdef HIPEMU
    H->t_first = first; H->t_last = now;
#else
    __hip_atomic_store(&H->t_first, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&H->t_last, now, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    svc_release_system();
    SVC_ST_SYS(&H->ended_launch, launch_id);
}

__global__ __launch_bounds__(LANES, ZS_WAVES_PER_SIMD) void zstd_service_kernel(tsx_svc_host* H, tsx_svc_dev* D, const tsx_svc_launch a ZS_PROF_PARAM) {
    __shared__ EncLds L;
    const uint32_t lane = threadIdx.x;
    const uint64_t t_start = svc_now();
    if (lane == 0) {
        if (!a.guest_launch && atomicAdd(&D->entered, 1u) == 0u) { SVC_ST_DEV(&D->t_first_lo, (uint32_t)t_start); SVC_ST_DEV(&D->t_first_hi, (uint32_t)(t_start >> 32)); }
        const uint32_t lv = atomicAdd(&D->live, 1u) + 1u;
        atomicMax(&D->live_max, lv);
        SVC_ST_MIRROR(&H->m_live, lv);
    }
    if (a.calibrate_ticks) {                                              // how many of these workgroups does the chip hold at once?  (svc_create)
        // Every wave stays until no wave has ARRIVED for calibrate_ticks: live_max is then what fits at once, however slowly the dispatcher fills a
        // cold chip.  (Until round 6 a wave stayed a fixed 300 us from its own start: on a box whose first launch placed one wave per ~20 us and CU
        // the first waves had left before the sixteenth arrived - 15 per CU "measured", a grid of 3840 instead of 6144 for the life of the process.)
        if (lane == 0) {
            SVC_ST_DEV(&D->poll_stamp, (uint32_t)t_start);
            while ((uint32_t)svc_now() - SVC_LD_DEV(&D->poll_stamp) < a.calibrate_ticks && svc_now() - t_start < 5000000u) svc_nap(1);     // (<= 50 ms whatever happens)
            svc_wave_exit(H, D, a.launch_id, 0u);
        }
        return;
    }
    const uint32_t key = UNI(svc_cu_key());
    const uint32_t* yield = nullptr;                                     // != nullptr: this wave is a guest on a reserved CU
    const uint32_t* off_limits = D->reserved;                            // != nullptr: this wave leaves when it finds itself on a reserved CU (see zstd_compress_chunk)
    const bool on_reserved = ((D->reserved[key >> 5] >> (key & 31)) & 1u) != 0;
    if (a.guest_launch) {                                                // a launch of guests only: the reserved CUs are where it is meant to land
        uint32_t stay = 0;
        // (A guest that lands anywhere else leaves at once.  Slots that the launch itself had not filled swallow guest after guest that way - each is free
        //  again the moment its guest has left - and of 736 - 768 guests 288 - 767 stay, by run.  Letting them stay wherever they land was tried: every slot
        //  of the chip is held then, the hardware's scheduler saved and restored the waves in two runs of four, and those runs lost 10 % - for 224 waves
        //  more that the saturated regime, bound by its line requests, has no use for: profiles/r06_ticket_storm.txt 6.)
        if (lane == 0 && on_reserved && !zs_yield_asked(&H->yield)) stay = 2;
        if (!UNI(stay)) { if (lane == 0) svc_wave_exit(H, D, a.launch_id, 1u); return; }
        off_limits = nullptr; yield = &H->yield;
    } else if (on_reserved) {                                            // a reserved CU (see tsx_internal.h): the first keep_waves to arrive stay for good,
        uint32_t stay = 0;                                               // the others work as guests while no fetch is about (a.guests: the CPU harness; on
        if (lane == 0) {                                                 // the device guests come in launches of their own), or leave at once
            if (a.keep_waves && atomicAdd(&D->kept[key >> 4], 1u) < a.keep_waves) stay = 1;
            else if (a.guests && !zs_yield_asked(&H->yield)) stay = 2;
        }
        stay = UNI(stay);
        if (!stay) {
            if (lane == 0) { atomicAdd(&D->stat_reserved_exits, 1u); svc_wave_exit(H, D, a.launch_id, 0u); }
            return;
        }
        off_limits = nullptr;
        if (stay == 2) yield = &H->yield;
    }
    if (lane == 0) atomicAdd(&D->stat_wave_starts, 1u);
    uint32_t key_busy = 0;                                              // the CU this wave's chunk in progress is counted on (tsx_svc_dev.cu_busy)
    for (;;) {
        uint32_t got = 0, ticket = 0, chunk = 0;
        if (lane == 0) {
            // (between two chunks: where is this wave now?  Restored onto a reserved CU, it leaves before it takes another ticket - when the CU is
            //  wanted.  On a quiet device (yield word down) it works on like a guest: every restore used to cost the launch those waves for the rest of
            //  its life - a saturated run seen going from 6143 to 4153 live waves in 12 s, profiles/r06_ticket_storm.txt 6)
            const uint32_t k = svc_cu_key();
            const bool moved = off_limits && ((off_limits[k >> 5] >> (k & 31)) & 1u);
            if (moved && zs_yield_asked(&H->yield)) got = 2;
            else got = svc_take(H, D, a, t_start, moved ? &H->yield : yield, k, &ticket, &chunk, moved);
            if (moved && got == 2) SVC_ST_MIRROR(&H->m_relocated, atomicAdd(&D->stat_relocated, 1u) + 1u);
            if (got == 1 || got == 3) key_busy = k;                       // (svc_take has counted the chunk on this CU)
#ifdef TSX_PROF
            g_prof_take = wall_clock64();
#endif
        }
        key_busy = UNI(key_busy);
        got = UNI(got); ticket = UNI(ticket); chunk = UNI(chunk);
        if (got != 1 && got != 3) break;
        svc_acquire_chunk();
        // the ticket's record (a chunk that was handed back comes with its content) and its member's entry, straight from host memory
        uint32_t mg = ticket;
        if (got == 1) {
            if (lane == 0) { const tsx_svc_ticket* t = &H->ticket[ticket & (TSX_SVC_TICKETS - 1)]; mg = SVC_LD_SYS(&t->member_gen); chunk = SVC_LD_SYS(&t->chunk); }
            mg = UNI(mg); chunk = UNI(chunk);
        }
        const uint32_t slot = mg & 0xFFFFu;
        uint64_t w = 0;
        if (lane < 16 && slot < TSX_SVC_MEMBERS) w = SVC_LD_SYS(reinterpret_cast<const uint64_t*>(&H->member[slot]) + lane);
        const uint64_t w0 = svc_word(w, 0), w1 = svc_word(w, 1);
        const uint32_t n = (uint32_t)w0, profile = (uint32_t)(w0 >> 32), gen = (uint32_t)w1;
        if (slot >= TSX_SVC_MEMBERS || (gen & 0xFFFFu) != (mg >> 16) || chunk >= n) {          // an abandoned member's ticket
            if (lane == 0) { atomicAdd(&D->stat_skipped, 1u); atomicAdd(&D->fin, 1u); atomicSub(&D->cu_busy[key_busy], 1u); atomicSub(&D->busy, 1u); }
            continue;
        }
        tsx_chain_fuse fuse;
        fuse.crc = (const tsx_crc_tables*)svc_word(w, 9); fuse.aes = (const tsx_aes_tables*)svc_word(w, 10);
        fuse.key = (const tsx_gcm_key*)svc_word(w, 11); fuse.out = (uint8_t*)svc_word(w, 12);
        { const uint64_t f = svc_word(w, 13); fuse.self_status = (uint32_t)f; fuse.key_on_host = (uint32_t)(f >> 32); }
        uint32_t* const done = (uint32_t*)svc_word(w, 14); uint32_t* const flag = (uint32_t*)svc_word(w, 15);
        const bool handed_back = zstd_compress_chunk(L, (const uint8_t*)svc_word(w, 2), (tsx_chunk_desc*)svc_word(w, 3), (uint8_t*)svc_word(w, 4), svc_word(w, 5),
                            (uint32_t*)svc_word(w, 6), (int32_t*)svc_word(w, 7), (uint8_t*)svc_word(w, 8), profile, a.sched, fuse, chunk, off_limits ? &H->yield : (a.guests & 2u) ? yield : nullptr, off_limits ZS_PROF_ARG);
#ifdef TSX_PROF
        if (lane == 0 && prof_out && !handed_back) { prof_out[(size_t)chunk * 24 + 21] = t_start; prof_out[(size_t)chunk * 24 + 22] = g_prof_take; prof_out[(size_t)chunk * 24 + 19] |= (unsigned long long)key_busy << 20; }   // [21], [22]: when this wave began, when it had the ticket
#endif
        if (handed_back) {
            // a fetch has arrived: the chunk goes back to the queue - every lane's stores into its workspace are complete and released (as
            // at the end of a finished chunk: the next wave may sit on another XCD, behind another L2) before another wave can start it
            // again - and this wave leaves its CU to the fetch's kernels
            __syncthreads();
            if (lane == 0) {
                svc_release_system(); svc_return_chunk(D, mg, chunk);
                if (yield) SVC_ST_MIRROR(&H->m_yields, atomicAdd(&D->stat_yields, 1u) + 1u);
                else SVC_ST_MIRROR(&H->m_relocated, atomicAdd(&D->stat_relocated, 1u) + 1u);      // (not a guest: it was moved onto a reserved CU)
                atomicSub(&D->cu_busy[key_busy], 1u);
                atomicSub(&D->busy, 1u);
            }
            break;
        }
        // ---- this chunk is done: tell its member's caller when it was the member's last one ----
        // The kernel goes on, so nothing here may rely on an end-of-kernel release: every lane's stores (ciphertext in device memory, which
        // the caller's copy engine reads next, or in the caller's registered buffer; descriptor in pinned host memory) are complete at the
        // barrier, lane 0 releases them to system scope, and only then counts the chunk.
        __syncthreads();
        if (lane == 0) {
            svc_release_system();
            SVC_ST_MIRROR(&H->m_chunks, atomicAdd(&D->stat_chunks, 1u) + 1u);
            atomicAdd(&D->fin, 1u);
            if (atomicAdd(done, 1u) + 1u == n) {
                atomicExch(done, 0u);                                    // ready for the context's next member (ordered before it by the flag)
                svc_release_system();
                // a plain system-scope store, not an atomic read-modify-write: the flag lives in HOST memory, and an atomic there would need
                // PCIe AtomicOps routed all the way to the root complex - not every server does that
                SVC_ST_SYS(flag, 1u);
            }
            atomicSub(&D->cu_busy[key_busy], 1u);
            atomicSub(&D->busy, 1u);
        }
        __syncthreads();
    }
    if (lane == 0) svc_wave_exit(H, D, a.launch_id, a.guest_launch);
}

// A launch that covers the chip (48 KiB of LDS per one-wave workgroup: three per CU) and notes every CU key it meets.
__global__ __launch_bounds__(LANES) void cu_probe_kernel(tsx_svc_dev* D) {
    __shared__ uint32_t big[12288];
    big[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint32_t key = UNI(svc_cu_key());
    if (threadIdx.x == 0) atomicOr(&D->seen[key >> 5], 1u << (key & 31));
    const uint64_t t0 = svc_now();
    while (svc_now() - t0 < 3000u && big[(threadIdx.x * 7u) & 63u] != 0xFFFFFFFFu) svc_nap(1);   // ~30 us: later workgroups must go elsewhere
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
size_t tsx_zstd_consts_bytes(void) { return sizeof(tsx_zstd_consts); }
void tsx_zstd_build_consts(tsx_zstd_consts* h) { h->abi = 1; h->pad[0] = h->pad[1] = h->pad[2] = 0; }
size_t tsx_zstd_workspace_bytes(uint32_t n, uint32_t /*max_len*/) { return (size_t)n * ZS_WS_BYTES; }

void tsx_launch_zstd_service(hipStream_t st, tsx_svc_host* hd, tsx_svc_dev* d, uint32_t grid, tsx_svc_launch a) {
    if (!grid) return;
    hipLaunchKernelGGL(zstd_service_kernel, dim3(grid), dim3(LANES), 0, st, hd, d, a
#ifdef TSX_PROF
                       , g_prof_out
#endif
                       );
}
void tsx_launch_cu_probe(hipStream_t st, tsx_svc_dev* d, uint32_t grid) {
    if (grid) hipLaunchKernelGGL(cu_probe_kernel, dim3(grid), dim3(LANES), 0, st, d);
}
