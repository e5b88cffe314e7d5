// Internal declarations shared by the kernel translation units and the C-ABI front end.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/tsxform.h"

// 16 bytes that are only guaranteed 4-byte aligned (the ciphertext of IV(12)||C||TAG starts at +12).
struct __attribute__((packed, aligned(4))) tsx_u128a4 { uint32_t v[4]; };

// ---------------------------------------------------------------------------------------------------
// CRC32C
// ---------------------------------------------------------------------------------------------------
#define TSX_CRC_THREADS 256
#define TSX_CRC_ROW_BYTES (TSX_CRC_THREADS * 16)        /* 4 KiB: one 16-byte piece per thread per row   */
#define TSX_CRC_ROWS 64
#define TSX_CRC_SUB_BYTES (TSX_CRC_ROW_BYTES * TSX_CRC_ROWS) /* 256 KiB per workgroup                    */
#define TSX_CRC_SUB_PIECES (TSX_CRC_SUB_BYTES / 16)

struct tsx_crc_tables {          // device-resident constants, built once per device at tsx_init
    uint32_t slice[16][256];     // slice[d][b]: byte b followed by d zero bytes (slicing tables 0..15)
    uint32_t stride[4][256];     // stride[k][b]: state byte k advanced by one 4 KiB row
    uint32_t piece_pow[512];     // x^(128*d) mod P, d = 0..511 (reflected; x^0 = 0x80000000)
    uint32_t pow2[32];           // x^(128*2^k) mod P
};
void tsx_crc_build_tables(tsx_crc_tables* host_out);

// One launch pair: partial CRCs per 256 KiB sub-block, then per-chunk combine.  `partials` needs
// n * max_sub u32.  Input chunk i = src + descs[i].src_off (16-byte aligned), length descs[i].src_len;
// result in descs[i].crc32c (device copy of the descriptors) and, when hd_mirror != nullptr, in hd_mirror[i].crc32c as well (the
// batch's descriptors in pinned host memory, as the device addresses them: no copy back).
void tsx_launch_crc32c(hipStream_t st, const tsx_crc_tables* d_tab, const uint8_t* src, tsx_chunk_desc* d_descs,
                       uint32_t n, uint32_t max_len, uint32_t* d_partials, int use_dst_side, tsx_chunk_desc* hd_mirror = nullptr);

// ---------------------------------------------------------------------------------------------------
// AES-256-GCM
// ---------------------------------------------------------------------------------------------------
#define TSX_GCM_THREADS 256
#define TSX_GCM_SUB_BLOCKS 4096                          /* AES blocks per workgroup (64 KiB)            */
#define TSX_GCM_SUB_BYTES (TSX_GCM_SUB_BLOCKS * 16)

struct alignas(16) tsx_gf128 { uint64_t hi, lo; };   // GF(2^128) element, big-endian halves: bit 63 of hi = x^0

struct tsx_gcm_key {             // device-resident, derived per batch key by the setup kernel
    uint32_t rk[60];             // AES-256 round keys, little-endian words of the FIPS-197 byte schedule
    uint32_t aad_len;
    uint8_t  aad[64];
    uint8_t  pad_[12];
    tsx_gf128 h;                 // H = E_K(0^128)
    tsx_gf128 hstride_tab[32][16];  // 4-bit tables of H^256: [nibble position][value]
    tsx_gf128 hpow[512];         // H^d, d = 0..511 (d = 0 is the identity)
    tsx_gf128 hpow2[32];         // H^(2^k)
    tsx_gf128 h64_tab[64][4];    // 2-bit tables of H^64: [bit-pair position][value] (one-wave-per-message GCM, gcm_dev.h)
};

struct tsx_aes_tables {          // device-resident constants, built once per device at tsx_init
    uint32_t te0[256];           // T0[x] = {02.S(x), S(x), S(x), 03.S(x)} packed little-endian
};
void tsx_aes_build_tables(tsx_aes_tables* host_out);

struct tsx_chain_fuse {          // stages the compressor wave of chunk i runs itself, around the compression (zstd_compress_kernel)
    const tsx_crc_tables* crc;   // != nullptr: CRC32C of the source chunk -> descs[i].crc32c, before parsing it
    const tsx_aes_tables* aes;
    const tsx_gcm_key* key;      // != nullptr: GCM over the finished frame: IV||C||TAG -> out + descs[i].dst_off,
    uint8_t* out;                //             descs[i].dst_len = frame + 28
    uint32_t self_status;        // != 0: the wave owns its chunk's status - it starts from TSX_OK without reading status[] and publishes it in
    uint32_t key_on_host;        //       descs[i].status itself, so that the batch needs no init / publish kernels around the launch
                                 // key_on_host != 0: `key` points into pinned HOST memory - the wave copies it into its chunk's workspace
                                 // before the GCM tail and wipes that copy afterwards (no upload, no device-side wipe: nothing but the launch)
};

struct tsx_zseg {                // one caller's batch ("member") in the device's compressor service queue (zstd_service_kernel, tsx_api.hip)
    uint32_t n, profile;         // chunks 0 .. n - 1 of the batch; TSX_ZSTD_PROFILE_*
    uint32_t gen, pad;           // generation of this member slot: a ticket that names an older generation is skipped (abandoned member)
    const uint8_t* src_base; tsx_chunk_desc* descs; uint8_t* mid; uint64_t mid_stride; uint32_t* zlen; int32_t* status; uint8_t* work;
    tsx_chain_fuse fuse;
    // Per-member completion: a wave that has finished its chunk - frame, GCM tail, descriptor, all released to system scope - counts
    // itself in *done (device memory); the one that completes the member's n resets the counter and raises *flag (pinned host memory),
    // which is what the member's caller waits for.  A member does not wait for any other member.
    uint32_t* done; uint32_t* flag;
};

// ---- the compressor service: one device-wide work queue, persistent waves --------------------------------------------------------
// Every compressing batch of a device - whichever thread or context it comes from - is a MEMBER of that device's queue: the host
// appends one ticket per chunk (ticket -> member slot + chunk index) to a ring in pinned host memory and publishes the new end of the
// ring; the waves of zstd_service_kernel take tickets one by one (a device-side counter), compress (+ checksum + encrypt) the chunk and
// come back for the next.  One kernel per device is alive while there is work (it ends when the queue has been dry and no wave busy for
// a moment; the host starts the next one with the next member), so nothing about hardware queues, launch order or batch boundaries
// decides when a chunk starts: a freed wave takes the next chunk of whatever member is next.
// Software CU reservation (mixed load): a wave that finds itself on a RESERVED compute unit (s_getreg HW_ID / XCC_ID against a bitmap the
// host built from a probe launch at tsx_init) leaves before it takes a ticket.  Compressor waves hold every other wave slot and LDS
// byte of the chip for as long as uploads go on - the reserved CUs are where a fetch's decoder kernels (2-8 waves, up to 19.5 KB of LDS
// per workgroup) find room at once (ChunkCache.java:85-108 waits get.timeout.ms for them).
#define TSX_SVC_MEMBERS 2048u         /* member slots (a caller holds up to 8 until it has collected its pieces) */
#define TSX_SVC_TICKETS 65536u        /* ticket ring (power of two): chunks published and not yet completed never exceed it */
#define TSX_SVC_MEMBER_MAX 16384u     /* chunks per member (a larger batch goes as several members) */
#define TSX_SVC_RETURNED_MAX 4096u    /* chunks handed back by guest waves and not yet taken again: never more than waves on reserved CUs (<= 128 x 32) */
struct tsx_svc_ticket { uint32_t member_gen; uint32_t chunk; };   // member slot in the low 16 bits, the slot's generation (16 bits) above
struct tsx_svc_host {                // pinned host memory, written by the host, read by the device (through its device alias) ...
    uint32_t published;              // tickets [.., published) are valid; release-stored after their records and member entries
    uint32_t stop;                   // != 0: waves leave after their current chunk (shutdown / pause for memory management)
    uint32_t yield;                  // != 0: a fetch (any batch of ordinary kernels) is or has lately been on the device - GUEST waves, the ones that
                                     // work on a reserved CU while the device does nothing but compress, hand their chunk back and leave (see below)
    uint32_t pad_[13];
    // ... except this line, which the LAST wave of a launch writes: the launch is over.  The service's stream carries no HIP event: a marker
    // queued behind the kernel is a barrier packet that waits, at the head of its hardware queue, for as long as the kernel lives - and
    // while it waited, the first command of every stream whose hardware queue shares that queue's pipe did not start either (measured:
    // the first fetch after uploads began returned when the service kernel ended; profiles/r05_mixed_probe_trace_first_fetch_blocked.txt).
    uint32_t ended_launch;           // id of the last launch that has ended (release-stored after the two stamps)
    uint32_t ended_pad_;
    uint64_t t_first, t_last;        // 100 MHz clock at the launch's first wave start / last wave exit
    // ... and these, mirrors of the device's statistics words (tsx_svc_dev) that waves store here as they change them: tsx_service_stats reads
    // them while a launch is alive.  A copy out of device memory is a blit KERNEL for sizes this small (__amd_rocclr_copyBuffer, 512-thread
    // workgroups), and next to guest waves - every wave slot of the chip taken - it waited for the launch to end (profiles/r06_guest_waves_root_cause.md).
    // Plain stores, last writer wins: a value may lag a few chunks behind; the exact words are read once the kernel is gone.
    uint32_t m_chunks, m_live, m_live_max, m_wave_starts, m_reserved_exits, m_skipped, m_yields, m_returned;
    uint32_t m_relocated;
    uint32_t g_ended_launch;         // id of the last GUEST launch that has ended (see tsx_svc_launch.guest_launch)
    tsx_zseg member[TSX_SVC_MEMBERS];
    tsx_svc_ticket ticket[TSX_SVC_TICKETS];
};
struct tsx_svc_dev {                 // device memory: the waves' shared state
    uint32_t next;                   // next ticket to hand out (wrap-safe comparisons against pub)
    uint32_t pub;                    // device mirror of tsx_svc_host.published (an elected idle wave refreshes it)
    uint32_t busy;                   // waves that hold a ticket
    uint32_t poll_stamp;             // low 32 bits of the 100 MHz clock at the last host poll
    uint32_t stop;                   // mirror of tsx_svc_host.stop
    uint32_t gen_start_lo, gen_start_hi, gen_started;   // clock at the first wave of this launch (generation age)
    uint32_t stat_chunks, stat_wave_starts, stat_reserved_exits, stat_skipped;
    uint32_t entered, exited;        // waves of the current launch that have started / left (the last one to leave reports the launch's end)
    uint32_t t_first_lo, t_first_hi; // clock at the first wave's start
    uint32_t live, live_max;         // waves resident right now / the most ever (statistics: is the whole launch resident at once?)
    uint32_t draining;               // != 0: this launch is ending - set by the first wave whose idle timer fires, read by every wave before it takes a
                                     // ticket.  Leaving is a decision of the LAUNCH: with per-wave timers the exits spread over ~0.5 ms, a member published
                                     // inside that window was picked up by the waves that had not left yet, and the launch lived on with a fraction of its
                                     // waves (measured: ONE wave serving 10 240 queued chunks until the 60 s age limit, profiles/r06_guest_waves_root_cause.md)
    uint32_t g_exited;               // waves of the current guest launch that have left
    uint32_t fin;                    // tickets whose chunk is finished (or was skipped): pub - fin chunks are outstanding - queued or in progress
    uint32_t avail;                  // (signed) rights to a ticket: the poll adds what it adds to pub, a wave that decrements it from > 0 takes the next ticket
    uint32_t stat_relocated;         // waves that found themselves on a reserved CU they had not started on (saved and restored by the hardware's scheduler) and left
    uint32_t reserved[128];          // bitmap over CU keys (xcc_id << 8 | HW_ID[15:8]): 1 = reserved for everything but the compressor
    uint32_t seen[128];              // the probe launch's bitmap: CU keys that exist on this chip
    uint32_t kept[256];              // per shader engine (key >> 4): waves of the current launch that stayed on the engine's reserved CU (tsx_svc_launch.keep_waves)
    // chunks that guest waves handed back (the ticket's CONTENT: its ring record may be reused once its member has been abandoned); every
    // wave looks here before it takes a fresh ticket.  A spin lock (lane 0 only, a handful of instructions inside) - this happens once per
    // guest wave when a fetch arrives after a quiet time
    uint32_t ret_lock, ret_n, stat_yields, stat_returned;
    tsx_svc_ticket ret[TSX_SVC_RETURNED_MAX];
    // chunks in progress per compute unit (by CU key): a partial load is spread over the chip.  The hardware fills a launch's CUs one after the other,
    // and a chunk next to 20 others takes a third longer than one next to 8; so no CU takes more than its share of the outstanding chunks + 1 (svc_take).
    uint32_t cu_busy[4096];
};
struct tsx_svc_launch {              // kernel arguments that shape a launch
    uint32_t launch_id;              // what the last wave writes to tsx_svc_host.ended_launch
    uint32_t calibrate_ticks;        // != 0: every wave just stays this long and leaves (tsx_svc_dev.live_max then says how many fit at once)
    uint32_t sched;                  // parser speculation schedule (0 = default)
    uint32_t poll_ticks;             // 100 MHz ticks between two host polls (device-wide)
    uint32_t idle_exit_ticks;        // a wave leaves when the queue has been dry and no wave busy for this long
    uint32_t max_age_ticks_lo, max_age_ticks_hi;   // != 0: waves stop taking tickets when the launch is older (the host starts the next one)
    uint32_t guests;                 // != 0: the other waves on reserved CUs work too, as GUESTS - they look at tsx_svc_host.yield before every block of
                                     // their chunk (~30 ms apart) and while idle; once it is raised they hand the chunk back (another wave starts it
                                     // again from its first byte) and leave for good.  0: they leave at once (the reservation is in force from the start)
    uint32_t guest_launch;           // != 0: a launch of GUESTS ONLY, made next to a running launch whose queue is deeper than its waves: as many workgroups as the
                                     // reserved CUs hold; with every other slot of the chip taken that is where they land - a wave that finds itself anywhere else, or
                                     // finds the yield word raised, leaves at once.  Guests come and go with the load (they leave when the queue has been dry for
                                     // guest_idle_ticks); the launch they help stays as it is
    uint32_t spread_cus;             // != 0: compute units a partial load is spread over (the ones the compressor uses); 0 = tickets go to whoever asks first
    uint32_t guest_idle_ticks;       // a guest that has found the queue dry for this long leaves when the chip is mostly idle (a chip whose every slot is held by mostly IDLE waves slows the busy ones down)
    uint32_t main_waves;             // guest launches: the waves of the launch they help - while more than half of them are busy a guest waits out a dry queue (50 x guest_idle_ticks)
    uint32_t idle_nap_max;           // longest nap of an idle wave between two looks at the queue, in 3.5 us (0 = 64: 224 us)
    uint32_t keep_waves;             // waves that stay on a reserved CU all the same (0 = the CU is left alone; the rest of it - LDS, registers, wave slots - is the room a fetch's workgroups find)
};
void tsx_launch_zstd_service(hipStream_t st, tsx_svc_host* hd, tsx_svc_dev* d, uint32_t grid, tsx_svc_launch a);
void tsx_launch_cu_probe(hipStream_t st, tsx_svc_dev* d, uint32_t grid);

struct tsx_gcm_chunk {           // per-chunk work item (device)
    uint64_t in_off;             // plaintext (encrypt) / IV||C||TAG (decrypt) offset within `in`
    uint64_t out_off;            // IV||C||TAG (encrypt) / plaintext (decrypt) offset within `out`
    uint32_t len;                // plaintext length n
    int32_t  skip;               // != 0: chunk already failed, do nothing
    uint8_t  iv[12];
};

// The same key material computed on the host (no kernel: a 256-thread workgroup that wants 11 KiB of LDS can wait hundreds of
// milliseconds for a slot on a chip full of compressor waves); the caller uploads *out with a plain copy and wipes it afterwards.
void tsx_gcm_key_build_host(const uint8_t key32[32], const uint8_t* aad, uint32_t aad_len, tsx_gcm_key* out);
void tsx_launch_gcm_setup(hipStream_t st, const tsx_aes_tables* d_aes, const uint8_t* d_key32, const uint8_t* d_aad, uint32_t aad_len, tsx_gcm_key* d_key);
// encrypt: writes IV||C||TAG.  decrypt: verifies TAG (status[i] = TSX_E_TAG_MISMATCH on failure) and
// writes the plaintext.  d_partials: n * max_sub * 4 u32.
void tsx_launch_gcm(hipStream_t st, const tsx_aes_tables* d_aes, const tsx_gcm_key* d_key, const tsx_gcm_chunk* d_chunks, uint32_t n,
                    uint32_t max_len, const uint8_t* in, uint8_t* out, uint32_t* d_partials, int32_t* d_status,
                    int decrypt);
