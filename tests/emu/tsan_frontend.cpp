// TEST HARNESS ONLY: the front end of libtsxform (csrc/tsx_api.hip: compressor service, context pools, device hints, copy pipeline) under
// ThreadSanitizer.  The kernel sources are compiled for the CPU emulator (tests/emu) with -fsanitize=thread and linked with this driver
// (`make -C csrc emu-tsan`): T threads issue context-less compressing batches (the broker's shape: members of the device's service queue), inverse
// batches, CRC-only batches and batches on explicit contexts at the same time; every result must equal the single-threaded one, and the
// tool must stay silent.  The emulator runs one grid at a time under a mutex, so what is examined is the library's own host code.
#include <tsxform.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include <vector>

static std::vector<uint8_t> make_chunk(uint32_t seed, uint32_t n) {
    std::vector<uint8_t> v(n);
    uint32_t x = seed * 2654435761u + 12345;
    static const char* words[] = {"{\"user\":", "\"offset\":", "\"topic\":\"orders\"", ",\"ts\":17", "\"partition\":", "null", "true", "kafka"};
    size_t p = 0;
    while (p < n) {
        x = x * 1664525u + 1013904223u;
        if ((x >> 28) < 11) { const char* w = words[(x >> 8) % 8]; for (size_t k = 0; w[k] && p < n; k++) v[p++] = (uint8_t)w[k]; }
        else v[p++] = (uint8_t)(x >> 16);
    }
    return v;
}

struct Job {
    std::vector<uint8_t> src; std::vector<tsx_chunk_desc> d; tsx_batch_params prm; size_t slot;
    std::vector<uint8_t> ref; std::vector<tsx_chunk_desc> dref;
};

static int run_forward(Job& j, tsx_ctx* ctx, int mem, std::vector<uint8_t>& out, std::vector<tsx_chunk_desc>& d) {
    d = j.d; out.assign(j.d.size() * j.slot, 0);
    return tsx_transform_batch(ctx, &j.prm, d.data(), (uint32_t)d.size(), j.src.data(), j.src.size(), out.data(), out.size(), mem);
}

int main(int argc, char** argv) {
    if (tsx_init(0, nullptr) <= 0) { fprintf(stderr, "tsx_init failed\n"); return 2; }
    const int T = argc > 1 ? atoi(argv[1]) : 8, REPS = argc > 2 ? atoi(argv[2]) : 2;
    const uint32_t scale = argc > 3 ? (uint32_t)atoi(argv[3]) : 8;        // chunk sizes are divided by this (the tool slows the emulated lanes ~50x)
    std::vector<Job> jobs(T);
    for (int t = 0; t < T; t++) {
        Job& j = jobs[t];
        const uint32_t sizes[5] = {(20000u + 977u * t) / scale, 1, 33000 / scale, 0, 4096 / scale};
        const uint32_t n = 2 + t % 3;                                   // (every chunk costs the emulated wave its table set-up: keep them few)
        uint64_t off = 0;
        memset(&j.prm, 0, sizeof j.prm);
        j.prm.flags = TSX_COMPRESS | TSX_ENCRYPT | TSX_CRC; j.prm.aad_len = 32; j.prm.zstd_profile = t & 1;
        for (int k = 0; k < 32; k++) { j.prm.key[k] = (uint8_t)(k * 7 + t); j.prm.aad[k] = (uint8_t)(k + 3 * t); }
        j.slot = (tsx_transformed_bound(40000, j.prm.flags) + 63) & ~(size_t)63;
        for (uint32_t i = 0; i < n; i++) {
            std::vector<uint8_t> c = make_chunk(100 * t + i, sizes[i]);
            tsx_chunk_desc d; memset(&d, 0, sizeof d);
            d.src_off = off; d.src_len = sizes[i]; d.dst_off = (uint64_t)i * j.slot; d.dst_cap = (uint32_t)j.slot;
            for (int k = 0; k < 12; k++) d.iv[k] = (uint8_t)(k + i + 16 * t);
            j.src.insert(j.src.end(), c.begin(), c.end()); off += sizes[i];
            while (off & 15) { j.src.push_back(0); off++; }              // slots are 16-byte aligned (tsx_api.hip validate)
            j.d.push_back(d);
        }
        if (j.src.empty()) j.src.push_back(0);
        const int rc0 = run_forward(j, nullptr, TSX_MEM_HOST, j.ref, j.dref);
        if (rc0 != TSX_OK) { fprintf(stderr, "reference run failed: %s\n", tsx_strerror(rc0)); return 2; }
        for (auto& d : j.dref) if (d.status != TSX_OK) { fprintf(stderr, "reference status %d\n", d.status); return 2; }
    }
    std::atomic<int> bad{0};
    static int racy = 0;                                                // TSAN_SELFTEST=1: a deliberate race, to see that the tool is awake in this build
    auto worker = [&](int t) {
        Job& j = jobs[t];
        if (getenv("TSAN_SELFTEST")) racy++;
        tsx_ctx* own = nullptr;
        if (t % 4 == 3 && tsx_ctx_create(0, 8, 40000, &own) != TSX_OK) { bad++; return; }
        for (int r = 0; r < REPS; r++) {
            if (t % 2) (void)tsx_set_thread_device(r % 2 ? -1 : 0);
            std::vector<uint8_t> out; std::vector<tsx_chunk_desc> d;
            if (run_forward(j, own, TSX_MEM_HOST, out, d) != TSX_OK || out != j.ref) { bad++; continue; }
            for (size_t i = 0; i < d.size(); i++) if (d[i].status != TSX_OK || d[i].dst_len != j.dref[i].dst_len || d[i].crc32c != j.dref[i].crc32c) bad++;
            // ... and back, through a pooled context of its own streams
            std::vector<tsx_chunk_desc> e(d.size());
            std::vector<uint8_t> back(j.src.size() + 64);
            for (size_t i = 0; i < d.size(); i++) { memset(&e[i], 0, sizeof e[i]); e[i].src_off = d[i].dst_off; e[i].src_len = d[i].dst_len; e[i].dst_off = j.d[i].src_off; e[i].dst_cap = j.d[i].src_len; }
            if (tsx_detransform_batch(nullptr, &j.prm, e.data(), (uint32_t)e.size(), out.data(), out.size(), back.data(), back.size(), TSX_MEM_HOST) != TSX_OK) { bad++; continue; }
            for (size_t i = 0; i < e.size(); i++) if (e[i].status != TSX_OK || e[i].crc32c != d[i].crc32c) bad++;
            for (size_t i = 0; i < e.size(); i++) if (e[i].dst_len != j.d[i].src_len || memcmp(back.data() + e[i].dst_off, j.src.data() + j.d[i].src_off, j.d[i].src_len)) bad++;
            std::vector<tsx_chunk_desc> c = j.d;
            if (tsx_crc32c_batch(nullptr, c.data(), (uint32_t)c.size(), j.src.data(), j.src.size(), TSX_MEM_HOST) != TSX_OK) { bad++; continue; }
            for (size_t i = 0; i < c.size(); i++) if (c[i].status != TSX_OK || c[i].crc32c != d[i].crc32c) bad++;
        }
        if (own) tsx_ctx_destroy(own);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto& x : th) x.join();
    uint32_t idle = 0, in_use = 0; uint64_t batches = 0;
    (void)tsx_pool_stats(0, &idle, &in_use, &batches);
    tsx_shutdown();
    if (bad || in_use) { fprintf(stderr, "tsan front end: %d mismatches, %u contexts still out\n", bad.load(), in_use); return 1; }
    printf("tsan front end ok: %d threads x %d rounds, %llu pooled batches\n", T, REPS, (unsigned long long)batches);
    return 0;
}
