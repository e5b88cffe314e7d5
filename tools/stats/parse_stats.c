/* Parse statistics of the level-3 double-fast parse on a chunk (test/analysis tool, not product code).
 * Includes the oracle's serial restatement with ORC_STAT hooks switched on.
 * usage: parse_stats <file> [chunk_bytes]      (reads chunk_bytes from the file, default whole file) */
#include <stdio.h>
#include <stdint.h>
static uint64_t st_run[65], st_ev[5], st_dist[33], st_visit, st_seq, st_immrep;
static uint64_t st_far[6];      /* winner farther than 4K,8K,16K,32K,64K,128K from ip */
static uint64_t st_cand_inwin_L, st_cand_inwin_S, st_cand_far4k_L, st_cand_far4k_S, st_cand_eq_L, st_cand_eq_S;
static uint32_t st_cur_run;
#define ORC_STAT_VISIT() do { st_visit++; st_cur_run++; } while (0)
#define ORC_STAT_EVENT(type, dist) do { st_ev[type]++; st_seq++; st_run[st_cur_run > 64 ? 64 : st_cur_run]++; st_cur_run = 0; \
    { uint32_t d_ = (uint32_t)(dist), b_ = 0; while (d_ >> b_ > 1) b_++; st_dist[b_]++; \
      for (int k_ = 0; k_ < 6; k_++) if ((uint32_t)(dist) > (4096u << k_)) st_far[k_]++; } } while (0)
#define ORC_STAT_IMMREP() do { st_immrep++; st_seq++; } while (0)
#define ORC_STAT_CAND(isLong, inwin, dist, eq) do { if (inwin) { if (isLong) { st_cand_inwin_L++; if ((dist) > 4096) st_cand_far4k_L++; if (eq) st_cand_eq_L++; } \
    else { st_cand_inwin_S++; if ((dist) > 4096) st_cand_far4k_S++; if (eq) st_cand_eq_S++; } } } while (0)
#include "../../oracle/zstd_l3.c"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    size_t n = argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 30;
    BYTE* src = malloc(n); n = fread(src, 1, n, f); fclose(f);
    size_t cap = orc_l3_compress_bound(n) + 64; BYTE* dst = malloc(cap);
    size_t c = orc_l3_compress(src, n, dst, cap, 1);
    printf("src %zu -> %zu (ratio %.3f)\n", n, c, (double)c / n);
    printf("sequences %lu (imm-rep %lu)  visited %lu  visited/seq %.2f  bytes/seq %.1f\n", st_seq, st_immrep, st_visit, (double)st_visit / st_seq, (double)n / st_seq);
    printf("events: rep %lu long %lu short %lu short->long+1 %lu\n", st_ev[1], st_ev[2], st_ev[3], st_ev[4]);
    printf("run length (positions visited incl. the event position) histogram:\n");
    uint64_t cum = 0, tot = 0; for (int i = 0; i <= 64; i++) tot += st_run[i];
    for (int i = 0; i <= 64; i++) { cum += st_run[i]; if (st_run[i]) printf("  %2d: %8lu  cum %.3f\n", i, st_run[i], (double)cum / tot); }
    printf("winner distance log2 histogram:\n");
    for (int i = 0; i < 33; i++) if (st_dist[i]) printf("  2^%d: %lu\n", i, st_dist[i]);
    for (int k = 0; k < 6; k++) printf("winner farther than %u: %.3f\n", 4096u << k, (double)st_far[k] / (st_seq - st_immrep));
    printf("probes with in-window entry: long %lu (far>4K %lu, bytes equal %lu)  short %lu (far>4K %lu, equal %lu)\n",
           st_cand_inwin_L, st_cand_far4k_L, st_cand_eq_L, st_cand_inwin_S, st_cand_far4k_S, st_cand_eq_S);
    return 0;
}
