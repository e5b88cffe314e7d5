/* Wave-step model of the GPU parser over the exact level-3 parse of a chunk (analysis tool, not product code).
 *
 * The serial restatement (oracle/zstd_l3.c, ORC_TRACE hooks on) yields the table-access trace of the parse: search runs, visited
 * positions, events, complementary insertions.  That trace is replayed through a model of csrc/zstd_enc.hip's match_block2:
 *   - a step probes K consecutive positions (schedule K0, K1, then doubling to 59), 2 table reads per position + 1 look-ahead;
 *   - an optional per-chunk SLOT CACHE in LDS (direct mapped, write back, exact): insertions of recurring content land there,
 *     a probe that hits needs no global read, and the first lane of a step that hits ends the step's speculation (it is almost
 *     surely the step's event), so the positions behind it are not probed at all.
 * Reported per sequence: global table reads / writes (line requests), dependent table round trips, far verifications.
 * usage: step_sim <file> [chunk_bytes] */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef struct { uint8_t k; uint32_t a, b, c; } trc_t;
static trc_t* g_trc; static size_t g_ntrc, g_captrc;
static void trc_push(int k, uint32_t a, uint32_t b, uint32_t c) {
    if (g_ntrc == g_captrc) { g_captrc = g_captrc ? g_captrc * 2 : (1u << 20); g_trc = realloc(g_trc, g_captrc * sizeof(trc_t)); }
    g_trc[g_ntrc].k = (uint8_t)k; g_trc[g_ntrc].a = a; g_trc[g_ntrc].b = b; g_trc[g_ntrc].c = c; g_ntrc++;
}
#define ORC_TRACE(kind, a, b, c) trc_push((kind), (a), (b), (c))
#include "../../oracle/zstd_l3.c"

/* ---- the slot cache: one array per table, key = slot number ---- */
typedef struct { uint32_t n; uint32_t* key; uint8_t* valid; uint8_t* dirty; uint32_t* pos; } cache_t;
static void cache_init(cache_t* c, uint32_t n) { c->n = n; c->key = calloc(n ? n : 1, 4); c->valid = calloc(n ? n : 1, 1); c->dirty = calloc(n ? n : 1, 1); c->pos = calloc(n ? n : 1, 4); }
static int cache_hit(const cache_t* c, uint32_t slot) { if (!c->n) return 0; uint32_t i = slot & (c->n - 1); return c->valid[i] && c->key[i] == slot; }
static uint32_t cache_pos(const cache_t* c, uint32_t slot) { return c->pos[slot & (c->n - 1)]; }
/* returns 1 when a dirty victim goes to global memory */
static int cache_put(cache_t* c, uint32_t slot, uint32_t pos) {
    uint32_t i = slot & (c->n - 1); int wb = c->valid[i] && c->key[i] != slot && c->dirty[i];
    c->valid[i] = 1; c->key[i] = slot; c->dirty[i] = 1; c->pos[i] = pos; return wb;
}
static void cache_update_if_present(cache_t* c, uint32_t slot, uint32_t pos) { if (cache_hit(c, slot)) { c->pos[slot & (c->n - 1)] = pos; c->dirty[slot & (c->n - 1)] = 0; } }

typedef struct {
    uint32_t k0, k1, nL, nS; int allocVisited;   /* allocVisited: 0 = only event positions + complementary insertions enter the cache */
    int cutOnHit;
    uint32_t fL, fS; int fAll;                   /* predictor-only filters (1 byte per entry, keyed by slot): entries, update policy */
    uint32_t gN, gD, gM;                         /* 4-gram recency filter over ALL bytes behind the step: entries, query at pos + gD, cut at firing lane + gM */
    uint64_t seqs, rdG, wrG, steps, freeSteps, farVerify, nearVerify, wasted, probes, cutFalse, evPredicted, events, wbacks;
} sim_t;

static const BYTE* g_src; static uint32_t g_ring = 4096;

static void run_sim(sim_t* S) {
    cache_t CL, CS; cache_init(&CL, S->nL); cache_init(&CS, S->nS);
    uint16_t* FL = calloc(S->fL ? S->fL : 1, 2); uint16_t* FS = calloc(S->fS ? S->fS : 1, 2);
#define FHIT(F, n, slot) ((n) && F[(slot) & ((n) - 1)] == (uint16_t)(((slot) / (n)) & 0xFF) + 1)
#define FPUT(F, n, slot) do { if (n) F[(slot) & ((n) - 1)] = (uint16_t)(((slot) / (n)) & 0xFF) + 1; } while (0)
    uint32_t off1 = 1, off2 = 4;
    uint8_t* G = calloc(S->gN ? S->gN : 1, 1); uint32_t gMarked = 0;
#define GH(q) (rd32(g_src + (q)) * 2654435761u)
    size_t i = 0;
    while (i < g_ntrc) {
        const trc_t* t = &g_trc[i];
        if (t->k == 'R') {
            /* collect the run: V's until E (or next R / end) */
            size_t j = i + 1, nv = 0; while (j < g_ntrc && g_trc[j].k == 'V') { j++; nv++; }
            int hasEv = j < g_ntrc && g_trc[j].k == 'E';
            const trc_t* V = &g_trc[i + 1];
            size_t c = 0; uint32_t width = S->k0;
            while (c < nv) {
                uint32_t K = width; if (K > nv - c && !hasEv) K = (uint32_t)(nv - c);
                /* lanes beyond the run's last visited position exist in the real parse only as speculation */
                uint32_t cut = K;                                     /* number of search lanes actually probed */
                int cutByCache = 0;
                if (S->gN) { const uint32_t p0 = V[c].a - 2; for (; gMarked < p0; gMarked++) { const uint32_t h = GH(gMarked); G[(h >> 8) & (S->gN - 1)] = (uint8_t)(h >> 24) | 1; } }
                for (uint32_t l = 0; l < K; l++) {
                    if (c + l >= nv) break;
                    const trc_t* v = &V[c + l];
                    const uint32_t p = v->a - 2;                      /* chunk offset (index - 2) */
                    int rep = off1 > 0 && p + 1 >= off1 && rd32(g_src + p + 1 - off1) == rd32(g_src + p + 1);
                    int hit = (S->cutOnHit && (cache_hit(&CL, v->b) || cache_hit(&CS, v->c))) || FHIT(FL, S->fL, v->b) || FHIT(FS, S->fS, v->c);
                    if (rep || hit) { cut = l + 1; cutByCache = !rep; break; }
                    if (S->gN) { const uint32_t h = GH(p + S->gD); if (G[(h >> 8) & (S->gN - 1)] == ((uint8_t)(h >> 24) | 1)) { if (l + 1 + S->gM < cut) { cut = l + 1 + S->gM; cutByCache = 1; } } }
                }
                uint32_t real = cut; if (c + real > nv) real = (uint32_t)(nv - c);            /* positions of the step that the serial parse visits */
                int evInStep = hasEv && c + cut >= nv;
                uint32_t reads = 0;
                for (uint32_t l = 0; l < cut; l++) {
                    if (c + l < nv) { const trc_t* v = &V[c + l]; reads += !cache_hit(&CL, v->b); reads += !cache_hit(&CS, v->c); }
                    else reads += 2;                                  /* speculation behind the event */
                }
                reads += 1;                                           /* look-ahead long probe (model: always global unless cached: unknown slot -> count it) */
                S->probes += 2 * cut + 1;
                if (c + cut > nv) S->wasted += 2 * (c + cut - nv);
                S->rdG += reads; S->steps++; if (reads <= 1) S->freeSteps++;
                if (cutByCache && !evInStep) S->cutFalse++;
                if (evInStep && cutByCache) S->evPredicted++;
                /* commit: visited positions insert themselves */
                for (uint32_t l = 0; l < real; l++) {
                    const trc_t* v = &V[c + l];
                    int isEv = hasEv && (c + l == nv - 1);
                    if (S->fAll || isEv) { FPUT(FL, S->fL, v->b); FPUT(FS, S->fS, v->c); }
                    if (S->nL && (S->allocVisited || isEv)) { S->wbacks += cache_put(&CL, v->b, v->a); } else { S->wrG++; cache_update_if_present(&CL, v->b, v->a); }
                    if (S->nS && (S->allocVisited || isEv)) { S->wbacks += cache_put(&CS, v->c, v->a); } else { S->wrG++; cache_update_if_present(&CS, v->c, v->a); }
                }
                c += real;
                if (evInStep) break;
                width = width < S->k1 ? S->k1 : (width * 2 > 59 ? 59 : width * 2);
            }
            i = j;
            continue;
        }
        if (t->k == 'E') {
            S->seqs++; S->events++;
            if (t->a != 1) { if (t->b > g_ring) S->farVerify++; else S->nearVerify++; off2 = off1; off1 = t->b; }
            i++; continue;
        }
        if (t->k == '1') { FPUT(FL, S->fL, t->b); if (S->nL) S->wbacks += cache_put(&CL, t->b, t->a); else S->wrG++; i++; continue; }
        if (t->k == 'C' || t->k == 'D') {
            uint32_t pL = t->k == 'C' ? t->a : t->a - 2, pS = t->k == 'C' ? t->a : t->a - 1;
            FPUT(FL, S->fL, t->b); FPUT(FS, S->fS, t->c);
            if (S->nL) S->wbacks += cache_put(&CL, t->b, pL); else S->wrG++;
            if (S->nS) S->wbacks += cache_put(&CS, t->c, pS); else S->wrG++;
            i++; continue;
        }
        if (t->k == 'I') {
            S->seqs++; { uint32_t x = off2; off2 = off1; off1 = x; }
            FPUT(FL, S->fL, t->b); FPUT(FS, S->fS, t->c);
            if (S->nL) S->wbacks += cache_put(&CL, t->b, t->a); else S->wrG++;
            if (S->nS) S->wbacks += cache_put(&CS, t->c, t->a); else S->wrG++;
            i++; continue;
        }
        i++;
    }
    S->wrG += S->wbacks;
}

int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    size_t n = argc > 2 ? (size_t)atol(argv[2]) : (size_t)1 << 30;
    BYTE* src = malloc(n + 64); n = fread(src, 1, n, f); fclose(f);
    size_t cap = orc_l3_compress_bound(n) + 64; BYTE* dst = malloc(cap);
    size_t csz = orc_l3_compress(src, n, dst, cap, 1);
    g_src = src;
    size_t nV = 0, nE = 0, nI = 0; for (size_t i = 0; i < g_ntrc; i++) { nV += g_trc[i].k == 'V'; nE += g_trc[i].k == 'E'; nI += g_trc[i].k == 'I'; }
    printf("src %zu -> %zu; trace %zu records: visited %zu events %zu imm-rep %zu (visited/seq %.2f)\n", n, csz, g_ntrc, nV, nE, nI, (double)nV / (nE + nI));
    printf("%-34s %7s %7s %7s %7s %7s %7s %7s %7s\n", "config", "rd/seq", "wr/seq", "ln/seq", "steps", "free", "farV", "waste", "cutFP");
    struct { const char* name; sim_t s; } cfg[] = {
        {"no cache (4,32)", {4, 32, 0, 0, 0, 0}},
        {"no cache (2,16)", {2, 16, 0, 0, 0, 0}},
        {"no cache (1,8)", {1, 8, 0, 0, 0, 0}},
        {"cache 128+128 ev-only (8,32)", {8, 32, 128, 128, 0, 1}},
        {"cache 256+256 ev-only (8,32)", {8, 32, 256, 256, 0, 1}},
        {"cache 512+512 ev-only (8,32)", {8, 32, 512, 512, 0, 1}},
        {"cache 1024+1024 ev-only (8,32)", {8, 32, 1024, 1024, 0, 1}},
        {"cache 256+256 all (8,32)", {8, 32, 256, 256, 1, 1}},
        {"cache 512+512 all (8,32)", {8, 32, 512, 512, 1, 1}},
        {"cache 1024+1024 all (8,32)", {8, 32, 1024, 1024, 1, 1}},
        {"cache 4096+4096 all (8,32)", {8, 32, 4096, 4096, 1, 1}},
        {"cache 512+512 ev-only (16,48)", {16, 48, 512, 512, 0, 1}},
        {"cache 512+512 ev-only (32,59)", {32, 59, 512, 512, 0, 1}},
        {"cache 512+512 ev-only nocut (4,32)", {4, 32, 512, 512, 0, 0}},
        {"gram 256 d4 m3 (3,59)", {3, 59, 0,0,0,0, 0,0,0, 256, 4, 3}},
        {"gram 512 d4 m3 (3,59)", {3, 59, 0,0,0,0, 0,0,0, 512, 4, 3}},
        {"gram 1K d4 m3 (3,59)", {3, 59, 0,0,0,0, 0,0,0, 1024, 4, 3}},
        {"gram 256 d4 m3 (4,32)", {4, 32, 0,0,0,0, 0,0,0, 256, 4, 3}},
        {"gram 512 d4 m3 (4,32)", {4, 32, 0,0,0,0, 0,0,0, 512, 4, 3}},
        {"gram 1K d4 m3 (4,32)", {4, 32, 0,0,0,0, 0,0,0, 1024, 4, 3}},
        {"gram 2K d4 m3 (2,59)", {2, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m3 (3,59)", {3, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m3 (4,59)", {4, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m3 (4,32)", {4, 32, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m2 (4,59)", {4, 59, 0,0,0,0, 0,0,0, 2048, 4, 2}},
        {"gram 2K d4 m2 (3,59)", {3, 59, 0,0,0,0, 0,0,0, 2048, 4, 2}},
        {"gram 2K d4 m1 (4,59)", {4, 59, 0,0,0,0, 0,0,0, 2048, 4, 1}},
        {"gram 2K d4 m3 (6,59)", {6, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m3 (8,59)", {8, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m2 (8,59)", {8, 59, 0,0,0,0, 0,0,0, 2048, 4, 2}},
        {"gram 2K d3 m2 (8,59)", {8, 59, 0,0,0,0, 0,0,0, 2048, 3, 2}},
        {"gram 2K d3 m1 (8,59)", {8, 59, 0,0,0,0, 0,0,0, 2048, 3, 1}},
        {"gram 2K d2 m1 (8,59)", {8, 59, 0,0,0,0, 0,0,0, 2048, 2, 1}},
        {"gram 2K d4 m3 (16,59)", {16, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 2K d4 m3 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 2048, 4, 3}},
        {"gram 1K d4 m3 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 1024, 4, 3}},
        {"gram 512 d4 m3 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 512, 4, 3}},
        {"gram 4K d4 m3 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 4096, 4, 3}},
        {"gram 2K d4 m2 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 2048, 4, 2}},
        {"gram 2K d3 m2 (59,59)", {59, 59, 0,0,0,0, 0,0,0, 2048, 3, 2}},
        {"filter 1K+1K ev (8,32)", {8, 32, 0, 0, 0, 0, 1024, 1024, 0}},
        {"filter 2K+2K ev (8,32)", {8, 32, 0, 0, 0, 0, 2048, 2048, 0}},
        {"filter 4K+4K ev (8,32)", {8, 32, 0, 0, 0, 0, 4096, 4096, 0}},
        {"filter 4K+4K ev (16,59)", {16, 59, 0, 0, 0, 0, 4096, 4096, 0}},
        {"filter 0+4K ev (8,32)", {8, 32, 0, 0, 0, 0, 0, 4096, 0}},
        {"filter 0+8K ev (8,32)", {8, 32, 0, 0, 0, 0, 0, 8192, 0}},
        {"filter 4K+4K all (8,32)", {8, 32, 0, 0, 0, 0, 4096, 4096, 1}},
        {"filter 16K+16K all (8,32)", {8, 32, 0, 0, 0, 0, 16384, 16384, 1}},
        {"filter 64K+64K all (8,32)", {8, 32, 0, 0, 0, 0, 65536, 65536, 1}},
        {"filter 64K+64K all (32,59)", {32, 59, 0, 0, 0, 0, 65536, 65536, 1}},
    };
    for (size_t k = 0; k < sizeof(cfg) / sizeof(cfg[0]); k++) {
        sim_t* S = &cfg[k].s; run_sim(S);
        double q = (double)S->seqs;
        printf("%-34s %7.2f %7.2f %7.2f %7.3f %7.3f %7.3f %7.2f %7.3f\n", cfg[k].name, S->rdG / q, S->wrG / q, (S->rdG + S->wrG) / q, S->steps / q, S->freeSteps / q,
               S->farVerify / q, S->wasted / q, S->cutFalse / q);
    }
    return 0;
}
