/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Thin driver around the REFERENCE's own hot path.  It is compiled together
 * with /root/reference/src/alignment.c and alignment_scoring.c *where they lie*
 * (see oracle/Makefile); nothing of the reference is copied into this repo.
 * Output goes to oracle/_ref/libswref.so, which is git-ignored but travels to
 * the GPU box, where bench.py times it as the "reference" CPU baseline and the
 * tests use it to validate oracle/sw_oracle.c.
 *
 * What it reproduces of the reference driver (src/alignment_cmdline.c:459-509):
 * one aligner_t per 16-lane batch (aligner_create), then
 *   #pragma omp parallel for schedule(dynamic, 1)
 *   for (i < batch_cnt) alignment_fill_matrices(aligners[i]);
 * with only that loop timed, macro-batches of omp_get_max_threads()*512.
 */
#define _POSIX_C_SOURCE 200809L
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "alignment.h" /* reference header, -I/root/reference/src */

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void make_scoring(scoring_t *sc, const int8_t sub[32][32], int gap_open,
                         int gap_extend)
{
    memset(sc, 0, sizeof(*sc));
    scoring_init(sc, 2, -2, gap_open, gap_extend, false);
    memcpy(sc->swap_scores, sub, 32 * 32);
    for (int a = 0; a < 32; a++) sc->swap_set[a] = 0xFFFFFFFFu;
    sc->use_match_mismatch = 0;
}

int swref_letters_to_index(int c)
{
    /* guard: the reference exits the process on an illegal character */
    if (!((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '*')) return -1;
    return letters_to_index((char)c);
}

/* One reference-shaped batch: db_t is [max_len][16] int8, 32-byte aligned copy
 * is made here.  out[16] receives aligner->max_scores. */
void swref_fill_batch16(const int8_t *q_idx, size_t lq, const int8_t *db_t,
                        size_t max_len, const int8_t sub[32][32], int gap_open,
                        int gap_extend, int16_t *out)
{
    scoring_t sc;
    make_scoring(&sc, sub, gap_open, gap_extend);
    int8_t *q = NULL, *d = NULL;
    if (posix_memalign((void **)&q, 32, (lq + 31) / 32 * 32 + 32)) abort();
    if (posix_memalign((void **)&d, 32, max_len * 16 + 32)) abort();
    memcpy(q, q_idx, lq);
    memcpy(d, db_t, max_len * 16);
    aligner_t *al = aligner_create(NULL, NULL, NULL, NULL, q, d, lq, max_len, 16, &sc);
    alignment_fill_matrices(al);
    memcpy(out, al->max_scores, 16 * sizeof(int16_t));
    aligner_destroy(al);
    free(al->max_scores);
    free(al);
    free(q);
    free(d);
}

/* Many batches, the reference's dispatch.  db_t[b] -> [max_len[b]][16] int8
 * (must be 32-byte aligned), out -> [n_batches][16] int16.  Returns seconds
 * spent inside the parallel fill regions only (src/alignment_cmdline.c:503-509).
 * threads<=0 keeps the OpenMP default. */
double swref_fill_batches(const int8_t *q_idx, size_t lq,
                          const int8_t *const *db_t, const size_t *max_len,
                          size_t n_batches, const int8_t sub[32][32],
                          int gap_open, int gap_extend, int threads,
                          int16_t *out)
{
    scoring_t sc;
    make_scoring(&sc, sub, gap_open, gap_extend);
    if (threads > 0) omp_set_num_threads(threads);
    const size_t macro = (size_t)omp_get_max_threads() * 512; /* BATCH_SIZE_FACTOR */
    int8_t *q = NULL;
    if (posix_memalign((void **)&q, 32, (lq + 31) / 32 * 32 + 32)) abort();
    memcpy(q, q_idx, lq);
    aligner_t **als = (aligner_t **)calloc(macro, sizeof(aligner_t *));
    double total = 0.0;
    for (size_t base = 0; base < n_batches; base += macro) {
        size_t cnt = n_batches - base < macro ? n_batches - base : macro;
        for (size_t i = 0; i < cnt; i++) {
            if (als[i] == NULL)
                als[i] = aligner_create(NULL, NULL, NULL, NULL, q,
                                        (int8_t *)db_t[base + i], lq,
                                        max_len[base + i], 16, &sc);
            else
                aligner_update(als[i], NULL, NULL, NULL, NULL, q,
                               (int8_t *)db_t[base + i], lq, max_len[base + i],
                               16, &sc);
        }
        double t0 = now_s();
#pragma omp parallel for schedule(dynamic, 1)
        for (long i = 0; i < (long)cnt; i++) alignment_fill_matrices(als[i]);
        total += now_s() - t0;
        for (size_t i = 0; i < cnt; i++)
            memcpy(out + (base + i) * 16, als[i]->max_scores, 16 * sizeof(int16_t));
    }
    for (size_t i = 0; i < macro; i++)
        if (als[i]) {
            aligner_destroy(als[i]);
            free(als[i]->max_scores);
            free(als[i]);
        }
    free(als);
    free(q);
    return total;
}

int swref_max_threads(void) { return omp_get_max_threads(); }
