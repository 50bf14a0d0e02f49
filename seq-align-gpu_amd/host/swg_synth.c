/*
 * swg_synth.c -- deterministic synthetic protein data.  The
 * reference ships neither data nor a generator (its database/ directory is
 * git-ignored); these definitions are this repo's and are frozen:
 *   - PRNG splitmix64; stream for sequence k = splitmix64 seeded with
 *     mix(seed, k), so generation is order-independent and parallel;
 *   - residues i.i.d. over the 20 standard amino acids, Swiss-Prot-like
 *     frequencies; no B/Z/X/'*'/J/O/U (the reference's table is undefined for
 *     J/O/U);
 *   - lengths log-normal(median, sigma_ln) clamped to [min_len, max_len],
 *     database emitted sorted longest first (the reference's precondition,
 *     src/alignment_cmdline.c:431-439).
 * Host-only C.
 */
#include "../../include/swg.h"
#include "../../include/swg_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint64_t mix(uint64_t seed, uint64_t k)
{
    uint64_t x = seed ^ (k * 0xD6E8FEB86659FD93ull);
    (void)splitmix64(&x);
    return splitmix64(&x);
}

/* amino acids by Swiss-Prot frequency (percent), SURVEY 8d */
static const char AA[20] = {'L', 'A', 'G', 'V', 'E', 'S', 'I', 'K', 'R', 'D',
                            'T', 'P', 'N', 'Q', 'F', 'Y', 'M', 'H', 'C', 'W'};
static const double FREQ[20] = {9.65, 8.25, 7.07, 6.86, 6.72, 6.65, 5.91, 5.80, 5.53, 5.46,
                                5.36, 4.74, 4.06, 3.93, 3.86, 2.92, 2.41, 2.27, 1.38, 1.10};

/* inverse CDF: 20 thresholds on a 16-bit uniform (the definition), and the same function tabulated
 * for all 65536 values (what the generator reads: one load per residue instead of a scan; a
 * 10M-sequence database is 3.8e9 residues) */
static uint16_t THR[20];
static int8_t IDX[20];
static int8_t PICK[65536];
static int tables_ready = 0;

static void init_tables(void)
{
#pragma omp critical(swg_synth_tables)
    if (!tables_ready) {
        double tot = 0, acc = 0;
        for (int i = 0; i < 20; i++) tot += FREQ[i];
        for (int i = 0; i < 20; i++) {
            acc += FREQ[i];
            double t = acc / tot * 65536.0;
            THR[i] = (uint16_t)(t >= 65535.0 ? 65535 : (uint32_t)t);
            IDX[i] = (int8_t)swg_letter_index(AA[i]);
        }
        THR[19] = 65535;
        for (uint32_t r = 0; r < 65536; r++) {
            int i = 0;
            while (r > THR[i]) i++; /* first threshold not below r */
            PICK[r] = IDX[i];
        }
        tables_ready = 1;
    }
}

static inline int8_t pick(uint32_t r16) { return PICK[r16 & 0xFFFF]; }

static void fill_random(uint64_t st, int8_t *dst, size_t len)
{
    size_t j = 0;
    while (j < len) {
        uint64_t r = splitmix64(&st);
        for (int q = 0; q < 4 && j < len; q++, r >>= 16) dst[j++] = pick((uint32_t)(r & 0xFFFF));
    }
}

void swg_synth_query(uint64_t seed, size_t lq, int8_t *out)
{
    init_tables();
    fill_random(mix(seed, 0x51u), out, lq);
}

typedef struct {
    uint32_t len;
    uint32_t planted;
} lenrec;

/* shard_count > 1: only the sequences of the global bins b (128 consecutive sorted ranks) with
 * b % shard_count == shard_rank are generated -- the same bins swg_db_pack(..., rank, count) keeps
 * of the whole database -- with offsets over the shard alone and index_out[i] = the sequence's
 * global index (= its sorted rank: the database is emitted sorted).  Every sequence is seeded by its
 * global rank, so the union of the shards is byte for byte the unsharded database. */
static int synth_impl(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                      uint32_t max_len, const int8_t *query, size_t lq, double fraction,
                      double subst, double subst_hi, int shard_rank, int shard_count, int8_t **flat_out,
                      uint64_t **offsets_out, uint32_t **index_out, size_t *n_local_out,
                      uint64_t *residues_total, size_t *n_planted)
{
    if (!flat_out || !offsets_out || min_len == 0 || max_len < min_len) return SWG_ERR_ARG;
    if (shard_count < 1 || shard_rank < 0 || shard_rank >= shard_count) return SWG_ERR_ARG;
    if (shard_count > 1 && (!index_out || !n_local_out)) return SWG_ERR_ARG;
    init_tables();
    *flat_out = NULL;
    *offsets_out = NULL;
    lenrec *rec = (lenrec *)malloc((n ? n : 1) * sizeof(lenrec));
    uint64_t *off = (uint64_t *)malloc((n + 1) * sizeof(uint64_t));
    if (!rec || !off) {
        free(rec);
        free(off);
        return SWG_ERR_NOMEM;
    }
    /* lengths: one stream, Box-Muller on 53-bit uniforms */
    uint64_t st = mix(seed, 0x1E46u);
    const double mu = log(median);
    const uint64_t plant_thr = fraction >= 1.0 ? UINT64_MAX : (uint64_t)(fraction * 18446744073709551615.0);
    size_t planted = 0;
    for (size_t i = 0; i < n; i++) {
        const double u1 = ((double)(splitmix64(&st) >> 11) + 1.0) / 9007199254740993.0;
        const double u2 = (double)(splitmix64(&st) >> 11) / 9007199254740992.0;
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double l = floor(exp(mu + sigma_ln * z) + 0.5);
        if (l < (double)min_len) l = (double)min_len;
        if (l > (double)max_len) l = (double)max_len;
        rec[i].len = (uint32_t)l;
        rec[i].planted = 0;
        if (query && fraction > 0.0 && mix(seed ^ 0x9147u, i) <= plant_thr) {
            rec[i].len = (uint32_t)lq;
            rec[i].planted = 1;
            planted++;
        }
    }
    /* stable counting sort, longest first */
    {
        uint32_t lmax = 0;
        for (size_t i = 0; i < n; i++)
            if (rec[i].len > lmax) lmax = rec[i].len;
        uint64_t *cnt = (uint64_t *)calloc((size_t)lmax + 2, sizeof(uint64_t));
        lenrec *srt = (lenrec *)malloc((n ? n : 1) * sizeof(lenrec));
        if (!cnt || !srt) {
            free(cnt);
            free(srt);
            free(rec);
            free(off);
            return SWG_ERR_NOMEM;
        }
        for (size_t i = 0; i < n; i++) cnt[lmax - rec[i].len + 1]++;
        for (size_t l = 1; l <= (size_t)lmax + 1; l++) cnt[l] += cnt[l - 1];
        for (size_t i = 0; i < n; i++) srt[cnt[lmax - rec[i].len]++] = rec[i];
        free(cnt);
        free(rec);
        rec = srt;
    }
    /* the sorted ranks this shard holds, in order */
    size_t n_local = 0;
    uint32_t *gidx = NULL;
    if (shard_count > 1) {
        for (size_t b = (size_t)shard_rank; b * 128 < n; b += (size_t)shard_count)
            n_local += (b * 128 + 128 <= n) ? 128 : n - b * 128;
        gidx = (uint32_t *)malloc((n_local ? n_local : 1) * sizeof(uint32_t));
        if (!gidx) {
            free(rec);
            free(off);
            return SWG_ERR_NOMEM;
        }
        size_t at = 0;
        for (size_t b = (size_t)shard_rank; b * 128 < n; b += (size_t)shard_count)
            for (size_t r = b * 128; r < b * 128 + 128 && r < n; r++) gidx[at++] = (uint32_t)r;
    } else {
        n_local = n;
    }
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += rec[i].len;
    off[0] = 0;
    for (size_t i = 0; i < n_local; i++) off[i + 1] = off[i] + rec[gidx ? gidx[i] : i].len;
    int8_t *flat = (int8_t *)malloc(off[n_local] ? off[n_local] : 1);
    if (!flat) {
        free(rec);
        free(off);
        free(gidx);
        return SWG_ERR_NOMEM;
    }
    /* subst_hi > subst: a family of relatives -- every planted sequence draws its own substitution rate from
     * [subst, subst_hi] (a stream of its own, so that subst_hi == subst is byte for byte the near-copy form) */
    const int ranged = subst_hi > subst;
#pragma omp parallel for schedule(dynamic, 256) num_threads(swg_host_threads())
    for (long long i = 0; i < (long long)n_local; i++) {
        const size_t k = gidx ? gidx[i] : (size_t)i; /* global sorted rank: the seed of the sequence */
        int8_t *dst = flat + off[i];
        const uint64_t s0 = mix(seed ^ 0x5EEDu, (uint64_t)k);
        if (!rec[k].planted) {
            fill_random(s0, dst, rec[k].len);
        } else {
            /* copy of the query with point substitutions */
            uint64_t s1 = s0;
            double rate = subst;
            if (ranged)
                rate += (subst_hi - subst) * ((double)(mix(seed ^ 0xFA41u, (uint64_t)k) >> 11) / 9007199254740992.0);
            const uint64_t sub_thr = (uint64_t)(rate * 65536.0);
            for (size_t j = 0; j < lq; j++) {
                const uint64_t r = splitmix64(&s1);
                dst[j] = ((r & 0xFFFF) < sub_thr) ? pick((uint32_t)((r >> 16) & 0xFFFF)) : query[j];
            }
        }
    }
    free(rec);
    *flat_out = flat;
    *offsets_out = off;
    if (index_out) *index_out = gidx; else free(gidx);
    if (n_local_out) *n_local_out = n_local;
    if (residues_total) *residues_total = total;
    if (n_planted) *n_planted = planted;
    return SWG_OK;
}

int swg_synth_db(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                 uint32_t max_len, int8_t **flat_out, uint64_t **offsets_out)
{
    return synth_impl(seed, n, median, sigma_ln, min_len, max_len, NULL, 0, 0.0, 0.0, 0.0, 0, 1, flat_out,
                      offsets_out, NULL, NULL, NULL, NULL);
}

int swg_synth_db_similar(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                         uint32_t max_len, const int8_t *query, size_t lq, double fraction,
                         double subst, int8_t **flat_out, uint64_t **offsets_out, size_t *n_planted)
{
    if (!query || lq == 0) return SWG_ERR_ARG;
    return synth_impl(seed, n, median, sigma_ln, min_len, max_len, query, lq, fraction, subst, subst, 0, 1,
                      flat_out, offsets_out, NULL, NULL, NULL, n_planted);
}

int swg_synth_db_family(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                        uint32_t max_len, const int8_t *query, size_t lq, double fraction,
                        double subst_lo, double subst_hi, int8_t **flat_out, uint64_t **offsets_out,
                        size_t *n_planted)
{
    if (!query || lq == 0 || subst_hi < subst_lo || subst_lo < 0.0 || subst_hi > 1.0) return SWG_ERR_ARG;
    return synth_impl(seed, n, median, sigma_ln, min_len, max_len, query, lq, fraction, subst_lo, subst_hi, 0, 1,
                      flat_out, offsets_out, NULL, NULL, NULL, n_planted);
}

int swg_synth_db_shard(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                       uint32_t max_len, const int8_t *query, size_t lq, double fraction, double subst,
                       int shard_rank, int shard_count, int8_t **flat_out, uint64_t **offsets_out,
                       uint32_t **index_out, size_t *n_local, uint64_t *residues_total, size_t *n_planted)
{
    if (!index_out || !n_local) return SWG_ERR_ARG;
    if (fraction > 0.0 && (!query || lq == 0)) return SWG_ERR_ARG;
    int rc = synth_impl(seed, n, median, sigma_ln, min_len, max_len, fraction > 0.0 ? query : NULL, lq, fraction,
                        subst, subst, shard_rank, shard_count, flat_out, offsets_out, index_out, n_local,
                        residues_total, n_planted);
    if (rc == SWG_OK && shard_count == 1 && !*index_out) {
        /* one shard: the identity map, so that callers have one code path */
        uint32_t *g = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
        if (!g) return SWG_ERR_NOMEM;
        for (size_t i = 0; i < n; i++) g[i] = (uint32_t)i;
        *index_out = g;
    }
    return rc;
}

void swg_synth_free(void *p) { free(p); }
