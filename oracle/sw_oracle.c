/*
 * sw_oracle.c -- TEST INFRASTRUCTURE ONLY (see sw_oracle.h).
 *
 * Scalar int32 restatement of the three-state local alignment fill of the
 * reference (src/alignment.c:47-187).  Written from the recurrence, not from
 * the reference's vector code: two rolling rows per state, one pair at a time.
 */
#include "sw_oracle.h"

#include <stdlib.h>
#include <string.h>

static inline int32_t max2(int32_t a, int32_t b) { return a > b ? a : b; }

/* Recurrence, reference src/alignment.c:124-161:
 *   s       = sub[q[i-1]][d[j-1]]
 *   H[j][i] = max(0, H[j-1][i-1]+s, A[j-1][i-1]+s, B[j-1][i-1]+s)
 *   A[j][i] = max(0, H[j-1][i]+go,  A[j-1][i]+ge,  B[j-1][i]+go)
 *   B[j][i] = max(0, H[j][i-1]+go,  A[j][i-1]+go,  B[j][i-1]+ge)
 *   go = gap_open + gap_extend, ge = gap_extend      (src/alignment.c:58-59)
 *   row 0 / column 0 of all three are 0               (src/alignment.c:77-97)
 *   score = max H                                     (src/alignment.c:133)
 */
int32_t sw_oracle_pair(const int8_t *q, size_t lq, const int8_t *d, size_t ld,
                       const int8_t sub[32][32], int gap_open, int gap_extend)
{
    if (lq == 0 || ld == 0) return 0;
    const int32_t go = gap_open + gap_extend, ge = gap_extend;
    int32_t *buf = (int32_t *)calloc(3 * (lq + 1), sizeof(int32_t));
    int32_t *H = buf, *A = buf + (lq + 1), *B = buf + 2 * (lq + 1);
    int32_t best = 0;
    for (size_t j = 1; j <= ld; j++) {
        const int dj = d[j - 1];
        int32_t h_diag = 0, a_diag = 0, b_diag = 0; /* [j-1][i-1] */
        int32_t h_left = 0, a_left = 0, b_left = 0; /* [j][i-1]   */
        for (size_t i = 1; i <= lq; i++) {
            const int32_t s = sub[(int)q[i - 1]][dj];
            const int32_t h_up = H[i], a_up = A[i], b_up = B[i];
            int32_t h = max2(max2(h_diag + s, a_diag + s), max2(b_diag + s, 0));
            int32_t a = max2(max2(h_up + go, a_up + ge), max2(b_up + go, 0));
            int32_t b = max2(max2(h_left + go, a_left + go), max2(b_left + ge, 0));
            if (h > best) best = h;
            H[i] = h; A[i] = a; B[i] = b;
            h_diag = h_up; a_diag = a_up; b_diag = b_up;
            h_left = h; a_left = a; b_left = b;
        }
    }
    free(buf);
    return best;
}

int32_t sw_oracle_pair_wrap16(const int8_t *q, size_t lq, const int8_t *d,
                              size_t ld, const int8_t sub[32][32], int gap_open,
                              int gap_extend)
{
    if (lq == 0 || ld == 0) return 0;
    const int16_t go = (int16_t)(gap_open + gap_extend), ge = (int16_t)gap_extend;
    int16_t *buf = (int16_t *)calloc(3 * (lq + 1), sizeof(int16_t));
    int16_t *H = buf, *A = buf + (lq + 1), *B = buf + 2 * (lq + 1);
    int16_t best = 0;
#define W16(x) ((int16_t)(uint16_t)(x))
#define MX(a, b) ((int16_t)((a) > (b) ? (a) : (b)))
    for (size_t j = 1; j <= ld; j++) {
        const int dj = d[j - 1];
        int16_t h_diag = 0, a_diag = 0, b_diag = 0, h_left = 0, a_left = 0, b_left = 0;
        for (size_t i = 1; i <= lq; i++) {
            const int16_t s = sub[(int)q[i - 1]][dj];
            const int16_t h_up = H[i], a_up = A[i], b_up = B[i];
            int16_t h = MX(MX(W16(h_diag + s), W16(a_diag + s)), MX(W16(b_diag + s), 0));
            int16_t a = MX(MX(W16(h_up + go), W16(a_up + ge)), MX(W16(b_up + go), 0));
            int16_t b = MX(MX(W16(h_left + go), W16(a_left + go)), MX(W16(b_left + ge), 0));
            if (h > best) best = h;
            H[i] = h; A[i] = a; B[i] = b;
            h_diag = h_up; a_diag = a_up; b_diag = b_up;
            h_left = h; a_left = a; b_left = b;
        }
    }
#undef W16
#undef MX
    free(buf);
    return best;
}

/* Alignment of one pair: the same recurrence kept whole, one byte per cell recording which
 * predecessor each of the three states took, then a walk back from the best match cell.
 * The reference prints scores only (its fork removed the traceback, Final Report p.7), so the
 * PATH has no reference output to be pinned by; its score is pinned like every other (the sum of
 * the path's substitution and gap scores equals sw_oracle_pair).  Tie rules are this build's
 * (include/swg.h, swg_align_hits): best cell = highest H, then smallest database position, then
 * smallest query position; a state whose maximum is 0 starts the alignment there; otherwise the
 * first maximal predecessor in the order H, A, B. */
int32_t sw_oracle_pair_trace(const int8_t *q, size_t lq, const int8_t *d, size_t ld,
                             const int8_t sub[32][32], int gap_open, int gap_extend,
                             uint32_t coords[4], char *ops, size_t ops_cap, size_t *n_ops)
{
    coords[0] = coords[1] = coords[2] = coords[3] = 0;
    *n_ops = 0;
    if (ops_cap) ops[0] = 0;
    if (lq == 0 || ld == 0) return 0;
    const int32_t go = gap_open + gap_extend, ge = gap_extend;
    int32_t *buf = (int32_t *)calloc(3 * (lq + 1), sizeof(int32_t));
    uint8_t *dir = (uint8_t *)malloc(lq * ld);
    int32_t *H = buf, *A = buf + (lq + 1), *B = buf + 2 * (lq + 1);
    int32_t best = 0;
    size_t bj = 0, bi = 0;
#define PICK(m, x, y, z) ((m) == 0 ? 0 : (x) == (m) ? 1 : (y) == (m) ? 2 : 3)
    for (size_t j = 1; j <= ld; j++) {
        const int dj = d[j - 1];
        int32_t h_diag = 0, a_diag = 0, b_diag = 0, h_left = 0, a_left = 0, b_left = 0;
        for (size_t i = 1; i <= lq; i++) {
            const int32_t s = sub[(int)q[i - 1]][dj];
            const int32_t h_up = H[i], a_up = A[i], b_up = B[i];
            const int32_t mh = max2(max2(h_diag, a_diag), max2(b_diag, 0));
            const int32_t ma = max2(max2(h_up + go, a_up + ge), max2(b_up + go, 0));
            const int32_t mb = max2(max2(h_left + go, a_left + go), max2(b_left + ge, 0));
            dir[(j - 1) * lq + (i - 1)] = (uint8_t)(PICK(mh, h_diag, a_diag, b_diag) |
                                                   PICK(ma, h_up + go, a_up + ge, b_up + go) << 2 |
                                                   PICK(mb, h_left + go, a_left + go, b_left + ge) << 4);
            const int32_t h = mh + s;
            if (h > best) { best = h; bj = j; bi = i; }
            H[i] = h; A[i] = ma; B[i] = mb;
            h_diag = h_up; a_diag = a_up; b_diag = b_up;
            h_left = h; a_left = ma; b_left = mb;
        }
    }
#undef PICK
    size_t n = 0, j = bj, i = bi;
    int state = 1;
    char *rev = (char *)malloc(lq + ld + 1);
    while (j > 0 && i > 0) {
        const uint8_t c = dir[(j - 1) * lq + (i - 1)];
        int p;
        if (state == 1) { rev[n++] = 'M'; p = c & 3; j--; i--; }
        else if (state == 2) { rev[n++] = 'I'; p = (c >> 2) & 3; j--; }
        else { rev[n++] = 'D'; p = (c >> 4) & 3; i--; }
        if (p == 0) break;
        state = p;
    }
    coords[0] = (uint32_t)i; coords[1] = (uint32_t)bi; coords[2] = (uint32_t)j; coords[3] = (uint32_t)bj;
    *n_ops = n;
    if (ops_cap > n) {
        for (size_t k = 0; k < n; k++) ops[k] = rev[n - 1 - k];
        ops[n] = 0;
    }
    free(rev); free(dir); free(buf);
    return best;
}

void sw_oracle_db(const int8_t *q, size_t lq, const int8_t *flat,
                  const uint64_t *offsets, size_t n, const int8_t sub[32][32],
                  int gap_open, int gap_extend, int32_t *scores)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (long k = 0; k < (long)n; k++) {
        scores[k] = sw_oracle_pair(q, lq, flat + offsets[k],
                                   (size_t)(offsets[k + 1] - offsets[k]), sub,
                                   gap_open, gap_extend);
    }
}

/* src/alignment_scoring.c:70-81 */
int sw_oracle_letter_index(int c)
{
    if (c >= 'a' && c <= 'z') return c - 'a' + 1;
    if (c >= 'A' && c <= 'Z') return c - 'A' + 1;
    if (c == '*') return 31;
    return -1;
}

typedef struct { int32_t score; uint32_t idx; } hit_t;

static int hit_cmp(const void *pa, const void *pb)
{
    const hit_t *a = (const hit_t *)pa, *b = (const hit_t *)pb;
    if (a->score != b->score) return a->score > b->score ? -1 : 1;
    if (a->idx != b->idx) return a->idx < b->idx ? -1 : 1;
    return 0;
}

size_t sw_oracle_topk(const int32_t *scores, size_t n, size_t k,
                      uint32_t *out_idx, int32_t *out_score)
{
    hit_t *h = (hit_t *)malloc((n ? n : 1) * sizeof(hit_t));
    for (size_t i = 0; i < n; i++) { h[i].score = scores[i]; h[i].idx = (uint32_t)i; }
    qsort(h, n, sizeof(hit_t), hit_cmp);
    size_t m = k < n ? k : n;
    for (size_t i = 0; i < m; i++) { out_idx[i] = h[i].idx; out_score[i] = h[i].score; }
    free(h);
    return m;
}
