/*
 * sw_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, scalar, int32) of the affine-gap Smith-Waterman
 * fill that the reference implements in src/alignment.c:47-187.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product library (seq-align-gpu_amd/csrc) never links or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 * (a) the committed golden vectors under tests/golden/, which were produced by
 * the reference's own alignment.c + alignment_scoring.c compiled unmodified
 * (oracle/Makefile -> oracle/_ref/libswref.so, generator tests/golden/make_golden.py)
 * and (b) that library directly whenever it has been built.
 */
#ifndef SW_ORACLE_H
#define SW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One query/database pair, truth domain int32 (SURVEY A.1 note 5).
 * q[lq], d[ld] are substitution-table indices (reference letters_to_index,
 * src/alignment_scoring.c:70-81); sub is the 32x32 int8 table indexed
 * [query][db] (src/alignment.c:33,41); gap_open/gap_extend as in scoring_t.
 * Returns max over all cells of the match matrix (src/alignment.c:133). */
int32_t sw_oracle_pair(const int8_t *q, size_t lq, const int8_t *d, size_t ld,
                       const int8_t sub[32][32], int gap_open, int gap_extend);

/* The pair's alignment: score as sw_oracle_pair; coords = {query begin, query end, database begin,
 * database end} (0-based, half-open); ops = 'M' residue pair, 'I' database residue against a gap
 * (state A), 'D' query residue against a gap (state B), first to last, NUL-terminated when
 * ops_cap > *n_ops.  The reference has no traceback: the path is this build's definition
 * (tie rules in sw_oracle.c), only its score is reference-pinned. */
int32_t sw_oracle_pair_trace(const int8_t *q, size_t lq, const int8_t *d, size_t ld,
                             const int8_t sub[32][32], int gap_open, int gap_extend,
                             uint32_t coords[4], char *ops, size_t ops_cap, size_t *n_ops);

/* Whole database: flat residue indices + offsets[n+1]; OpenMP over pairs.
 * scores[n] in database order. */
void sw_oracle_db(const int8_t *q, size_t lq, const int8_t *flat,
                  const uint64_t *offsets, size_t n, const int8_t sub[32][32],
                  int gap_open, int gap_extend, int32_t *scores);

/* The reference's int16 lanes wrap (src/alignment.c:124-161 uses
 * _mm256_add_epi16).  Same recurrence evaluated in wrapping int16 so tests can
 * show where the reference itself stops being the truth. */
int32_t sw_oracle_pair_wrap16(const int8_t *q, size_t lq, const int8_t *d,
                              size_t ld, const int8_t sub[32][32], int gap_open,
                              int gap_extend);

/* Reference residue map, src/alignment_scoring.c:70-81.  Returns -1 instead of
 * calling exit(1) for an illegal character. */
int sw_oracle_letter_index(int c);

/* Global top-K with the build's tie rule (higher score first, then lower
 * database index).  out_idx/out_score have k entries; returns entries written. */
size_t sw_oracle_topk(const int32_t *scores, size_t n, size_t k,
                      uint32_t *out_idx, int32_t *out_score);

#ifdef __cplusplus
}
#endif
#endif
