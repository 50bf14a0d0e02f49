/*
 * swg_host.h -- plain-C host helpers that sit beside the hot path: the pieces of
 * the reference's driver a caller needs to feed swg.h.  None of them touches a
 * GPU.  Each cites the reference code whose behaviour it mirrors; the reference's
 * own FASTA and line readers (libs/seq_file, libs/string_buffer) are un-vendored
 * submodules that are absent from the reference tree, so parsing parity is
 * pinned only by SURVEY A.5/A.6, not by reference code.
 */
#ifndef SWG_HOST_H
#define SWG_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scoring (reference src/alignment_scoring.{h,c}) ------------------- */

typedef struct swg_scoring {
    int gap_open, gap_extend; /* scoring_t.gap_open/gap_extend */
    int match, mismatch;      /* parsed by the CLI, unused by the fill */
    int8_t sub[32][32];       /* scoring_t.swap_scores, zero-initialised here (A.7-1) */
    uint32_t set[32];         /* scoring_t.swap_set: bit b of set[a] <=> (a,b) defined */
} swg_scoring;

/* Residue -> table index, reference letters_to_index (src/alignment_scoring.c:70-81):
 * a-z/A-Z -> 1..26, '*' -> 31; returns -1 where the reference calls exit(1). */
int swg_letter_index(int c);
/* Inverse, reference index_to_letters (src/alignment_scoring.c:83-92); 0 if illegal. */
int swg_index_letter(int idx);

/* Defaults of the smith_waterman tool (src/tools/sw_cmdline.c:27-35):
 * match 2, mismatch -2, gap_open -2, gap_extend -1; empty table. */
void swg_scoring_init(swg_scoring *sc);
/* reference scoring_add_mutation (src/alignment_scoring.c:60-68); returns
 * SWG_ERR_ARG instead of asserting when score is outside -127..127 or a letter
 * is illegal. */
int swg_scoring_add(swg_scoring *sc, int a, int b, int score);
/* reference align_scoring_load_matrix (src/alignment_scoring_load.c:57-215),
 * file format in SURVEY A.5; gz-transparent like the reference's gzopen
 * (src/alignment_cmdline.c:230).  On error returns SWG_ERR_IO and a message
 * in err (the reference prints and exits). */
int swg_scoring_load_matrix(swg_scoring *sc, const char *path, char *err, size_t errlen);

/* Query sanitisation of the reference driver (src/alignment_cmdline.c:391-396):
 * a residue whose self-pair is undefined in the table becomes 'X'. */
void swg_query_sanitize(const swg_scoring *sc, int8_t *idx, size_t n);

/* ---- sequence files (reference: libs/seq_file via src/alignment_cmdline.c:335-457) */

typedef struct swg_seqs {
    size_t n;
    char *names;        /* NUL-separated header texts (without '>'/'@') */
    uint64_t *name_off; /* [n+1] */
    char *seq;          /* concatenated residue letters, no newlines */
    uint64_t *seq_off;  /* [n+1] */
} swg_seqs;

/* FASTA (multi-line), FASTQ, or one sequence per line; gzip-transparent;
 * "-" reads stdin.  max_records 0 = all. */
int swg_seqs_read(const char *path, size_t max_records, swg_seqs *out, char *err, size_t errlen);
void swg_seqs_free(swg_seqs *s);
/* letters -> indices for the whole set; on an illegal letter returns
 * SWG_ERR_RESIDUE and stores it in *bad (reference: message + exit(1)). */
int swg_seqs_to_indices(const swg_seqs *s, int8_t *out, char *bad);

/* ---- synthetic protein data (the reference ships no data) --- */

/* splitmix64 streams; residues i.i.d. over the 20 standard amino acids with
 * Swiss-Prot-like frequencies; lengths log-normal(median, sigma_ln) clamped to
 * [min_len, max_len], emitted sorted longest first.  Buffers are malloc'ed
 * here and released with swg_synth_free. */
int swg_synth_db(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                 uint32_t max_len, int8_t **flat_out, uint64_t **offsets_out);
void swg_synth_query(uint64_t seed, size_t lq, int8_t *out);
/* As swg_synth_db, but a seeded `fraction` of the sequences are full-length
 * copies of the query with `subst` point substitutions per residue: the
 * high-similarity set that drives int16 scores into saturation and exercises
 * the int32 re-score path. */
int swg_synth_db_similar(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                         uint32_t max_len, const int8_t *query, size_t lq, double fraction,
                         double subst, int8_t **flat_out, uint64_t **offsets_out, size_t *n_planted);
/* A family of relatives instead of near-copies: as swg_synth_db_similar, but every planted sequence draws
 * its own substitution rate from [subst_lo, subst_hi] (identity 1 - rate: 0.3 .. 0.7 gives scores between
 * the f16 cells' ceiling and int16's for a 3000-aa query).  subst_hi == subst_lo is swg_synth_db_similar. */
int swg_synth_db_family(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                        uint32_t max_len, const int8_t *query, size_t lq, double fraction,
                        double subst_lo, double subst_hi, int8_t **flat_out, uint64_t **offsets_out,
                        size_t *n_planted);
/* One shard of the same database without generating the rest: the sequences of the global bins
 * (128 consecutive sorted ranks) b with b % shard_count == shard_rank -- exactly the bins
 * swg_db_pack(..., shard_rank, shard_count) keeps of the whole database.  offsets_out[n_local+1]
 * are over the shard alone; index_out[n_local] are the sequences' global indices (a sequence is
 * seeded by its global sorted rank, so the union of all shards is byte for byte the database of
 * swg_synth_db / swg_synth_db_similar with the same arguments); residues_total = sum of lengths
 * of the WHOLE database.  fraction == 0: no planted copies (query may be NULL).  Feed the result
 * to swg_db_pack_shard. */
int swg_synth_db_shard(uint64_t seed, size_t n, double median, double sigma_ln, uint32_t min_len,
                       uint32_t max_len, const int8_t *query, size_t lq, double fraction, double subst,
                       int shard_rank, int shard_count, int8_t **flat_out, uint64_t **offsets_out,
                       uint32_t **index_out, size_t *n_local, uint64_t *residues_total, size_t *n_planted);
void swg_synth_free(void *p);

/* Threads the host helpers' parallel loops use: the smallest of OpenMP's maximum (OMP_NUM_THREADS),
 * the CPUs this process may run on, and its cgroup CPU quota.  (The reference sizes its loop by
 * omp_get_max_threads() alone, src/alignment_cmdline.c:341-347.) */
int swg_host_threads(void);

#ifdef __cplusplus
}
#endif
#endif /* SWG_HOST_H */
