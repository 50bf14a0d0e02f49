/*
 * swg.h -- C ABI of libswg: MI355X-native Smith-Waterman protein database search.
 *
 * This is the drop-in boundary for the one hot path of Aseeef/seq-align-gpu:
 * the affine-gap three-state fill `alignment_fill_matrices`
 * (reference src/alignment.c:47-187) as it is driven, one query against a
 * database, by the OpenMP batch-of-batches loop at
 * reference src/alignment_cmdline.c:501-509.  Plain pointers and sizes only;
 * no C++ or torch types; no function here aborts or exits the process (the
 * reference asserts/exit()s: src/alignment.c:63-66, src/alignment_scoring.c:78-79)
 * -- every entry point returns 0 or a negative swg_status and leaves a message
 * retrievable with swg_last_error()/swg_global_error().
 *
 * Residues are exchanged as the reference's substitution-table indices
 * (`letters_to_index`, reference src/alignment_scoring.c:70-81): 1..26 for
 * A..Z case-folded, 31 for '*'.  Index 0 is reserved by this library as the
 * padding residue and is rejected in input.
 *
 * Threading: a swg_ctx is not thread-safe; distinct contexts may be used from
 * distinct threads (the reference's fill is re-entrant per aligner_t,
 * src/alignment_cmdline.c:504-507).  All calls are synchronous at the ABI.
 */
#ifndef SWG_H
#define SWG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWG_ABI_VERSION 3

typedef enum swg_status {
    SWG_OK = 0,
    SWG_ERR_ARG = -1,     /* bad argument (NULL, zero length, out-of-range value) */
    SWG_ERR_HIP = -2,     /* HIP runtime / launch failure; text in swg_last_error */
    SWG_ERR_NOMEM = -3,   /* host or device allocation failed */
    SWG_ERR_STATE = -4,   /* call order violated (no scoring / no query / db not resident) */
    SWG_ERR_RESIDUE = -5, /* residue index outside 1..31 (reference: exit(1) in letters_to_index) */
    SWG_ERR_IO = -6,      /* file could not be read / parsed (host helpers) */
    SWG_ERR_NODEVICE = -7 /* no usable GPU: the product path never falls back to the CPU */
} swg_status;

typedef struct swg_ctx swg_ctx; /* one GPU, one stream, one query + scoring system */
typedef struct swg_db swg_db;   /* length-sorted, binned, dword-packed database (shard) */

typedef struct swg_config {
    int device;      /* HIP device ordinal */
    int reserved[7]; /* must be zero */
} swg_config;

/* One hit of the top-K report.  Order: higher score first, ties by lower
 * database index (total order => deterministic across shards and GPUs). */
typedef struct swg_hit {
    int32_t score;
    uint32_t index; /* ORIGINAL database index (position in the caller's input) */
} swg_hit;

/* Filled by swg_search.  Times are device times from HIP events on the
 * library's stream; the reference's `Total Time` likewise counts only the fill
 * (src/alignment_cmdline.c:503-509). */
typedef struct swg_stats {
    uint64_t cells;         /* lq * sum(len_i): real cells, what GCUPS counts */
    uint64_t cells_padded;  /* cells actually computed (bin + strip padding) */
    uint64_t bytes_alg;     /* algorithmic HBM bytes of the fill */
    uint64_t n_rescored;    /* SEQUENCES that reached the ceiling of the cells they ran on and were run again on wider
                             * ones.  cell_form 0, 1: those whose int16 / wide score saturated (run again in int32);
                             * 2: those the f16 cells flagged (score >= 4096; their PAIRS are run again on int16 / wide
                             * cells, and only what saturates those too goes on to int32 -- not counted a second time);
                             * 4, 5: the f16 part's flagged sequences plus the int16 / wide part's saturated ones */
    double fill_ms;         /* 16-bit fill kernel (or the int32 fill when forced) */
    double rescore_ms;      /* everything after the fill that runs flagged work again: collection of the lists, the int16 /
                             * wide list re-run of what the f16 cells flagged, the int32 re-score */
    double topk_ms;         /* device top-K selection */
    double total_ms;        /* first kernel start .. last kernel end */
    int32_t path_bits;      /* 16 or 32: arithmetic of the main fill */
    int32_t cols_per_wave;  /* K: query columns held in registers per wavefront (systolic) / per lane (diagonal) */
    int32_t waves;          /* W: wavefronts of one workgroup (systolic: chained over the query) */
    int32_t passes;         /* query passes: systolic ceil(lq / (W*K)), diagonal ceil(lq / (group_lanes*K)) */
    int32_t workgroups;     /* grid size of the fill */
    int32_t engine;         /* 1 systolic (waves chained over the query), 2 diagonal (lane groups) */
    int32_t group_lanes;    /* diagonal engine: lanes sharing one pair of sequences (16/32/64) */
    int32_t streams;        /* diagonal engine: lane groups working in parallel */
    int32_t long_pairs;     /* diagonal engine: longest pairs run as their own class, 64 lanes each */
    int32_t long_cols_per_lane;
    int32_t long_streams;
    int32_t work_queue;     /* diagonal engine: 1 = pairs handed out by the device-side work queue */
    /* Two classes (bulk and long pairs) are launched on two HIP streams and are MEANT to run side by
     * side on every CU (the plan's cost model assumes it).  Whether they did is measured, not assumed:
     * each kernel stamps the wall clock when its first wavefront starts and when its last one ends.
     * 1 = the long class started before a tenth of the bulk's run had passed; 0 = it did not (streams
     * sharing a hardware queue -- a host program with many HIP streams should raise GPU_MAX_HW_QUEUES --
     * or its workgroups found no room); -1 = a single class, nothing to overlap. */
    int32_t classes_overlapped;
    /* launches of the main fill kernel in this search: one per pass of the query, times the segments a
     * pass of a very large database is cut into; 1 for a query of one pass */
    int32_t fill_launches;
    /* the cells the 16-bit fill ran on: 0 packed int16 (scores to 32767), 1 the wide int16 form (to 65535),
     * 2 packed f16 with gfx950's three-operand maxima (exact below 4096; a sequence that reaches it is flagged, and
     * the flagged pairs are run again on the int16 cells -- the wide form if the query can pass 32767 -- by the same
     * kernel in list mode; int32 only for what saturates those too), 3 (swg_search_multi only) the f16 cells with two
     * QUERIES per lane against one sequence instead of two sequences against one query, 4 both of the 16-bit
     * forms in one search: a query long enough to score beyond 32767 runs the sequences that could reach the f16
     * cells' ceiling -- those of split_rows rows or more -- on the wide form and everything shorter on the f16
     * cells (what those flag all the same is run again on the wide form, so the scores are exact either way), 5 the same
     * with option wide16 = 0: the long sequences on the plain int16 cells, everything from 32767 up re-scored in int32 */
    int32_t cell_form;
    int32_t split_rows;     /* cell_form 4, 5: the length from which sequences took the int16 cells; else 0 */
    int32_t fill_f16_launches; /* cell_form 4, 5: how many of fill_launches ran the f16 cells; else 0 */
    double fill_f16_ms;     /* cell_form 4, 5: the part of fill_ms spent on the f16 cells; else 0 */
    uint64_t cells_f16;     /* cell_form 4, 5: the real cells computed on the f16 cells; else 0 */
    int32_t last_pass_cols; /* diagonal engine, several passes: columns per lane of the last pass when it has a geometry of
                             * its own (fewer than cols_per_wave); else 0 */
} swg_stats;

/* ---- context ---------------------------------------------------------- */

/* Create a context on cfg->device.  Fails with SWG_ERR_NODEVICE when there is
 * no GPU -- there is deliberately no CPU backend behind this ABI. */
int swg_create(const swg_config *cfg, swg_ctx **out);
void swg_destroy(swg_ctx *ctx);
const char *swg_last_error(const swg_ctx *ctx);
const char *swg_global_error(void); /* errors of calls that have no context */
int swg_abi_version(void);

/* Tuning / test switches.  Keys: "force_bits" (0 auto | 16 | 32),
 * "engine" (0 auto | 1 systolic | 2 diagonal; int16 path only),
 * "cols_per_wave" (0 auto | systolic: 16, 24, 32 or 48 columns per wavefront; diagonal: 2..32 columns per lane),
 * "max_waves" (0 auto | systolic: waves chained over the query, 1..16; diagonal:
 * waves per workgroup, 4, 8, 12 or 16), "group_lanes" (0 auto | 16 | 32 | 64; with force_bits = 32 a forced
 * cols_per_wave x group_lanes whose int32 profile does not fit LDS is replaced by the library's pick, which
 * swg_stats reports),
 * "long_split" (-1 off | 0 auto | rows above which a pair joins the long class),
 * "autotune" (1 default: on the first search of a query length the few geometries the cost model
 * ranks best are timed on the device and the fastest is kept for that database | 0 model only),
 * "workgroups" (0 auto), "work_queue" (1 default: single-pass diagonal fills hand pairs to lane
 * groups through device-wide counters | 0 fixed streams laid out on the host), "long_helps"
 * (0 default | 1: lane groups of the long class go on with the bulk's pairs when their own are
 * done -- three wavefronts per SIMD issue as fast as four, so the less efficient helpers cost 1.5 %),
 * "prio_share" (percent of a lane group's mean share above which a bulk pair runs at raised
 * priority; default 150), "wide16" (1 default: when the query is long enough for a score to pass
 * 32767 the diagonal engine runs its wide form, exact to 65535, and only scores beyond that are
 * re-scored in int32 | 0: plain int16 and int32 re-score from 32767), "f16" (1 default: the diagonal engine's
 * 16-bit fill runs on packed-f16 cells -- gfx950's three-operand maxima, 8.5 instead of 10 instructions per
 * column pair, exact below 4096, every pair with a sequence that reaches 4096 flagged and run again on int16 cells
 * (two levels: f16 -> int16 / wide -> int32 for what saturates those) -- unless the
 * query is long enough for scores beyond 32767 (then the sequences too short to get there still do and the rest
 * runs on the wide form: swg_stats.cell_form 4) or the flagged pairs of this database held more than 1/16 of its
 * pair rows for this query | 0: int16 cells only | 2: f16 cells whenever the gap magnitudes are at most 2048), "last_pass" (1 default: the
 * last pass of a query of several passes runs with the fewest columns per lane that cover what is left | 0: like the
 * other passes), "qq" (1 default: a batch of
 * queries on the f16 cells runs two queries per lane | 0: two sequences per lane as a single query does), "side_readout" (1 default:
 * top-K selection and read-out of a search run on their own stream, beside the fill of the search
 * queued next), "batch" (8 default: pairs one work-queue request claims where pairs are short -- at most
 * "batch_blocks" 4-row token blocks long (0 default: about 40 us of work at the launch's geometry: 30 blocks at 2
 * columns per lane, 5 at 32); batch 0 / 1: every request claims one pair). */
int swg_set_option(swg_ctx *ctx, const char *key, long value);

/* Replaces scoring_t for the path (reference src/alignment_scoring.h:21-37):
 * sub[a][b] = score of query residue index a against database residue index b
 * (row = query: src/alignment.c:33,41); gap of length n costs
 * gap_open + n*gap_extend (src/alignment_scoring.c:35-36). */
int swg_set_scoring(swg_ctx *ctx, const int8_t sub[32][32], int gap_open, int gap_extend);

/* Replaces the query half of aligner_create (src/alignment.h:64-68):
 * idx[lq] are table indices; the library copies them. */
int swg_set_query(swg_ctx *ctx, const int8_t *idx, size_t lq);

/* ---- database --------------------------------------------------------- */

/* Host-only (needs no GPU).  Replaces the 16-lane transpose+pad packer of
 * reference src/alignment_cmdline.c:429-452: sequences are sorted by length
 * (descending, stable; the reference requires a pre-sorted input,
 * src/alignment_cmdline.c:431-439) and stored as one byte per residue by sorted
 * rank; 128 consecutive ranks form a bin, the unit of sharding.  That image is
 * what swg_db_upload copies to the GPU; the kernels' own layouts are built from
 * it on the device.
 * flat[offsets[i] .. offsets[i+1]) are the indices of sequence i.
 * shard_count > 1 keeps only bins b with b % shard_count == shard_rank of the
 * GLOBAL bin sequence (round-robin over GPUs: adjacent bins have near-equal
 * work); indices reported later are always original ones. */
int swg_db_pack(const int8_t *flat, const uint64_t *offsets, size_t n,
                int shard_rank, int shard_count, swg_db **out);
/* The same for a shard that was cut elsewhere (a rank that generated or read only
 * its own bins of a large database): the n_local sequences given are exactly the
 * shard, global_index[i] is sequence i's index in the whole database of n_total
 * sequences.  When they are the bins b % count == rank of the whole database's
 * sorted order, in that order, the result equals swg_db_pack(whole, rank, count). */
int swg_db_pack_shard(const int8_t *flat, const uint64_t *offsets, size_t n_local,
                      const uint32_t *global_index, size_t n_total, swg_db **out);
/* All shard_count shards of one database at once: out[r] == swg_db_pack(flat, offsets, n, r, shard_count), from ONE
 * global sort and with the shards built side by side (what swg_group_load uses: one process driving several GPUs
 * should not sort a 10M-sequence database once per device).  On error no shard is returned. */
int swg_db_pack_shards(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_count, swg_db **out);
int swg_db_upload(swg_ctx *ctx, swg_db *db); /* H2D; db becomes resident on ctx's GPU */
/* Packed-database file (host-only): the sorted, re-coded image of swg_db_pack,
 * so a large database is ingested once; swg_db_load validates the structure it reads. */
int swg_db_save(const swg_db *db, const char *path);
int swg_db_load(const char *path, swg_db **out);
void swg_db_free(swg_db *db);
size_t swg_db_count(const swg_db *db);          /* sequences in this shard */
size_t swg_db_total_count(const swg_db *db);    /* sequences given to swg_db_pack */
uint64_t swg_db_residues(const swg_db *db);     /* sum of lengths in this shard */
uint64_t swg_db_packed_bytes(const swg_db *db); /* bytes swg_db_upload copies to the GPU */
/* original index of the i-th sequence of this shard in its sorted order */
const uint32_t *swg_db_order(const swg_db *db);

/* ---- the hot path ----------------------------------------------------- */

/* Replaces the whole timed region of the reference
 * (src/alignment_cmdline.c:503-509) plus the read-out of aligner->max_scores
 * (src/tools/sw_cmdline.c:55-72) for every sequence of `db`:
 *   scores_out  NULL, or an array of swg_db_total_count(db) int32 indexed by
 *               ORIGINAL database index; only this shard's entries are written.
 *   topk_out/k  NULL/0, or room for k hits of this shard (fewer are written
 *               when the shard is smaller; *n_hits tells how many).
 * Scores are exact int32 local-alignment maxima: the 16-bit kernels flag every
 * sequence whose score reaches their ceiling and those are run again on wider cells
 * (f16 -> int16 or its wide form -> int32). */
int swg_search(swg_ctx *ctx, const swg_db *db, int32_t *scores_out,
               swg_hit *topk_out, size_t k, size_t *n_hits, swg_stats *stats);

/* The same search in two halves, so that a caller can keep the GPU busy: swg_search_begin queues
 * everything on the context's stream and returns; swg_search_end waits for that search and
 * delivers its results.  Up to 4 searches may be in flight per context (each has its own output
 * buffers; they execute in order).  want_scores != 0 is required to get scores_out at the end.
 * swg_search(...) == begin + end.  The database must stay alive and resident in between. */
int swg_search_begin(swg_ctx *ctx, const swg_db *db, int want_scores, size_t k, int *ticket);
int swg_search_end(swg_ctx *ctx, int ticket, int32_t *scores_out, swg_hit *topk_out, size_t *n_hits,
                   swg_stats *stats);

/* Many queries against one resident database in ONE pass of the hot path.  The reference takes a
 * single query per run (src/alignment_cmdline.c:381-396); its report says the design "extends
 * naturally" to many-to-many (Final Report p.7).  Query i is queries[q_offsets[i] .. q_offsets[i+1])
 * (table indices, as swg_set_query takes them); the scoring system is the context's, the context's own
 * query is left as it was.  All queries of a batch share one launch per class: a database too small to
 * fill the GPU with one query fills it with many.  Batches that cannot take that path (a query of
 * several passes, scores that may pass 32767, gap scores outside the packed int16 form, engine
 * options) are searched one query after another: same results.
 *   scores_out  NULL, or n_queries rows of swg_db_total_count(db) int32 by ORIGINAL database index;
 *   topk_out/k  NULL/0, or n_queries rows of k hits; n_hits NULL or n_queries counts;
 *   stats       NULL, or ONE record for the whole batch (times and cells summed). */
int swg_search_multi(swg_ctx *ctx, const swg_db *db, const int8_t *queries, const uint64_t *q_offsets,
                     size_t n_queries, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                     swg_stats *stats);

/* Reference-shaped replay of the call site itself: n_batches 16-lane batches
 * exactly as `alignment_fill_matrices` receives them -- db_idx_t is
 * aligner_t.seq_b_batch_indexes, [max_len][16] int8 (src/alignment.h:28,
 * src/alignment_cmdline.c:434,445), padded rows included and computed as real
 * rows like the reference does; max_scores receives
 * aligner_t.max_scores for lanes 0..vector_size-1, saturated to int16.
 * Uses the scoring and query already set on ctx. */
typedef struct swg_batch16 {
    const int8_t *db_idx_t;
    size_t max_len;
    size_t vector_size;
    int16_t *max_scores;
} swg_batch16;
int swg_fill_batches16(swg_ctx *ctx, const swg_batch16 *batches, size_t n_batches,
                       double *fill_seconds);

/* ---- alignments of reported hits -------------------------------------- */

/* The reference reports scores only (its fork removed upstream's traceback: Final Report p.7; the
 * comment at src/alignment.c:46 is what is left of it).  swg_align_hits re-runs the given pairs on
 * the GPU with the recurrence of src/alignment.c:124-161 kept whole and walks back from the best
 * match cell.  Coordinates are 0-based and half-open.  ops spells the path first to last:
 * 'M' query residue against database residue, 'I' database residue against a gap (state A of the
 * recurrence), 'D' query residue against a gap (state B); the path's substitution and gap scores
 * (a gap position costs gap_extend when it continues a gap of the same kind, else
 * gap_open + gap_extend) add up to `score`.  Ties, which the reference never had to define: the
 * best cell is the one with the highest score, then the smallest database position, then the
 * smallest query position; a state whose maximum is 0 starts the alignment; otherwise the first
 * maximal predecessor in the order H, A, B. */
typedef struct swg_alignment {
    int32_t score;  /* recomputed here; equals the search's score for the pair */
    uint32_t index; /* ORIGINAL database index, copied from the hit */
    uint32_t q_begin, q_end; /* aligned part of the query */
    uint32_t d_begin, d_end; /* aligned part of the database sequence */
    uint32_t n_ops;          /* steps of the path (0 when the score is 0) */
    uint32_t reserved;
} swg_alignment;

/* hits[n_hits]: only .index is read (every sequence must belong to this shard).  out[n_hits].
 * ops: NULL, or n_hits strings of ops_stride bytes each, NUL-terminated;
 * swg_align_ops_bound(ctx, db) = query length + longest sequence + 1 is always enough. */
int swg_align_hits(swg_ctx *ctx, const swg_db *db, const swg_hit *hits, size_t n_hits,
                   swg_alignment *out, char *ops, size_t ops_stride);
size_t swg_align_ops_bound(const swg_ctx *ctx, const swg_db *db);

/* ---- multi-GPU merge -------------------------------------------------- */

/* 64-bit sort key of a hit: (score << 32) | (0xFFFFFFFF - index).  Larger key =
 * better hit, so one max-all-reduce (RCCL, ncclMax on n_gpus*k keys) or one
 * all-gather merges the shards' top-K lists. */
uint64_t swg_hit_key(int32_t score, uint32_t index);
void swg_key_hit(uint64_t key, swg_hit *out);
/* keys[n] (zeros ignored) -> best k hits, sorted.  Returns hits written. */
size_t swg_topk_merge_keys(const uint64_t *keys, size_t n, size_t k, swg_hit *out);

/* ---- several GPUs in one process ----------------------------------------- */

/* A group owns one context per device.  swg_group_load deals the database's bins round-robin
 * over the devices; swg_group_search runs all shards concurrently and merges their top-K lists
 * with a single RCCL max-all-reduce of n_gpus*k hit keys (RCCL is loaded on first use; with one
 * device no collective is issued unless force_collective is set, which is how a 1-GPU box
 * rehearses the path).  scores_out is indexed by original database index; stats, if given, is
 * an array of swg_group_size() entries.  (bench.py uses the other arrangement: one process per
 * GPU, torch.distributed over RCCL.) */
typedef struct swg_group swg_group;
int swg_group_create(const int *devices /* NULL: 0..n-1 */, int n, int force_collective, swg_group **out);
void swg_group_destroy(swg_group *g);
int swg_group_size(const swg_group *g);
const char *swg_group_last_error(const swg_group *g);
int swg_group_set_option(swg_group *g, const char *key, long value);
int swg_group_set_scoring(swg_group *g, const int8_t sub[32][32], int gap_open, int gap_extend);
int swg_group_set_query(swg_group *g, const int8_t *idx, size_t lq);
int swg_group_load(swg_group *g, const int8_t *flat, const uint64_t *offsets, size_t n);
int swg_group_search(swg_group *g, int32_t *scores_out, swg_hit *topk_out, size_t k, size_t *n_hits,
                     swg_stats *stats);
/* swg_align_hits for a group: every hit is re-run on the GPU whose shard holds the sequence. */
int swg_group_align_hits(swg_group *g, const swg_hit *hits, size_t n_hits, swg_alignment *out, char *ops,
                         size_t ops_stride);
size_t swg_group_align_ops_bound(const swg_group *g);

#ifdef __cplusplus
}
#endif
#endif /* SWG_H */
