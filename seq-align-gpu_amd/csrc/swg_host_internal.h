// swg_host_internal.h -- host-side object definitions behind the opaque handles
// of include/swg.h.  Not part of the public ABI.
#pragma once
#include "../../include/swg.h"
#include "swg_internal.h"

#include <functional>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <utility>
#include <vector>

// Allocator whose resize() leaves new elements uninitialised: the big residue arrays are filled
// by parallel loops, and pages should be first touched by the threads that fill them instead of by
// one thread writing zeros (a 10M-sequence database is 8 GB of them).
template <class T> struct SwgNoInit {
    using value_type = T;
    SwgNoInit() = default;
    template <class U> SwgNoInit(const SwgNoInit<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t) { ::operator delete(p); }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&...a)
    {
        ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
    }
    template <class U> bool operator==(const SwgNoInit<U> &) const { return true; }
    template <class U> bool operator!=(const SwgNoInit<U> &) const { return false; }
};

// Stream layout of the diagonal engine: pairs of adjacent sorted ranks, dealt to
// n_streams lane groups longest first; tokens are stored stream-major.
struct SwgDiagLayout {
    uint32_t n_streams = 0;
    uint32_t streams_per_wg = 0;
    uint64_t pair_begin = 0, pair_end = 0; // pairs of the sorted order laid out here
    uint64_t total_blocks = 0;      // 4-row token blocks over all streams
    uint64_t max_stream_blocks = 0;
    uint64_t pair_rows_total = 0;   // sum over pairs of (2 + longer length), unpadded
    std::vector<uint64_t> stream_off;      // [n_streams+1]
    std::vector<uint32_t> stream_pairs;    // pair ids, stream-major
    std::vector<uint32_t> stream_pair_off; // [n_streams+1]
    std::vector<uint32_t> tok;             // 4 dwords per block (one 32-bit token per row)
    // device image
    uint4 *d_tok = nullptr;
    uint64_t *d_stream_off = nullptr;
    uint32_t *d_stream_pairs = nullptr;
    uint32_t *d_stream_pair_off = nullptr;
    uint2 *d_scratch = nullptr;
    uint64_t d_scratch_rows = 0;
};

// Per-search device counters: words [0..15] work queue of the systolic engine, saturated count,
// re-score queue, top-K; then the sharded pair queues of the diagonal engine's two classes and their
// per-SIMD wavefront-rank counters.
#define SWG_QUEUE_WORD(c) (32u + (uint32_t)(c) * SWG_DYN_SHARDS * SWG_DYN_SHARD_STRIDE)
#define SWG_RANK_WORD(c) (SWG_QUEUE_WORD(2) + (uint32_t)(c) * SWG_DYN_SIMD_SLOTS)
#define SWG_COUNTER_BYTES ((size_t)SWG_RANK_WORD(2) * 4u)

// Pair-major tokens for the work-queue form of the diagonal engine: pair p of the sorted
// order owns blocks [pair_off[p], pair_off[p+1]); independent of the launch geometry.
struct SwgPairTokens {
    bool tried = false, ok = false;
    bool host_built = false; // diagnostics: tokens came from the host builder (option "host_tokens")
    uint64_t total_blocks = 0;
    uint4 *d_tok = nullptr;
    uint32_t *d_pair_off = nullptr;
    uint2 *d_edge[2] = {nullptr, nullptr};    // multi-pass: (M,B) per row between consecutive passes, ping-pong
    int2 *d_edge32[2] = {nullptr, nullptr};   // the same for the int32 work-queue kernel: per row and per sequence of the pair
    int32_t *d_edge32d[2] = {nullptr, nullptr}; // ... and the third edge value of its exact cells (gap scores of any sign)
    uint64_t edge_blocks = 0, edge32_blocks = 0, edge32d_blocks = 0; // token blocks the edge buffers were allocated for (a re-filled database may have grown)
    std::vector<uint32_t> pair_blocks_prefix; // host copy of pair_off
};

struct SwgDiagPlan {
    int variant = 0, K = 0, G = 0, npass = 0, W = 0, workgroups = 0;
    int wide = 0; // scores to 65535 (values biased by -32768)
    int f16 = 0;  // packed-f16 cells with three-operand maxima: scores below 4096, anything above flagged and re-scored
    // wide && f16_from > 0: both forms in one class -- the pairs before f16_from (the longest: sorted order) on the wide
    // form, those from it on on the f16 cells (same geometry, a launch each per pass)
    uint32_t f16_from = 0;
    // Several passes: the last one covers what is left of the query with the fewest columns per lane that do
    // (its own kernel instantiation and profile layout; edges do not depend on the geometry).  -1: as the others.
    int last_variant = -1, last_K = 0;
    uint32_t n_streams = 0;
    size_t lds_bytes = 0;
    double est_ms = 0.0;
};
// The diagonal engine's work split: class 0 = the bulk of the pairs, class 1 = the few
// longest ones, which would otherwise be the serial tail of the whole search.  The long
// class runs beside the bulk on a second HIP stream with 64 lanes per pair and as few
// columns per lane as cover the query, i.e. with the shortest possible chain per row.
struct SwgDiagWork {
    int n_classes = 0;
    SwgDiagPlan plan[2];
    uint64_t pair_begin[2] = {0, 0}, pair_end[2] = {0, 0};
};

// What the autotuner keeps per query length: the engine and its geometry.
struct SwgTuned {
    int engine = 2;    // 1 systolic, 2 diagonal
    int systolic_K = 0; // engine 1: columns per wavefront of the chosen instantiation
    double ms = 0;
    SwgDiagWork wk;    // engine 2
};

struct swg_db {
    // host image: what swg_db_pack builds and swg_db_save writes.  Sequences by sorted rank (length
    // descending, stable); 128 consecutive ranks form a bin (the unit of sharding and of the systolic
    // engine); the last bin of a shard may have empty slots (order = ~0, length 0).
    size_t n_total = 0;             // sequences of the whole database
    size_t n_local = 0;             // sequences of this shard
    uint32_t n_bins = 0;            // bins of this shard
    uint32_t max_nblk = 0;          // row-blocks of the longest bin
    uint64_t residues = 0;          // sum of lengths (this shard)
    uint64_t rows_padded = 0;       // sum over bins of nblk*4*128 (rows the bin-based kernels walk)
    std::vector<uint64_t> bin_off;  // [n_bins] dword offset of a bin in the device bin image
    std::vector<uint32_t> bin_nblk; // [n_bins]
    std::vector<uint32_t> order;    // [n_bins*128] original index of each slot, ~0u = empty
    std::vector<uint32_t> lens;     // [n_bins*128]
    // residue bytes (index<<3) by sorted rank; every sequence starts on a 4-byte boundary and is
    // filled up to one with the padding residue 0, so that a sequence is a run of whole dwords
    std::vector<uint8_t, SwgNoInit<uint8_t>> codes;
    std::vector<uint64_t> code_off; // [n_bins*128+1] byte offsets into codes (multiples of 4)
    SwgPairTokens ptok;             // pair-major tokens (work-queue form of the diagonal engine)
    SwgDiagLayout diag[2];          // stream layouts of the diagonal engine: [0] bulk, [1] long pairs
    std::vector<uint64_t> pair_rows_prefix; // rows of the pairs before pair p (swg_db_pair_rows: built on first use)
    std::map<uint64_t, SwgTuned> tuned; // query length -> engine + geometry that measured fastest on this device
    // What the last finished search of this database saw (plans of later searches only: results never depend
    // on it).  sat_hint: sequences its 16-bit fill flagged for the re-score (-1: no search yet), by which
    // the re-score's lane-group width is picked without reading the count back in the middle of a search;
    // f16_veto_epoch: the (query, scoring) epoch for which the packed-f16 cells flagged so many rows (a
    // database full of close relatives of the query) that the int16 cells are the faster first step.
    long long sat_hint = -1;
    uint64_t f16_veto_epoch = 0;
    // both 16-bit forms in one search (swg_search_begin): for the last length threshold asked about, the first pair
    // of the sorted order whose sequences are all shorter, and the residues from it on
    uint32_t split_rows = 0, split_pair = 0;
    uint64_t split_residues = 0;
    // a database whose pair tokens were built straight from reference-shaped 16-lane batches
    // (swg_fill_batches16): there are no residue bytes by sorted rank, so nothing that needs them can run
    bool tokens_only = false;
    // device image (valid after swg_db_upload): the residue bytes and three words per slot; the pair
    // tokens (ptok) and the bin image are built FROM them on the device, the bins only when an
    // engine that reads them is used (swg_ensure_bins)
    int device = -1;
    uint32_t *d_codes = nullptr;    // codes as dwords
    uint64_t *d_code_off = nullptr; // [n_slots+1] DWORD offsets into d_codes
    uint32_t *d_lens = nullptr;     // [n_slots]
    uint32_t *d_order = nullptr;    // [n_slots]
    uint32_t *d_packed = nullptr;   // bin image (systolic engine, int32 kernels): lazily built
    uint64_t *d_bin_off = nullptr;
    uint32_t *d_bin_nblk = nullptr;
    uint64_t upload_bytes = 0;      // bytes the last swg_db_upload copied over PCIe
    // per-search output buffers, one set per in-flight slot (allocated on first use); the
    // plain members below point at the set of the search being queued
    struct Bufs {
        int32_t *d_scores = nullptr;    // [n_bins*128] by sorted rank
        uint32_t *d_list = nullptr;     // [2*n_bins*128]: flagged pairs of the f16 fill, then (second half) saturated ranks
        uint32_t *d_counters = nullptr; // [0] work queue, [1] saturated count, [2] re-score queue, [3..5] top-K, [6] flagged rows / 16,
                                        // [8..15] class stamps, [16] sequences the f16 fill flagged, [17] their pairs
        uint64_t *d_keys = nullptr;     // top-K candidate keys (SWG_TOPK_CAND_CAP)
        uint32_t *d_hist = nullptr;     // top-K score histogram
    } bufs[4];
    int32_t *d_scores = nullptr;
    uint32_t *d_list = nullptr;
    uint32_t *d_counters = nullptr;
    uint64_t *d_keys = nullptr;
    uint32_t *d_hist = nullptr;
};

// One search in flight: its timing events, host-side landing buffers and what swg_search_end
// needs to finish it.  Device buffers are shared: everything of one context runs in stream
// order, so search i+1 cannot touch them before search i has copied its results out.
#define SWG_MAX_INFLIGHT 4
struct SwgSlot {
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_done = nullptr; // polled from user space instead of a blocking stream sync
    uint64_t score_bound = ~0ull; // the largest score this search's query can reach against this database (launch_diag: f16_wipe)
    bool main_f16 = false;        // the systolic fill ran on the packed-f16 cells
    bool side = false;            // this search's top-K and read-out were queued on the read-out stream (stream3)
    bool busy = false;
    const swg_db *db = nullptr;
    swg_db::Bufs bufs; // the output buffers this search writes
    size_t k = 0, first_chunk = 0;
    bool want_scores = false, dev_topk = false, need_scores = false, two_ends = false, may_saturate = false;
    bool use_diag = false, use_diag32 = false, use_q32 = false;
    bool used_f16 = false;  // the fill ran on the packed-f16 cells (all of it, or the pairs from plan.f16_from on)
    uint32_t split_rows = 0;     // both forms: the length threshold, and the residues that ran on the f16 cells
    uint64_t split_residues = 0;
    uint64_t epoch = 0;     // the context's (query, scoring) epoch the search was queued under
    SwgDiagWork wk32; // int32 work-queue fill of the whole database
    int bits = 0, npass32 = 0, main_K = 0, main_W = 0, main_npass = 0, main_wgs = 0;
    int fill_launches = 0; // launches of the bulk class's fill kernel (passes x segments)
    int fill_f16_launches = 0; // both forms in one class: those of them that ran the f16 cells
    SwgDiagWork wk;
    swg_stats st;
    uint64_t *h_cand = nullptr;     // pinned, SWG_TOPK_CAND_CAP keys
    uint32_t *h_counters = nullptr; // pinned, 32 words
    int32_t *h_scores = nullptr;    // pinned landing buffer of the score read-out, grown on demand
    size_t h_scores_cap = 0;        // entries
};

// What swg_fill_batches16 keeps between calls (the reference calls it once per macro-batch with buffers of
// recurring size, src/alignment_cmdline.c:459-509): pinned staging for the batches as they are, their device
// copy, and a database object whose device buffers are re-filled, grown only when a call needs more.
struct SwgBatch16Cache {
    swg_db *db = nullptr;
    size_t slots_cap = 0, pairs_cap = 0, stage_cap = 0;
    uint64_t blocks_cap = 0;
    uint8_t *h_stage = nullptr, *d_stage = nullptr;   // the batches' [max_len][16] bytes, end to end
    uint8_t *h_meta = nullptr;                        // pinned: pair sources, pair lengths, pair offsets, lengths, order
    uint64_t *d_pair_src = nullptr;
    uint32_t *d_pair_len = nullptr;
};

struct swg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr; // long-pair kernel runs beside the bulk kernel
    hipStream_t stream3 = nullptr; // top-K and read-out of a finished fill, beside the next search's fill
    int n_cu = 0;
    unsigned long n_begun = 0; // searches queued so far (the read-out stream is made for the second)
    std::string err;
    // scoring
    bool have_scoring = false;
    int8_t sub[32][32];
    int gap_open = 0, gap_extend = 0;
    // query: the host copy, and pinned staging buffers for the copy to the device (a buffer is not
    // written again before the copy that reads it has completed, so set_query / search_begin can be
    // streamed without a wait)
    std::vector<int8_t> query;
    int8_t *h_query_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t h_query_stage_cap[4] = {0, 0, 0, 0};
    hipEvent_t ev_query_stage[4] = {nullptr, nullptr, nullptr, nullptr};
    int query_stage_next = 0;
    // options
    long opt_force_bits = 0, opt_cols = 0, opt_max_waves = 0, opt_workgroups = 0, opt_engine = 0, opt_group = 0, opt_long_split = 0, opt_autotune = 1, opt_dynamic = 1, opt_prio_share = 150, opt_long_helps = 0, opt_wide = 1, opt_side_readout = 1, opt_f16 = 1, opt_qq = 1, opt_last_pass = 1;
    long opt_wave_budget = 0, opt_q32_waves = 0;
    long opt_batch = 8, opt_batch_blocks = 0; // work queue: pairs one request claims where pairs are short (blocks; 0: about 40 us of work, from the geometry)
    uint32_t opt_seg_blocks = SWG_DYN_SEG_BLOCKS; // token blocks per launch of the multi-pass fill (option "segment_blocks": tests)
    // device state
    int8_t *d_sub = nullptr;
    int8_t *d_query = nullptr;
    size_t d_query_cap = 0;
    // [0] int16, whole 4-column chunks per lane; [1] int32 (bin-based kernels); [2] / [3] int16 in per-lane
    // slices padded to whole chunks, long class / bulk; [4] / [5] int32 in 2-column chunks (work-queue
    // int32 kernel), bulk or list / long class
    // int32 kernel), bulk or list / long class; [6] int16, the re-run of the pairs the f16 cells flagged
    uint8_t *d_profile[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t d_profile_cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t profile_tag[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // identifies (query, scoring, geometry) currently built
    uint64_t epoch = 1;               // bumps whenever scoring or query change
    uint32_t *d_scratch = nullptr;
    size_t d_scratch_cap = 0; // dwords
    SwgBatch16Cache b16;
    SwgSlot slots[SWG_MAX_INFLIGHT];
    SwgSlot *cur = nullptr; // slot whose events the launch helpers record into
    int next_slot = 0;
};

int swg_set_global_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int swg_set_ctx_error(swg_ctx *ctx, int code, const char *fmt, ...)
    __attribute__((format(printf, 3, 4)));
void swg_db_release_device(swg_db *db);
// test hook: the pair-token image as the device built it, or as the host restatement builds it
// test hook: the next visit of the named site throws std::bad_alloc (swg_api.cpp); 0 disarms
extern "C" void swg_debug_fail_alloc(int site);
extern "C" int swg_debug_plan(const swg_db *db, size_t lq, int n_cu, int32_t *out);
// the systolic engine's estimate from the host's bin table (swg_diag_host.cpp); it is picked over the lane groups when it
// wins by this margin (both models are good to about 10 %)
#define SWG_SYSTOLIC_MARGIN 0.85
double swg_systolic_estimate_ms(const swg_db *db, size_t lq, int n_cu, int *best_K, bool f16 = false);
double swg_diag_short_pair_factor(const swg_db *db, const SwgDiagPlan &pl, int form); // lane groups on short pairs: what the fitted estimate misses
extern "C" int swg_debug_engine(const swg_db *db, size_t lq, int n_cu, int form, int32_t *out);
extern "C" int swg_debug_split(swg_db *db, size_t lq, uint64_t qbound, uint64_t *out);
// decisions swg_search_begin makes on top of the planner's geometry (host only: swg_diag_host.cpp)
bool swg_plan_last_pass(const SwgDiagPlan &pl, size_t lq, int *variant, int *K);
uint32_t swg_split_rows(size_t lq, uint64_t qbound);
void swg_db_split_at(swg_db *db, uint32_t rows);
extern "C" int swg_debug_pair_tokens(swg_ctx *ctx, swg_db *db, int from_host, uint32_t *out, size_t cap_dwords,
                                     size_t *n_dwords);

// swg_diag_host.cpp (host only)
// geometry of both classes for one query length on one device; returns the number of
// classes (0: the diagonal engine cannot run this with the given options)
int swg_plan_diag_work(const swg_db *db, size_t lq, int n_cu, long opt_cols, long opt_group, long opt_waves,
                       long opt_long_split, bool allow_split, bool work_queue, SwgDiagWork *wk, double copies = 1.0,
                       int form = 0); // form: the cells the plan is for (2: packed f16, 8.5 instead of 10 instructions per column pair)
// every geometry the model considered, best estimate first (the autotuner times the first few)
int swg_plan_diag_candidates(const swg_db *db, size_t lq, int n_cu, long opt_cols, long opt_group, long opt_waves,
                             long opt_long_split, bool allow_split, bool work_queue,
                             std::vector<SwgDiagWork> *cands, double copies = 1.0, int form = 0);
// 0 on success; -1 when the database is too large for 32-bit block offsets.  tok == NULL: only
// pair_off (the tokens themselves are built on the device, swg_launch_build_tokens); otherwise also
// the host builder's token image, which the tests compare the device's with.
int swg_build_pair_tokens(const swg_db *db, std::unique_ptr<uint32_t[]> *tok, size_t *tok_dwords,
                          std::vector<uint32_t> *pair_off);
void swg_build_diag_layout(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint32_t n_streams,
                           uint32_t streams_per_wg, SwgDiagLayout *out);
// every shard of one database from one global sort (swg_pack.cpp); ready(r, shard) runs on the thread that built shard r
int swg_pack_shards(const int8_t *flat, const uint64_t *offsets, size_t n, int shard_count, swg_db **out,
                    const std::function<int(int, swg_db *)> &ready);
extern "C" unsigned long swg_debug_sort_count(void); // test hook: global sorts run by this process so far
void swg_stage_batches16(const swg_batch16 *batches, const uint32_t *order, const uint64_t *stage_off, size_t lo, size_t hi,
                         uint8_t *stage);
void swg_untranspose_batches16(const swg_batch16 *batches, size_t n_batches, const size_t *first_rec,
                               const uint64_t *rec_off, int8_t *flat);
uint64_t swg_db_pair_count(const swg_db *db);
// rows (2 reset rows + longer length) of the pairs [pair_begin, pair_end): total and longest
uint64_t swg_db_pair_rows(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint64_t *longest_rows);
// how many leading (longest) pairs have more than `rows` rows
uint64_t swg_db_pairs_longer_than(const swg_db *db, uint64_t rows);
