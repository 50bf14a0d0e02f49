// swg_internal.h -- shared between the C-ABI host layer (swg_api.cpp) and the
// gfx950 kernels (swg_kernels.hip).  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Bin image in HBM (built on the device by swg_build_bins_kernel when an engine that reads it is used):
//   sequences sorted by length (descending), 128 consecutive ones form a BIN;
//   a bin of nblk row-blocks is nblk*128 dwords:  dword[blk*128 + SWG_BIN_COLUMN(rank)]
//   holds rows 4*blk..4*blk+3 of the bin's rank-th sequence, one byte per row, byte = index<<3
//   (the LDS byte offset of that residue's profile row inside a 256-byte chunk);
//   rows past a sequence's end are 0 = the padding residue.
// A wavefront therefore reads 64 consecutive dwords (256 B) per row-block per
// half-bin: fully coalesced, each residue byte fetched from HBM exactly once.
#define SWG_BIN 128        // sequences per bin (64 lanes x 2 packed int16 halves)
#define SWG_ROWS_PER_BLK 4 // DB rows per row-block = residues per dword
// dword column of a bin's sorted rank s: ranks 2l and 2l+1 share lane l (low/high int16 half)
#define SWG_BIN_COLUMN(s) ((((s)&1u) * 64u) + ((s) >> 1))

struct SwgFillParams {
    const uint32_t *residues;   // packed bins
    const uint64_t *bin_off;    // [n_bins] dword offset of a bin's block 0
    const uint32_t *bin_nblk;   // [n_bins] row-blocks of a bin
    uint32_t n_bins;
    const uint8_t *profile;     // [npass][W] slices, see build_profile
    int32_t *scores;            // [n_bins*128] by sorted slot id; atomicMax target
    uint32_t *queue;            // work counter, zeroed before the launch
    const uint32_t *list;       // int32 path: optional list of slot ids to (re)score
    const uint32_t *list_count; // device-side length of `list` (NULL: use n_items)
    uint32_t n_items;           // work items when list_count == NULL
    uint32_t npass;             // query passes
    int32_t go, ge;             // int16 path: |gap_open+gap_extend|, |gap_extend| in both halves
                                // int32 path: signed gap_open+gap_extend, gap_extend
    uint32_t *scratch;          // multi-pass boundary spill, per workgroup
    uint64_t scratch_wg_dwords; // dwords of scratch owned by one workgroup
};

// Diagonal engine (swg_diag_kernel): streams of sequence pairs, one per lane group.
struct SwgDiagParams {
    const uint4 *tok;                // stream-major token blocks (4 rows, one 32-bit token each)
    const uint64_t *stream_off;      // [n_streams+1] block offset of each stream
    const uint32_t *stream_pairs;    // pair ids in stream order
    const uint32_t *stream_pair_off; // [n_streams+1]
    uint32_t n_streams;
    const uint8_t *profile;          // [npass][G lanes][KP/4 chunks][32][4] int16
    int32_t *scores;                 // by sorted rank: pair p -> 2p, 2p+1
    uint2 *scratch;                  // multi-pass spill, one (M,B) per stream row
    uint32_t npass, G;
    uint32_t go, ge;                 // |gap_open+gap_extend|, |gap_extend| in both halves
    uint32_t prio_blocks;            // wavefronts whose longest stream has >= this many blocks run at raised priority
    uint64_t *trace;                 // diagnostics (SWG_TRACE): per wavefront {start, end, blocks}, or null
};

// Row tokens of the lane-group kernels: byte 0 = residue of the pair's first sequence << 3, byte 1 = of
// its second, then flags.
#define SWG_TOK_RESET 0x10000u
#define SWG_TOK_LAST 0x20000u
#define SWG_TOK_RESET2 0x80000u // with RESET: the second of a pair's two reset rows (the wide cells wipe their state on the first only)
#define SWG_TOK_IDLE 0x40000u // first row after a lane group's last pair (several passes only): the tail lane stops storing edges

// The same fill with pairs handed out by device-wide counters.
#define SWG_DYN_SHARDS 8u        // counters per range; shard c hands out pairs begin + c + 8k
#define SWG_DYN_SHARD_STRIDE 32u // dwords between counters: one 128-byte line each
#define SWG_DYN_SIMD_SLOTS 8192u // wavefront-rank counters, one per physical SIMD (xcc, se, sh, cu, simd)
#define SWG_DYN_SEG_BLOCKS ((1u << 26) - 64u) // token blocks one launch with edges can address (see swg_diag_dyn_kernel)
struct SwgDiagDynParams {
    const uint4 *tok;         // pair-major token blocks (4 rows, one 32-bit token each), longest pair first
    uint32_t zero_block;      // index of a block of zeros in tok (padding rows for lanes that feed no pair)
    const uint32_t *pair_off; // [n_pairs+1] block offset of each pair's tokens
    uint32_t q_begin, q_end;  // this launch serves pairs [q_begin, q_end) ...
    uint32_t *queue;          // ... handed out by these SWG_DYN_SHARDS counters (zero before the launch)
    // short pairs by the batch (0, 0, 0: every request claims one pair): after batch_u1 single pairs per shard a
    // request claims batch_B consecutive pairs, batch_u2 times per shard; what is left of the range goes out one by
    // one again (see the kernel's event code for the counter -> pair map)
    uint32_t batch_u1, batch_u2, batch_B;
    // f16 cells: 1 = a score of this search may reach 32768, beyond which two reset rows no longer clear a lane's state:
    // lanes test their best on reset rows and wipe by hand (0: no score can get there, the test is skipped)
    uint32_t f16_wipe;
    // list mode (or null): the counters hand out positions 0 .. *list_count - 1 of `list`, whose entries are pair
    // ids; an entry outside [q_begin, q_end) is skipped (it belongs to another segment's launch).  No second range.
    const uint32_t *list;
    const uint32_t *list_count;
    uint32_t q2_begin, q2_end; // then helps with [q2_begin, q2_end) (empty: none), which another
    uint32_t *queue2;          // launch is serving off these counters
    const uint8_t *profile;   // [G lanes][KP/4 chunks][32][4] int16
    int32_t *scores;          // by sorted rank: pair p -> 2p, 2p+1
    uint32_t pair_limit;      // pairs the score array has room for: nothing is written beyond it
    uint32_t G;
    uint32_t go, ge;          // |gap_open+gap_extend|, |gap_extend| in both halves
    uint32_t prio_blocks;     // a wavefront feeding a pair of >= this many blocks runs at raised priority
    uint32_t prio_blocks2;    // the same for pairs of the second range
    uint32_t *simd_ranks;     // [SWG_DYN_SIMD_SLOTS] zero before the search
    const uint2 *edge_in;     // one pass of several: (M,B) left edge per row from the previous pass (null: first)
    uint2 *edge_out;          // ... right edge per row for the next pass (null: last)
    // ... of token blocks [seg_origin, seg_origin + seg_blocks): a launch with edges addresses tokens and
    // edges by 32-bit byte offsets from the segment's start, so seg_blocks <= SWG_DYN_SEG_BLOCKS and the
    // pairs [q_begin, q_end) lie inside the segment
    uint32_t seg_origin, seg_blocks;
    uint32_t turn_levels;     // priorities the other wavefronts rotate through: 3 beside a long class, else 4
    uint64_t *trace;          // diagnostics (SWG_TRACE) or null
    // several queries in one launch (swg_search_multi): grid.y = query; workgroup row y reads profile +
    // y * profile_stride (bytes), takes pairs off queue + y * queue_stride (dwords) and writes scores +
    // y * score_stride (entries).  All zero for a single query.
    uint64_t profile_stride, score_stride;
    uint32_t queue_stride;
    // diagnostics that cost nothing: [0] wall clock when the first wavefront of the launch started
    // (atomic min), [1] when the last one ended (atomic max); zero-initialised = not run.  Null: none.
    unsigned long long *stamps;
};

// The int32 fill with a work queue (swg_diag32q_kernel): items are sequences (sorted ranks), either
// the ranks [q_begin, q_end) or the entries of a device-side list.
struct SwgDiagQ32Params {
    const uint4 *tok;           // pair-major token blocks
    uint32_t zero_block;        // index of a block of zeros in tok
    const uint32_t *pair_off;   // [n_pairs+1]
    uint32_t q_begin, q_end;    // ranks to score (list == NULL)
    const uint32_t *list;       // or: ranks to score ...
    const uint32_t *list_count; // ... and how many (device side)
    uint32_t *queue;            // SWG_DYN_SHARDS counters, zero before the launch
    const uint8_t *profile;     // [G lanes][KP/2 chunks][32][2] int32
    int32_t *scores;            // by sorted rank
    uint32_t seq_limit;         // ranks the score array has room for
    uint32_t G;
    int32_t go, ge;             // |gap_open+gap_extend|, |gap_extend|
    uint32_t prio_blocks;       // a wavefront feeding a sequence of >= this many blocks runs at raised priority
    uint32_t *simd_ranks;       // [SWG_DYN_SIMD_SLOTS] zero before the launch
    uint32_t turn_levels;
    // one pass of several: (M, B) left edge per row from the previous pass (null: first), right edge per row
    // for the next (null: last); index = 2 * (row's position in the pair-major token order) + (X or Y)
    const int2 *edge_in;
    int2 *edge_out;
    // the exact cells (gap scores of any sign; go / ge are then the SIGNED scores gap_open + gap_extend, gap_extend):
    // their third edge value, D = max(H, A, B), same index
    const int32_t *edge_d_in;
    int32_t *edge_d_out;
};
// Two queries per lane against one sequence (swg_diag_qq_kernel): items are the sequences (sorted ranks)
// [q_begin, q_end); row y of the grid works for query pair y.
struct SwgDiagQQParams {
    const uint4 *tok;         // pair-major token blocks
    uint32_t zero_block;      // index of a block of zeros in tok
    const uint32_t *pair_off; // [n_pairs+1]
    uint32_t q_begin, q_end;  // ranks to score
    uint32_t *queue;          // SWG_DYN_SHARDS counters per query pair (+ y * queue_stride dwords), zero before the launch
    uint32_t queue_stride;
    const uint8_t *profile;   // [G lanes][KP/2 chunks][32][2] x (2 queries x f16), + y * profile_stride bytes
    uint64_t profile_stride;
    int32_t *scores;          // query q's scores by sorted rank at scores + q * score_stride; pair y = queries 2y, 2y+1
    uint64_t score_stride;
    uint32_t n_queries;       // (an odd batch's last pair has no second query)
    uint32_t seq_limit;       // ranks a score array has room for
    uint32_t G;
    uint32_t go, ge;          // |gap_open+gap_extend|, |gap_extend| as f16 numbers in both halves
    uint32_t prio_blocks;     // a wavefront feeding a sequence of >= this many blocks runs at raised priority
    uint32_t *simd_ranks;     // [SWG_DYN_SIMD_SLOTS] zero before the launch
    uint32_t turn_levels;
};
hipError_t swg_launch_diag_qq(int variant, int W, int workgroups, int n_pairs, const SwgDiagQQParams &p, hipStream_t stream);
// profiles of the query pairs (order[2y], order[2y+1]) of a batch: pair y's profile of ncols layout columns goes to
// d_profiles + y * ncols * 32 * 4
hipError_t swg_launch_build_profiles_qq(const int8_t *d_sub, const int8_t *d_queries, const uint32_t *d_q_off, const uint32_t *d_order,
                                        uint32_t n_queries, uint32_t ncols, int k_real, int k_padded, uint8_t *d_profiles,
                                        hipStream_t stream, int swizzle_lanes = 0);
int swg_q32_padded_cols(int K);
size_t swg_diag32q_lds_bytes(int K, int G, int W);
// variant: index into the diagonal variants (swg_diag_variant_info gives its K)
#define SWG_X32_WAVES_ABOVE16 12 // wavefronts per CU the exact int32 cells are compiled for when K > 16
#define SWG_X32_MAX_K 28         // ... and the most columns per lane they hold without spilling
// exact: the reference's recurrence term by term (12 instructions per cell, gap scores of any sign) instead of the
// reduced algebra (8 per cell, non-positive gap scores)
hipError_t swg_launch_diag32q(int variant, bool edges, bool exact, int W, int workgroups, const SwgDiagQ32Params &p, hipStream_t stream);

struct SwgKernelInfo {
    int bits;      // 16 or 32
    int K;         // query columns per wavefront
    int max_waves; // largest W this instantiation was compiled for
    size_t lds_per_wave;
    size_t lds_fixed;
    int nb;        // boundary dwords per row
    int elem_size; // profile element bytes
};

// Launchers (swg_kernels.hip).  variant selects the (K, MAXW) instantiation.
int swg_num_variants(int bits);
SwgKernelInfo swg_variant_info(int bits, int variant);
hipError_t swg_launch_fill(int bits, int variant, int W, int workgroups,
                           const SwgFillParams &p, hipStream_t stream, bool f16 = false);

// int32 diagonal engine: 64 lanes x SWG_DIAG32_K columns per pass, one sequence per wavefront.
// Uses SwgFillParams: list/list_count/n_items = sequences (sorted ranks) to score, queue = work
// counter, profile = int32 [npass][16*K][32][4], scratch_wg_dwords = dwords per WAVEFRONT.
#define SWG_DIAG32_K 16
hipError_t swg_launch_diag32(int W, int workgroups, const SwgFillParams &p, hipStream_t stream);

int swg_num_diag_variants();
SwgKernelInfo swg_diag_variant_info(int variant); // K, max_waves (wave budget of one CU)
hipError_t swg_launch_diag(int variant, bool multipass, bool wide, int W, int workgroups, size_t lds_bytes,
                           const SwgDiagParams &p, hipStream_t stream);
size_t swg_diag_dyn_lds_bytes(int K, int G, int W);
// form: the cells (CellsDiag): 0 packed int16, 1 wide int16 (edges only), 2 packed f16 with three-operand maxima
hipError_t swg_launch_diag_dyn(int variant, bool edges, int form, int W, int workgroups, const SwgDiagDynParams &p,
                               hipStream_t stream, int n_queries = 1);
// profiles of n_queries queries (query i = queries[q_off[i] .. q_off[i+1])) in one launch: query i's
// profile of ncols layout columns goes to d_profiles + i * ncols * 32 * 2 (int16)
hipError_t swg_launch_build_profiles_multi(const int8_t *d_sub, const int8_t *d_queries, const uint32_t *d_q_off,
                                           uint32_t n_queries, uint32_t ncols, int k_real, int k_padded,
                                           uint8_t *d_profiles, hipStream_t stream, int swizzle_lanes = 0, int f16 = 0);

// LDS bank swizzle of the lane-group kernels' profile (1 = on): lane g of a group keeps the row of residue r
// of each of its chunks at position r ^ (g & 31) instead of r, and forms its read address with an XOR instead
// of an add (same instruction count).  Lanes of one LDS access that hold the SAME residue -- the common
// case, proteins being what they are -- then read different bank pairs instead of the same one.
#ifndef SWG_LDS_SWIZZLE
#define SWG_LDS_SWIZZLE 1
#endif

// profile[(col/4)*32*4 + code*4 + col%4] = sub[query[col]][code] (code 0 and
// col >= lq: pad value).  elem_size 2 -> int16 pad -32768, 4 -> int32 pad -2^29.
// chunk_cols: columns per chunk, [col/chunk][32][chunk] (4 everywhere except diagonal K % 4 == 2)
// k_real / k_padded: a lane's slice of the diagonal engine is k_padded layout columns holding k_real
// query columns (equal everywhere except for an odd K); ncols counts layout columns
// swizzle_lanes: 0 = rows in residue order (systolic engine, bin-based int32 kernel); G = the lane-group
// width of the kernel that will read it: a lane's rows are stored at residue ^ (lane-in-group & 31)
hipError_t swg_launch_build_profile(const int8_t *d_sub, const int8_t *d_query,
                                    uint32_t lq, uint32_t ncols, int elem_size, int chunk_cols, int k_real,
                                    int k_padded, uint8_t *d_profile, hipStream_t stream, int swizzle_lanes = 0,
                                    int f16 = 0, // f16: elem_size 2 entries are f16 numbers (pad -65504) for the packed-f16 cells
                                    uint32_t qcol0 = 0); // the query column layout column 0 stands for
int swg_diag_padded_cols(int K); // layout columns of a lane's slice

// Per-database layouts from the uploaded residue dwords (d_code_off in dwords): the pair-major
// token array of the diagonal engine, and the bin image of the systolic engine / int32 kernels.
hipError_t swg_launch_build_tokens(const uint32_t *d_codes, const uint64_t *d_code_off, const uint32_t *d_lens,
                                   const uint32_t *d_pair_off, uint32_t n_pairs, uint64_t total_blocks, uint4 *d_tok,
                                   hipStream_t stream);
// Pair tokens straight from reference-shaped batches ([max_len][16] int8 table indices, two adjacent lanes = one
// pair): pair p reads its X residue of row j at stage[pair_src[p] + 16 j], its Y residue one byte further (bit 63
// of pair_src: no Y, an odd lane count); *d_bad becomes non-zero if an index is outside 1..31.
hipError_t swg_launch_build_tokens16(const uint8_t *d_stage, const uint64_t *d_pair_src, const uint32_t *d_pair_len,
                                     const uint32_t *d_pair_off, uint32_t n_pairs, uint64_t total_blocks, uint4 *d_tok,
                                     uint32_t *d_bad, hipStream_t stream);
hipError_t swg_launch_build_bins(const uint32_t *d_codes, const uint64_t *d_code_off, const uint32_t *d_lens,
                                 const uint64_t *d_bin_off, const uint32_t *d_bin_nblk, uint32_t n_bins,
                                 uint32_t *d_packed, hipStream_t stream);

// Appends every slot id whose 16-bit score saturated (>= ceiling: 32767, 65535 in the wide form, 4096 for the
// packed-f16 cells) to list.
hipError_t swg_launch_zero2(void *a, size_t a_bytes, void *b, size_t b_bytes, hipStream_t stream);
// (pairs first_pair .. n_pairs-1 of the sorted order: the ones that ran on the f16 cells)
hipError_t swg_launch_collect_flagged_pairs(const int32_t *d_scores, uint32_t first_pair, uint32_t n_pairs, int32_t ceiling,
                                            uint32_t *d_list, uint32_t *d_count, uint32_t *d_seqs, const uint32_t *d_lens,
                                            uint32_t *d_rows16, hipStream_t stream);
// d_lens / d_rows16 (or NULL): also adds up the flagged sequences' lengths, in units of 16 rows.
hipError_t swg_launch_collect_saturated(const int32_t *d_scores, uint32_t n_slots, int32_t ceiling,
                                        uint32_t *d_list, uint32_t *d_count, const uint32_t *d_lens, uint32_t *d_rows16,
                                        hipStream_t stream);

// Device top-K (see swg_kernels.hip): after the call d_thr[1] == 0 means d_cand[0..min(*d_count,cap))
// holds every hit with score >= d_thr[0] as 64-bit keys (unsorted); otherwise fall back.
#define SWG_TOPK_CAND_CAP 8192u
hipError_t swg_launch_topk(const int32_t *d_scores, const uint32_t *d_order, uint32_t n_slots, uint32_t k,
                           uint32_t *d_hist, uint32_t *d_thr, uint64_t *d_cand, uint32_t cap, uint32_t *d_count,
                           hipStream_t stream);
// ... and for the n_queries score rows of a batch (swg_search_multi) at once: d_hist[n_queries][4096],
// d_meta[n_queries][4] = {threshold, status, candidate count, -}, d_cand[n_queries][cap]
#define SWG_TOPK_MULTI_CAP 1024u
hipError_t swg_launch_topk_multi(const int32_t *d_scores, uint64_t score_stride, const uint32_t *d_order, uint32_t n_slots,
                                 uint32_t n_queries, uint32_t k, uint32_t *d_hist, uint32_t *d_meta, uint64_t *d_cand, uint32_t cap,
                                 hipStream_t stream);
