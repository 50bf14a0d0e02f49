// swg_diag_host.cpp -- host side of the diagonal engine: geometry planning, the
// pair-major token array of the work queue, and the stream layout of the fixed-stream
// form (which pair of sequences runs in which lane group, in what order).  Host-only C++ (OpenMP); no GPU needed.
#include "swg_host_internal.h"
#include "../../include/swg_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>

// Cycles one SIMD needs per packed-int16 / DPP / v_perm wave-instruction when `wps` waves share
// it.  One wave: the microbenchmark (profiles/r01_valu_issue_rates.txt).  Two to four: from the
// fill kernel itself on a uniform database (tools/sweeps/occ.sh): three waves per SIMD are as good
// as four, two cost 5 %; the microbenchmark's 5.3 / 5.05 / 4.56 were pessimistic.
static const double kCyclesPerInstr[5] = {0.0, 6.8, 4.55, 4.27, 4.35};
// ... and per instruction of a raised-priority wavefront beside three others (traced: the long
// class advances a 4-row block of 64 x 6 columns, 296 instructions, every 1.6 us)
static const double kHotCycles = 13.0;

uint64_t swg_db_pair_count(const swg_db *db) { return (db->n_local + 1) / 2; }

// Rows of the pairs [pair_begin, pair_end) (two reset rows + the longer sequence's residues each) and the longest of
// them.  The planner asks on every search, for prefixes of the sorted order: a prefix sum, built on first use (8 bytes per
// pair), makes that O(1) -- the loop it replaces walked 5 million pairs per search of the 10M-sequence database and, for
// 2 million peptides, cost more than the fill (round 4: step 1.39 ms for a 0.97 ms fill).
uint64_t swg_db_pair_rows(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint64_t *longest_rows)
{
    const uint64_t n = swg_db_pair_count(db);
    std::vector<uint64_t> &pre = const_cast<swg_db *>(db)->pair_rows_prefix;
    if (pre.size() != n + 1) {
        pre.assign(n + 1, 0);
        for (uint64_t p = 0; p < n; ++p) pre[p + 1] = pre[p] + 2ull + db->lens[2 * p];
    }
    pair_end = std::min(pair_end, n);
    if (pair_begin >= pair_end) {
        if (longest_rows) *longest_rows = 0;
        return 0;
    }
    if (longest_rows) *longest_rows = 2ull + db->lens[2 * pair_begin]; // (sorted longest first)
    return pre[pair_end] - pre[pair_begin];
}

uint64_t swg_db_pairs_longer_than(const swg_db *db, uint64_t rows)
{
    // pairs are sorted by length, longest first: binary search the prefix
    uint64_t lo = 0, hi = swg_db_pair_count(db);
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (2ull + db->lens[2 * mid] > rows) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Cost model (cycles of one SIMD per packed/DPP wave-instruction, measured):
//   a lane group spends (10*K + overhead) instructions per database row; a SIMD shared by
//   `wps` waves issues one such instruction every kCyclesPerInstr[wps] cycles, so one wave
//   advances a row every instr * kCyclesPerInstr[wps] * wps cycles.  A search lasts as long
//   as the larger of (a) all work divided by all SIMDs and (b) the longest chain of rows any
//   one lane group has to walk.  The few longest pairs can be split off into their own class
//   (64 lanes per pair, fewest columns per lane, raised wave priority) so that (b) does not
//   dominate small databases.
// 10 per column pair plus the row's bookkeeping: 14 instructions, but worth about 30 issue slots
// (DPP wait states, the wait for the row's first profile read) by the K = 16 / 24 / 32 comparison
// (the packed-f16 cells, form 2, take 8.5 per column pair: DESIGN 4.1)
static double instr_per_row(int K, int G, int form = 0) { return (form == 2 ? 8.5 : 10.0) * K + (G == 32 ? 34.0 : 30.0); }
// one pass of several through the work queue: row index, edge hand-over to the leader, the tail's
// edge store
static const double kEdgeInstr = 7.0;

// Long class: the cheapest geometry (instructions per pair-row, column padding included)
// whose longest chain still finishes within `budget_cycles`; if none does, the shortest chain.
long g_swg_long_cols = 0;  // experiment switch: restrict the long class to this K (0 = free)
long g_swg_long_group = 0; // experiment switch: restrict the long class to this G (0 = free)

static bool long_class_geometry(size_t lq, uint64_t longest_rows, double budget_cycles, bool dynamic, SwgDiagPlan *lp, int form)
{
    bool ok = false, ok_fit = false;
    double best_cost = 1e300, best_depth = 1e300;
    SwgDiagPlan fit, fast;
    const int groups[2] = {32, 64};
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const SwgKernelInfo info = swg_diag_variant_info(v);
        if (g_swg_long_cols > 0 && info.K != (int)g_swg_long_cols) continue;
        // with static streams the 2-column-chunk instantiations (K % 4 == 2) measured slower as the
        // long class than the next multiple of 4 despite fewer instructions: only on request there.
        // With the work queue they win (64 x 6 = 384 columns for a 367-column query: no padding
        // beyond the bulk's).
        if (!dynamic && g_swg_long_cols == 0 && info.K % 4 != 0) continue;
        for (int gi = 0; gi < 2; ++gi) {
            const int G = groups[gi];
            if (g_swg_long_group > 0 && G != (int)g_swg_long_group) continue;
            const size_t cols = (size_t)G * info.K;
            if ((size_t)G * swg_diag_padded_cols(info.K) * 64 > 160 * 1024) continue;
            const int npass = (int)((lq + cols - 1) / cols);
            if (dynamic && npass > 1) continue; // the queue serves single-pass classes only
            const double instr = npass * instr_per_row(info.K, G, form);
            const double cost = instr / (64 / G);                     // per pair-row
            const double depth = ((double)longest_rows + G) * instr * kHotCycles;
            SwgDiagPlan c;
            c.variant = v;
            c.K = info.K;
            c.G = G;
            c.npass = npass;
            c.W = 4;
            c.lds_bytes = (size_t)G * swg_diag_padded_cols(info.K) * 64;
            if (depth <= budget_cycles && cost < best_cost) {
                best_cost = cost;
                fit = c;
                ok_fit = true;
            }
            if (depth < best_depth) {
                best_depth = depth;
                fast = c;
                ok = true;
            }
        }
    }
    if (ok_fit) *lp = fit; else if (ok) *lp = fast;
    return ok;
}

int swg_plan_diag_candidates(const swg_db *db, size_t lq, int n_cu, long opt_cols, long opt_group, long opt_waves,
                             long opt_long_split, bool allow_split, bool work_queue, std::vector<SwgDiagWork> *cands,
                             double copies, int form)
{
    // copies > 1: the same database is searched by that many queries in ONE launch (swg_search_multi):
    // all throughput terms grow with it, the longest chain of rows does not
    cands->clear();
    const uint64_t n_pairs = swg_db_pair_count(db);
    if (n_pairs == 0) return 0;
    if (copies < 1.0) copies = 1.0;
    uint64_t longest = 0;
    const uint64_t rows_once = swg_db_pair_rows(db, 0, n_pairs, &longest);
    const double rows_all = (double)rows_once * copies;
    const bool have_long = allow_split && opt_long_split >= 0;
    const double simds = 4.0 * n_cu;
    const int groups[3] = {16, 32, 64};
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const SwgKernelInfo info = swg_diag_variant_info(v);
        if (opt_cols > 0 && info.K != (int)opt_cols) continue;
        for (int gi = 0; gi < 3; ++gi) {
            const int G = groups[gi];
            if (opt_group > 0 && G != (int)opt_group) continue;
            const size_t cols = (size_t)G * info.K;
            const size_t lds = (size_t)G * swg_diag_padded_cols(info.K) * 64;
            if (lds > 160 * 1024) continue;
            const int npass = (int)((lq + cols - 1) / cols);
            const int NG = 64 / G;
            // with the work queue (several passes: one launch per pass) there are no fixed shares, the
            // chain that matters is the longest pair at the rate of a wavefront that gets its fair
            // share of the SIMD
            const bool dynamic = work_queue && (npass == 1 || db->n_local < (1u << 30));
            const double instr = instr_per_row(info.K, G, form) + (dynamic && npass > 1 ? kEdgeInstr : 0.0);
            for (int wps = 1; wps <= 4; ++wps) {
                const int W = 4 * wps;
                if (W > info.max_waves) continue;
                if (opt_waves > 0 && W != (int)opt_waves) continue;
                // with the work queue the workgroup size does not matter for balance; four wavefronts
                // (one per SIMD) interleave the two classes on every CU, larger workgroups partition the
                // CUs between them (timed faster in isolation, slower back to back).  Larger ones only
                // where LDS allows a single workgroup per CU and more wavefronts mean more occupancy.
                if (dynamic && opt_waves == 0 && wps != 1) {
                    // (occupancy by the workgroup's real LDS size, lane-group records included: at exactly two
                    // profiles per CU they decide whether a second workgroup fits)
                    auto occupancy = [&](int w) {
                        const size_t l = swg_diag_dyn_lds_bytes(info.K, G, 4 * w);
                        return std::min(4, w * std::max(1, std::min<int>(info.max_waves / (4 * w), (int)((160 * 1024) / l))));
                    };
                    bool improves = true;
                    for (int w2 = 1; w2 < wps; ++w2)
                        if (occupancy(w2) >= occupancy(wps)) improves = false;
                    if (!improves) continue;
                }
                // (the work-queue kernels keep a 512-byte record per lane group behind the profile)
                const size_t lds_wg = dynamic ? swg_diag_dyn_lds_bytes(info.K, G, W) : lds;
                const int per_cu = std::max(1, std::min<int>(info.max_waves / W, (int)((160 * 1024) / lds_wg)));
                const int eff_wps = std::min(4, wps * per_cu);
                const uint64_t spw = (uint64_t)W * NG;
                const uint64_t hw_streams = (uint64_t)n_cu * per_cu * spw;
                // (a long class beside a multi-pass queue launch would need its own edge buffers: not built)
                for (int split = 0; split <= (have_long && !(dynamic && npass > 1) ? 3 : 0); ++split) {
                    uint64_t n_long = 0, longest_bulk = longest, longest_long = 0;
                    double rows_long = 0;
                    uint64_t streams0 = std::max<uint64_t>(1, std::min<uint64_t>(hw_streams, (uint64_t)((double)n_pairs * copies)));
                    streams0 = (streams0 + spw - 1) / spw * spw;
                    if (split) {
                        if (split == 3 && opt_long_split > 0) continue;
                        // experiment switch long_group=16: only the bulk's own geometry as the long class
                        if (g_swg_long_group == 16 && split != 2) continue;
                        if (g_swg_long_group != 16 && g_swg_long_group > 0 && split == 2) continue;
                        const double frac = split == 3 ? 0.6 : 0.33;
                        uint64_t thr = opt_long_split > 0 ? (uint64_t)opt_long_split
                                                          : (uint64_t)(frac * rows_all / (double)streams0);
                        if (dynamic && opt_long_split <= 0) {
                            // longest pair a fair-share wavefront finishes within the whole search
                            const double cps0 = kCyclesPerInstr[eff_wps]; // (a database that needs a long class fills its slots)
                            const double all_cycles = rows_all / NG * instr * cps0 / simds;
                            thr = (uint64_t)((split == 3 ? 0.65 : 0.9) * all_cycles / (instr * cps0 * eff_wps));
                        }
                        thr = std::max<uint64_t>(thr, 64);
                        if (longest <= thr) continue;
                        n_long = swg_db_pairs_longer_than(db, thr);
                        if (n_long == 0 || n_long * 4 > n_pairs) continue;
                        rows_long = (double)swg_db_pair_rows(db, 0, n_long, &longest_long) * copies;
                        longest_bulk = 2ull + db->lens[2 * n_long];
                    }
                    const uint64_t n_bulk = n_pairs - n_long;
                    const double rows_bulk = rows_all - rows_long;
                    uint64_t streams = std::max<uint64_t>(1, std::min<uint64_t>(hw_streams, (uint64_t)((double)n_bulk * copies)));
                    streams = (streams + spw - 1) / spw * spw;
                    // A database of few pairs does not fill the wave slots the geometry allows: the chain's wavefront
                    // shares its SIMD with as many others as the launch really has (a 300 000-row sequence among 400
                    // short ones, lq 400: 64 lanes x 17 columns ranked first -- "one wavefront per SIMD" by its LDS
                    // size, where 64 x 7 has the SIMD to itself just the same at half the instructions per row).
                    const int eff = std::max(1, std::min(eff_wps, (int)std::ceil((double)(streams / spw) * W / simds)));
                    const double cps = kCyclesPerInstr[eff];
                    // (a) throughput: SIMD-cycles of both classes over all SIMDs
                    double work = rows_bulk / NG * npass * instr * cps;
                    double crit = (std::max<double>(rows_bulk / streams, (double)longest_bulk) + G) * npass *
                                  instr * cps * eff;
                    uint64_t lstreams = 0;
                    SwgDiagPlan lp;
                    if (split) {
                        // the long pairs have to be done by the time the bulk is
                        const double bulk_cycles = std::max(work / simds, crit);
                        if (split == 2) {
                            // the bulk's own geometry (no extra column padding), own streams, raised priority
                            lp.variant = v;
                            lp.K = info.K;
                            lp.G = G;
                            lp.npass = npass;
                            lp.W = 4;
                            lp.lds_bytes = lds;
                        } else if (!long_class_geometry(lq, longest_long, 0.8 * bulk_cycles, dynamic, &lp, form)) {
                            continue;
                        }
                        const uint64_t lspw = 4ull * (64 / lp.G);
                        // static: two pairs per stream to balance; queue: a wavefront on every SIMD
                        lstreams = std::max<uint64_t>(1, std::min<uint64_t>(dynamic ? (uint64_t)((double)n_long * copies) : (n_long + 1) / 2,
                                                                            (uint64_t)n_cu * lspw));
                        lstreams = (lstreams + lspw - 1) / lspw * lspw;
                        const double linstr = instr_per_row(lp.K, lp.G, form);
                        work += rows_long / (64 / lp.G) * lp.npass * linstr * cps;
                        const double lcrit = (std::max<double>(rows_long / lstreams, (double)longest_long) + lp.G) *
                                             lp.npass * linstr * kHotCycles;
                        crit = std::max(crit, lcrit);
                    }
                    double cycles = std::max(work / simds, crit);
                    if (dynamic)
                        cycles += (npass - 1) * 0.15e-3 * 2.35e9; // a launch per pass: drain and ramp
                    else
                        cycles *= 1.0 + 0.04 * (npass - 1); // profile reloads, pass barriers, edge spills
                    // fixed streams lose what the work queue was built to recover (uneven wavefront
                    // rates, a thinning tail): measured 4 900 against 5 600 GCUPS on config 2
                    if (!dynamic) cycles *= 1.15;
                    // Workgroups of more than four wavefronts are candidates only where they raise the
                    // occupancy (above), and then they deliver it -- round 2, 200 000 sequences: lq 800 7 030
                    // against 6 750 GCUPS, lq 1000 7 060 against 6 820, lq 2500 (3 passes) 6 800 against 6 450
                    // with 12 wavefronts instead of 4 -- so they carry no penalty of their own (round 1 had one).
                    // Where a third wavefront per SIMD hurts, it is through the longest pair's chain, which
                    // `crit` prices (config 5's 8 200-row near-copies: 6 070 with 12 against 6 590 with 8).
                    // a second class is a second launch that takes LDS and issue slots from the bulk: worth
                    // it where the chain decides, not for a tie (lq 600: 6 420 with 19 long pairs, 6 880 without)
                    if (split) cycles *= 1.02;
                    // (where a chain decides, the plans it ties are told apart by their throughput time)
                    const double ms = (cycles + 1e-3 * work / simds) / 2.35e9 * 1e3;
                    {
                        SwgDiagWork one;
                        SwgDiagWork *wk = &one;
                        SwgDiagPlan &b = wk->plan[0];
                        b.variant = v;
                        b.K = info.K;
                        b.G = G;
                        b.npass = npass;
                        b.W = W;
                        b.n_streams = (uint32_t)streams;
                        b.workgroups = (int)(streams / spw);
                        b.lds_bytes = lds;
                        b.est_ms = ms;
                        wk->pair_begin[0] = n_long;
                        wk->pair_end[0] = n_pairs;
                        wk->n_classes = 1;
                        if (split) {
                            wk->n_classes = 2;
                            wk->plan[1] = lp;
                            wk->plan[1].n_streams = (uint32_t)lstreams;
                            wk->plan[1].workgroups = (int)(lstreams / (4ull * (64 / lp.G)));
                            wk->plan[1].est_ms = ms;
                            wk->pair_begin[1] = 0;
                            wk->pair_end[1] = n_long;
                        }
                        cands->push_back(one);
                    }
                }
            }
        }
    }
    // (a positive long_split is a request: plans with the class it asks for rank before those without, whatever
    // the model thinks of them -- when no pair is that long there are none, and the rest is ranked as usual)
    const bool want_long = have_long && opt_long_split > 0;
    std::stable_sort(cands->begin(), cands->end(), [want_long](const SwgDiagWork &a, const SwgDiagWork &b) {
        const bool la = want_long && a.n_classes == 2, lb = want_long && b.n_classes == 2;
        return la != lb ? la : a.plan[0].est_ms < b.plan[0].est_ms;
    });
    return (int)cands->size();
}

int swg_plan_diag_work(const swg_db *db, size_t lq, int n_cu, long opt_cols, long opt_group, long opt_waves,
                       long opt_long_split, bool allow_split, bool work_queue, SwgDiagWork *wk, double copies, int form)
{
    std::vector<SwgDiagWork> c;
    wk->n_classes = 0;
    if (swg_plan_diag_candidates(db, lq, n_cu, opt_cols, opt_group, opt_waves, opt_long_split, allow_split, work_queue,
                                 &c, copies, form) > 0)
        *wk = c[0];
    return wk->n_classes;
}

// Tokens of one pair of sequences (two reset rows, then one row per residue of the longer
// one, the last block filled up with padding rows), one 32-bit token per row: byte 0 = X residue
// byte, byte 1 = Y residue byte, bit 16 = reset row, bit 17 = last row; returns the number of 4-row
// blocks.  The device builds the same image from the resident residue bytes
// (swg_build_tokens_kernel); this is its host restatement, used by the fixed-stream layout and by
// the tests that compare the two.
static const uint32_t kTokReset = SWG_TOK_RESET, kTokLast = SWG_TOK_LAST;
static uint64_t write_pair_tokens(const swg_db *db, size_t p, uint32_t *t)
{
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
    const uint32_t lx = db->lens[2 * p];
    const bool has_y = 2 * p + 1 < n_slots && db->order[2 * p + 1] != 0xFFFFFFFFu;
    const uint32_t ly = has_y ? db->lens[2 * p + 1] : 0;
    const uint8_t *cx = db->codes.data() + db->code_off[2 * p];
    const uint8_t *cy = has_y ? db->codes.data() + db->code_off[2 * p + 1] : nullptr;
    t[0] = kTokReset; // reset rows: padding residue for both sequences
    t[1] = kTokReset | SWG_TOK_RESET2;
    uint32_t *r = t + 2;
    const uint32_t both = std::min(lx, ly); // (= ly: sorted order)
    for (uint32_t j = 0; j < both; ++j) r[j] = (uint32_t)cx[j] | (uint32_t)cy[j] << 8;
    for (uint32_t j = both; j < lx; ++j) r[j] = cx[j];
    const uint64_t blocks = (2ull + lx + 3) / 4;
    for (uint64_t j = 2ull + lx; j < blocks * 4; ++j) t[j] = 0; // rest of the last block: padding rows
    // last row of the pair: X's last residue; for an empty pair the second reset row, so that every
    // pair is finished by the tail lane exactly once
    t[lx + 1] |= kTokLast;
    return blocks;
}

void swg_build_diag_layout(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint32_t n_streams,
                           uint32_t streams_per_wg, SwgDiagLayout *L)
{
    const size_t n_pairs = (size_t)(pair_end - pair_begin);
    L->pair_begin = pair_begin;
    L->pair_end = pair_end;
    L->n_streams = n_streams;
    // longest-processing-time-first: pairs are already sorted by length (descending)
    std::vector<uint32_t> owner(n_pairs);
    std::vector<uint64_t> load(n_streams, 0);
    std::vector<uint32_t> count(n_streams, 0);
    {
        typedef std::pair<uint64_t, uint32_t> item; // (blocks so far, stream)
        std::priority_queue<item, std::vector<item>, std::greater<item>> heap;
        for (uint32_t s = 0; s < n_streams; ++s) heap.push(item(0, s));
        // Heap slot h fills up in longest-first order, so slots 0,1,2,.. hold the longest
        // pairs.  Physical stream = lane group of a workgroup: deal the slots round-robin
        // over the workgroups so every CU gets its share of long streams instead of one CU
        // getting all of them.
        const uint32_t spw = std::max<uint32_t>(1, streams_per_wg);
        const uint32_t n_wgs = (n_streams + spw - 1) / spw;
        auto phys = [&](uint32_t h) -> uint32_t {
            const uint32_t s = (h % n_wgs) * spw + (h / n_wgs);
            return s < n_streams ? s : h; // (n_streams is a multiple of spw in practice)
        };
        for (size_t q = 0; q < n_pairs; ++q) {
            const size_t p = pair_begin + q;
            const uint64_t blocks = (2ull + db->lens[2 * p] + 3) / 4;
            item it = heap.top();
            heap.pop();
            const uint32_t s = phys(it.second);
            owner[q] = s;
            load[s] = it.first + blocks;
            count[s]++;
            heap.push(item(it.first + blocks, it.second));
        }
    }
    L->stream_off.assign((size_t)n_streams + 1, 0);
    L->stream_pair_off.assign((size_t)n_streams + 1, 0);
    for (uint32_t s = 0; s < n_streams; ++s) {
        L->stream_off[s + 1] = L->stream_off[s] + load[s];
        L->stream_pair_off[s + 1] = L->stream_pair_off[s] + count[s];
    }
    L->total_blocks = L->stream_off[n_streams];
    L->max_stream_blocks = n_streams ? *std::max_element(load.begin(), load.end()) : 0;
    L->stream_pairs.assign(n_pairs, 0);
    {
        std::vector<uint32_t> fill(n_streams, 0);
        for (size_t q = 0; q < n_pairs; ++q) {
            const uint32_t s = owner[q];
            L->stream_pairs[L->stream_pair_off[s] + fill[s]++] = (uint32_t)(pair_begin + q);
        }
    }
    uint64_t rows_total = 0;
    for (size_t p = pair_begin; p < pair_end; ++p) rows_total += 2ull + db->lens[2 * p];
    L->pair_rows_total = rows_total;
    L->tok.assign(L->total_blocks * 4, 0u);
    uint32_t *base = L->tok.data(); // one 32-bit token per row
#pragma omp parallel for schedule(dynamic, 16) num_threads(swg_host_threads())
    for (long long s = 0; s < (long long)n_streams; ++s) {
        uint32_t *t = base + L->stream_off[s] * 4;
        for (uint32_t i = L->stream_pair_off[s]; i < L->stream_pair_off[s + 1]; ++i)
            t += write_pair_tokens(db, L->stream_pairs[i], t) * 4; // rest of the last block stays padding
    }
}

int swg_build_pair_tokens(const swg_db *db, std::unique_ptr<uint32_t[]> *tok, size_t *tok_dwords,
                          std::vector<uint32_t> *pair_off)
{
    if (tok_dwords) *tok_dwords = 0;
    const uint64_t n_pairs = swg_db_pair_count(db);
    if (n_pairs >= (1ull << 31)) return -1;
    pair_off->assign((size_t)n_pairs + 1, 0u);
    uint64_t total = 0;
    for (uint64_t p = 0; p < n_pairs; ++p) {
        total += (2ull + db->lens[2 * p] + 3) / 4;
        if (total + 1 >= (1ull << 32)) return -1; // block offsets are 32-bit on the device (and one block of zeros follows the last pair)
        (*pair_off)[p + 1] = (uint32_t)total;
    }
    if (!tok) return 0; // offsets only: the tokens themselves are built on the device
    // every pair writes all rows of its blocks, so the buffer needs no zero fill: its pages are
    // first touched by the threads that fill them
    tok->reset(new uint32_t[std::max<size_t>(4, (size_t)total * 4)]);
    *tok_dwords = (size_t)total * 4;
    uint32_t *base = tok->get();
#pragma omp parallel for schedule(dynamic, 64) num_threads(swg_host_threads())
    for (long long p = 0; p < (long long)n_pairs; ++p)
        write_pair_tokens(db, (size_t)p, base + (size_t)(*pair_off)[p] * 4);
    return 0;
}

// Several passes of G*K columns each leave the last one partly empty (3000 columns in 6 passes of 512: 72 of them, 2.3 %
// of the work): it runs the instantiation with the fewest columns per lane that cover what is left.  Returns false when
// there is nothing to gain (one pass; the rest needs all K columns; no instantiation takes the plan's workgroup size).
bool swg_plan_last_pass(const SwgDiagPlan &pl, size_t lq, int *variant, int *K)
{
    if (pl.npass < 2 || pl.G <= 0 || pl.K <= 0 || lq <= (size_t)(pl.npass - 1) * pl.G * pl.K) return false;
    const size_t rest = lq - (size_t)(pl.npass - 1) * pl.G * pl.K;
    const int need = (int)((rest + pl.G - 1) / pl.G);
    int best = -1, bestK = pl.K;
    for (int v = 0; v < swg_num_diag_variants(); ++v) {
        const SwgKernelInfo info = swg_diag_variant_info(v);
        if (info.K >= need && info.K < bestK && pl.W <= info.max_waves) best = v, bestK = info.K;
    }
    if (best < 0) return false;
    *variant = best;
    *K = bestK;
    return true;
}

// Both 16-bit forms in one search (swg_search_begin): the length from which a sequence can reach the f16 cells' ceiling
// as an exact copy of a stretch of the query -- such a copy scores qbound / lq per row on average, qbound being the
// query's best possible total --, and where that length cuts the sorted pair order (kept per database and length).
uint32_t swg_split_rows(size_t lq, uint64_t qbound)
{
    if (qbound == 0) return 0u;
    return (uint32_t)std::min<uint64_t>((4096ull * lq + qbound - 1) / qbound, 1u << 30);
}

void swg_db_split_at(swg_db *db, uint32_t rows)
{
    if (db->split_rows == rows) return;
    const size_t n = db->lens.size();
    const size_t first_short = (size_t)(std::partition_point(db->lens.begin(), db->lens.end(), [rows](uint32_t l) { return l >= rows; }) -
                                        db->lens.begin());
    db->split_rows = rows;
    db->split_pair = (uint32_t)((first_short + 1) / 2); // (a pair with one long member is a long pair)
    uint64_t sum = 0;
    for (size_t i = std::min(n, (size_t)db->split_pair * 2); i < n; ++i) sum += db->lens[i];
    db->split_residues = sum;
}

// Test hook (not part of the public ABI): the both-forms cut of a packed database for a query of lq columns whose best
// possible total is qbound.  out[0..2] = rows, first pair of the f16 part, residues of the f16 part.
extern "C" int swg_debug_split(swg_db *db, size_t lq, uint64_t qbound, uint64_t *out)
{
    if (!db || !out || lq == 0 || qbound == 0) return SWG_ERR_ARG;
    const uint32_t rows = swg_split_rows(lq, qbound);
    swg_db_split_at(db, rows);
    out[0] = rows;
    out[1] = db->split_pair;
    out[2] = db->split_residues;
    return SWG_OK;
}

// What the systolic engine (swg_fill_kernel: lane l owns sequences 2l, 2l+1 of a 128-sequence bin, W wavefronts of a
// workgroup chained over the query, int16 cells) would need for this database, by the host's bin table: a bin costs its
// LONGEST member's rows (rounded to 4-row blocks) on every one of the W wavefronts, 10 K + 12 instructions per row, plus
// ~5 000 cycles per bin and wavefront for taking it off the work counter; and the search cannot end before the longest
// bin's chain has, at the rate of a wavefront that shares its SIMD.  Calibrated on 2 M peptides (round 4,
// profiles/r04_peptides_systolic.txt: lq 128 K 32 x 4 wavefronts 1.15 ms measured / 1.21 estimated, lq 30 K 16 x 2 0.37 /
// 0.33) and on config 2, where the 5 000-row bin's chain is the whole search (11.8 ms measured, 11 estimated).
// Returns the estimate in ms and the columns per wavefront of the best single-pass instantiation (0: none).
double swg_systolic_estimate_ms(const swg_db *db, size_t lq, int n_cu, int *best_K, bool f16)
{
    *best_K = 0;
    if (db->n_bins == 0 || db->bin_nblk.size() != db->n_bins) return 1e300;
    uint64_t blocks = 0;
    for (uint32_t b : db->bin_nblk) blocks += b;
    const double rows = 4.0 * (double)blocks, longest = 4.0 * (double)db->max_nblk;
    double best = 1e300;
    for (int v = 0; v < swg_num_variants(16); ++v) {
        const SwgKernelInfo info = swg_variant_info(16, v);
        const int W = (int)((lq + (size_t)info.K - 1) / (size_t)info.K);
        if (W > info.max_waves) continue; // (several passes: never the better engine)
        if (info.K > 32) continue;         // (the 48-column instantiation runs at two wavefronts per SIMD: measured 45 % over its count)
        const size_t lds = info.lds_per_wave * (size_t)W + info.lds_fixed;
        const int per_cu = std::max(1, std::min<int>(info.max_waves / W, (int)((160 * 1024) / lds)));
        const double waves_per_simd = std::max(1.0, per_cu * W / 4.0);
        const double instr = (f16 ? 8.5 : 10.0) * info.K + 12.0; // (packed-f16 cells where no score can reach 4096)
        const double thr = (rows * W * instr * 4.06 + (double)db->n_bins * W * 5000.0) / (4.0 * n_cu);
        // bins are whole work units of a workgroup: with few bins per workgroup the search lasts as many ROUNDS as the
        // busiest workgroup has bins -- the first of them the longest bin (bins go out longest first), the others about
        // average -- at the rate of a wavefront that shares its SIMD (300 000 sequences of ~250 residues, lq 64: 2 344
        // bins on 1 536 workgroups are two rounds, 0.97 ms measured where the throughput term alone says 0.58)
        const double wps = std::min(4.0, waves_per_simd);
        const double n_wgs = std::max(1.0, (double)n_cu * per_cu);
        const double rounds = std::ceil((double)db->n_bins / n_wgs);
        const double quant = (longest + (rounds - 1.0) * rows / (double)db->n_bins) * instr * 4.06 * wps;
        const double chain = std::max(longest * instr * 4.06 * wps, quant);
        // (the 24-column instantiation measures 30 % over its count -- peptides lq 128: 1.48 ms against 1.05 for 8 x 16 columns --
        // and is only picked where that still wins)
        const double ms = std::max(thr, chain) / 2.4e6 * (info.K == 24 ? 1.3 : 1.0);
        if (ms < best) best = ms, *best_K = info.K;
    }
    return best;
}

// The planner's estimate for the lane groups was fitted on BASELINE's length distribution (pairs of ~380 rows), where
// about 40 % of a wavefront's steps find one of its 64 lanes on a flagged row (a pair's two reset rows and its last row)
// and take the rare branch; its cost there is inside the fitted intercept.  A database of SHORT pairs takes that branch
// on nearly every step: 19 instructions more per wavefront-row than the fit knows, scaled by how much more often
// (peptides: K = 8, 98 fitted instructions per row, measured 1.56 ms where the unadjusted estimate says 1.41).  Used where
// the two engines are compared, not in the ranking of geometries (the term is the same for all of them).
double swg_diag_short_pair_factor(const swg_db *db, const SwgDiagPlan &pl, int form)
{
    const uint64_t n_pairs = swg_db_pair_count(db);
    if (n_pairs == 0) return 1.0;
    uint64_t longest = 0;
    const double L = std::max(4.0, (double)swg_db_pair_rows(db, 0, n_pairs, &longest) / (double)n_pairs);
    auto p_flag = [](double rows) { return 1.0 - std::pow(std::max(0.0, 1.0 - 3.0 / rows), 64.0); };
    const double extra = 19.0 * std::max(0.0, p_flag(L) - p_flag(380.0));
    return 1.0 + extra / instr_per_row(pl.K, pl.G, form);
}

// test hook: both engines' estimates for a database and a query length (cells: 0 int16, 2 packed f16), without a device:
// out[0] lane groups (us), out[1] systolic (us), out[2] its columns per wavefront, out[3] 1 = the model picks the systolic engine
extern "C" int swg_debug_engine(const swg_db *db, size_t lq, int n_cu, int form, int32_t *out)
{
    if (!db || !out || lq == 0 || n_cu <= 0) return SWG_ERR_ARG;
    SwgDiagWork wk;
    try {
        if (swg_plan_diag_work(db, lq, n_cu, 0, 0, 0, 0, true, true, &wk, 1.0, form) <= 0) return SWG_ERR_ARG;
    } catch (const std::exception &) {
        return SWG_ERR_NOMEM;
    }
    int K = 0;
    // (the hook has no scoring system: the f16 cells are assumed where BLOSUM62's largest entry, 11, keeps every score below 4096)
    const bool f16 = form == 2 && (uint64_t)db->max_nblk * SWG_ROWS_PER_BLK * 11ull < 4096ull;
    const double sys = swg_systolic_estimate_ms(db, lq, n_cu, &K, f16);
    const double diag = wk.plan[0].est_ms * swg_diag_short_pair_factor(db, wk.plan[0], form);
    out[0] = (int32_t)(diag * 1e3);
    out[1] = K > 0 ? (int32_t)std::min(sys * 1e3, 2.0e9) : -1;
    out[2] = K;
    out[3] = K > 0 && sys < SWG_SYSTOLIC_MARGIN * diag ? 1 : 0;
    return SWG_OK;
}

// Test hook (not part of the public ABI, declared in swg_host_internal.h): the cost model's first choice
// for a packed database and a query length on a device of n_cu compute units, without a device.
// out[0..12] = classes, K, G, W, passes, workgroups, long pairs, long K, long G, long W, long workgroups,
// estimated microseconds, columns per lane of the last pass (0: as the other passes).
extern "C" int swg_debug_plan(const swg_db *db, size_t lq, int n_cu, int32_t *out)
{
    if (!db || !out || lq == 0 || n_cu <= 0) return SWG_ERR_ARG;
    SwgDiagWork wk;
    try {
        if (swg_plan_diag_work(db, lq, n_cu, 0, 0, 0, 0, true, true, &wk) <= 0) return SWG_ERR_ARG;
    } catch (const std::exception &) {
        return SWG_ERR_NOMEM;
    }
    const SwgDiagPlan &b = wk.plan[0], &l = wk.plan[1];
    const bool two = wk.n_classes == 2;
    int last_variant = -1, last_K = 0;
    if (two || !swg_plan_last_pass(b, lq, &last_variant, &last_K)) last_K = 0;
    const int32_t v[13] = {wk.n_classes, b.K, b.G, b.W, b.npass, b.workgroups, two ? (int32_t)(wk.pair_end[1] - wk.pair_begin[1]) : 0,
                           two ? l.K : 0, two ? l.G : 0, two ? l.W : 0, two ? l.workgroups : 0, (int32_t)(b.est_ms * 1e3), last_K};
    memcpy(out, v, sizeof v);
    return SWG_OK;
}
