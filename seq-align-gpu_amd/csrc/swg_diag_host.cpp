// swg_diag_host.cpp -- host side of the diagonal engine: geometry planning and
// the stream layout (which pair of sequences runs in which lane group, in what
// order).  Host-only C++ (OpenMP); no GPU needed.
#include "swg_host_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>

// Measured on MI355X (profiles/r01_valu_issue_rates.txt): cycles one SIMD needs per
// packed-int16 / DPP / v_perm wave-instruction when `wps` waves share it.
static const double kCyclesPerInstr[5] = {0.0, 6.8, 5.1, 4.8, 4.56};

uint64_t swg_db_pair_count(const swg_db *db) { return (db->n_local + 1) / 2; }

uint64_t swg_db_pair_rows(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint64_t *longest_rows)
{
    uint64_t total = 0, longest = 0;
    for (uint64_t p = pair_begin; p < pair_end; ++p) {
        const uint64_t r = 2ull + db->lens[2 * p];
        total += r;
        longest = std::max(longest, r);
    }
    if (longest_rows) *longest_rows = longest;
    return total;
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

bool swg_plan_diag(size_t lq, uint64_t n_pairs, uint64_t pair_rows_total, uint64_t longest_rows, int n_cu,
                   long opt_cols, long opt_group, long opt_waves, SwgDiagPlan *out)
{
    const int nv = swg_num_diag_variants();
    const int groups[3] = {16, 32, 64};
    bool found = false;
    SwgDiagPlan best;
    memset(&best, 0, sizeof best);
    best.est_ms = 1e300;
    for (int v = 0; v < nv; ++v) {
        const SwgKernelInfo info = swg_diag_variant_info(v);
        if (opt_cols > 0 && info.K != (int)opt_cols) continue;
        for (int gi = 0; gi < 3; ++gi) {
            const int G = groups[gi];
            if (opt_group > 0 && G != (int)opt_group) continue;
            const size_t cols = (size_t)G * info.K;
            const size_t lds = cols * 64;
            if (lds > 160 * 1024) continue;
            const int npass = (int)((lq + cols - 1) / cols);
            const int NG = 64 / G;
            const double instr = 11.0 * info.K + (G == 32 ? 18.0 : 14.0); // per lane per step
            for (int wps = 1; wps <= 4; ++wps) {
                const int W = 4 * wps;
                if (W > info.max_waves) continue;
                if (opt_waves > 0 && W != (int)opt_waves) continue;
                // workgroups resident per CU: wave budget and LDS
                int per_cu = std::max(1, std::min<int>(info.max_waves / W, (int)((160 * 1024) / lds)));
                // several workgroups per CU raise the waves per SIMD
                const int eff_wps = std::min(4, wps * per_cu);
                const uint64_t hw_streams = (uint64_t)n_cu * per_cu * W * NG;
                const uint64_t spw = (uint64_t)W * NG; // lane groups of one workgroup
                uint64_t n_streams = std::max<uint64_t>(1, std::min<uint64_t>(hw_streams, n_pairs));
                n_streams = (n_streams + spw - 1) / spw * spw;
                const double cps = kCyclesPerInstr[eff_wps];
                // balanced share: rows per stream * steps; each wave-step serves NG streams
                const double rows_per_stream = (double)pair_rows_total / (double)n_streams;
                const double crit_rows = std::max<double>(rows_per_stream, (double)longest_rows) + G;
                const double cycles = crit_rows * npass * instr * cps * eff_wps;
                const double ms = cycles / 2.35e9 * 1e3;
                if (ms < best.est_ms) {
                    best.variant = v;
                    best.K = info.K;
                    best.G = G;
                    best.npass = npass;
                    best.W = W;
                    best.n_streams = (uint32_t)n_streams;
                    best.workgroups = (int)((n_streams + (uint64_t)W * NG - 1) / ((uint64_t)W * NG));
                    best.lds_bytes = lds;
                    best.est_ms = ms;
                    found = true;
                }
            }
        }
    }
    if (found) *out = best;
    return found;
}

void swg_build_diag_layout(const swg_db *db, uint64_t pair_begin, uint64_t pair_end, uint32_t n_streams,
                           uint32_t streams_per_wg, SwgDiagLayout *L)
{
    const size_t n_slots = (size_t)db->n_bins * SWG_BIN;
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
    L->tok.assign(L->total_blocks * 2, 0u);
    uint16_t *base = reinterpret_cast<uint16_t *>(L->tok.data()); // one 16-bit token per row
#pragma omp parallel for schedule(dynamic, 16)
    for (long long s = 0; s < (long long)n_streams; ++s) {
        uint16_t *t = base + L->stream_off[s] * 4;
        for (uint32_t i = L->stream_pair_off[s]; i < L->stream_pair_off[s + 1]; ++i) {
            const size_t p = L->stream_pairs[i];
            const uint32_t lx = db->lens[2 * p];
            const bool has_y = 2 * p + 1 < n_slots && db->order[2 * p + 1] != 0xFFFFFFFFu;
            const uint32_t ly = has_y ? db->lens[2 * p + 1] : 0;
            const uint8_t *cx = db->codes.data() + db->code_off[2 * p];
            const uint8_t *cy = has_y ? db->codes.data() + db->code_off[2 * p + 1] : nullptr;
            t[0] = 1; // reset rows: flag bit0, padding residue for both sequences
            t[1] = 1;
            for (uint32_t j = 0; j < lx; ++j)
                t[2 + j] = (uint16_t)(cx[j] | (j + 1 == lx ? 2u : 0u) | (j < ly ? (uint32_t)cy[j] << 8 : 0u));
            t += ((2ull + lx + 3) / 4) * 4; // rest of the last block stays padding
        }
    }
}
