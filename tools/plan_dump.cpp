// The planner's ranking for a synthetic database, on the host (no GPU): what the cost model of
// swg_diag_host.cpp makes of a query length, best candidates first.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tools/plan_dump.cpp \
//       -Lseq-align-gpu_amd -lswg -Wl,-rpath,$PWD/seq-align-gpu_amd -o /tmp/plan_dump
//   /tmp/plan_dump <lq> <n_seqs> [seed] [top]
#include "../seq-align-gpu_amd/csrc/swg_host_internal.h"
#include "../include/swg_host.h"
#include <cstdio>
#include <cstdlib>

int main(int argc, char **argv)
{
    const size_t lq = argc > 1 ? strtoul(argv[1], 0, 10) : 500, n = argc > 2 ? strtoul(argv[2], 0, 10) : 200000;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], 0, 0) : 0x5EED0003ull;
    const int top = argc > 4 ? atoi(argv[4]) : 8;
    int8_t *flat = nullptr;
    uint64_t *off = nullptr;
    if (swg_synth_db(seed, n, 290.0, 0.75, 20, 5000, &flat, &off) != 0) return 1;
    swg_db *db = nullptr;
    if (swg_db_pack(flat, off, n, 0, 1, &db) != 0) return 1;
    std::vector<SwgDiagWork> c;
    swg_plan_diag_candidates(db, lq, 256, 0, 0, 0, 0, true, true, &c);
    for (int i = 0; i < (int)c.size() && i < top; ++i) {
        const SwgDiagWork &w = c[i];
        printf("%2d est %.3f ms  K %d G %d W %d P %d wgs %d", i, w.plan[0].est_ms, w.plan[0].K, w.plan[0].G, w.plan[0].W,
               w.plan[0].npass, w.plan[0].workgroups);
        if (w.n_classes == 2)
            printf("  + long: %llu pairs K %d G %d wgs %d", (unsigned long long)w.pair_end[1], w.plan[1].K, w.plan[1].G,
                   w.plan[1].workgroups);
        printf("\n");
    }
    swg_db_free(db);
    return 0;
}
