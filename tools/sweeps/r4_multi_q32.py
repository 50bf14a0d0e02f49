"""Round 4, after the score hop reached swg_diag_qq_kernel and swg_diag32q_kernel: batches of queries on small and
large databases (64 x config 1's shape, 8 x config 2's), on a peptide database, and the int32 fills."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np
import swg_loader
swg = swg_loader.load(); orc = swg_loader.oracle()
ctx = swg.Context(0)
ctx.set_option("autotune", 0)

def multi(name, sc, flat, off, qs, check=2):
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(qs[0])
    db = swg.Database(flat, off).upload(ctx)
    ctx.search_multi(db, qs, want_scores=False)
    fills = []
    for _ in range(5):
        got, _, st = ctx.search_multi(db, qs, want_scores=True)
        fills.append(st["fill_ms"])
    f = float(np.median(fills))
    ok = all(np.array_equal(got[i], orc.score_db(qs[i], flat, off, sc.table(), -2, -1)) for i in range(min(check, len(qs))))
    print("%s: %d queries in one pass: fill %.3f ms, %.0f GCUPS, form %d K %d G %d W %d; first %d queries equal the oracle: %s"
          % (name, len(qs), f, st["cells"] / f / 1e6, st["cell_form"], st["cols_per_wave"], st["group_lanes"], st["waves"], check, ok), flush=True)
    assert ok
    db.close()

b62, pam = swg.load_scoring("BLOSUM62"), swg.load_scoring("PAM250")
flat, off = swg.synth_db(0x5EED0001, 1024)
multi("config 1's shape", b62, flat, off, [swg.synth_query(100 + i, 128) for i in range(64)])
flat, off = swg.synth_db(0x5EED0002, 100000)
multi("config 2's shape", pam, flat, off, [swg.synth_query(1000 + i, 367) for i in range(8)], check=1)
flat, off = swg.synth_db(0xBEEF, 500000, median=29.0, sigma_ln=0.25, min_len=20, max_len=40)
multi("500k peptides", b62, flat, off, [swg.synth_query(2000 + i, 40 + 3 * i) for i in range(16)])
