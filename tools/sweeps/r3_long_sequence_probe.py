import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, swg_loader
swg = swg_loader.load(); orc = swg_loader.oracle()
sc = swg.load_scoring("BLOSUM62")
ctx = swg.Context(0)
for lq in (400, 2500):
    q = swg.synth_query(77, lq)
    lens = [300000, 120000, 7] + [int(v) for v in np.random.default_rng(3).integers(1, 600, size=400)]
    seqs = [swg.synth_query(1000 + i, L) for i, L in enumerate(lens)]
    seqs[0][150000:150000 + lq] = q           # the query inside the giant
    flat = np.concatenate(seqs); off = np.zeros(len(lens) + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    ctx.set_scoring(sc, -2, -1); ctx.set_query(q)
    for opts in ({}, {"f16": 0}, {"force_bits": 32}, {"autotune": 0, "long_split": -1}):
        for k in ("force_bits", "long_split"): ctx.set_option(k, 0)
        for k in ("f16", "autotune"): ctx.set_option(k, 1)
        for k, v in opts.items(): ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        got, hits, st = ctx.search(db, k=5)
        print("lq", lq, opts, "equal", bool(np.array_equal(got, want)), "best", hits[0], "form", st["cell_form"], "K", st["cols_per_wave"], "G", st["group_lanes"], "passes", st["passes"],
              "long_pairs", st["long_pairs"], "fill %.2f ms" % st["fill_ms"], flush=True)
        assert np.array_equal(got, want)
        db.close()
print("ok")
