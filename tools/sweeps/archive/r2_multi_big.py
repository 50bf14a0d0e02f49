"""swg_search_multi on a large database: 8 / 32 queries of config 2's shape in one pass against 100k sequences."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
import numpy as np
import swg_loader
swg = swg_loader.load()
sc = swg.load_scoring("PAM250")
flat, off = swg.synth_db(0x5EED0002, 100000)
ctx = swg.Context(0)
ctx.set_scoring(sc, -2, -1)
ctx.set_option("autotune", 0)
db = swg.Database(flat, off).upload(ctx)
q0 = swg.synth_query(0x5EED0002, 367)
ctx.set_query(q0)
ctx.search(db, want_scores=False)
_, _, one = ctx.search(db, want_scores=False)
print("one query: fill %.3f ms, %.0f GCUPS" % (one["fill_ms"], one["cells"] / one["fill_ms"] / 1e6))
for nq in (2, 8, 32):
    qs = [swg.synth_query(1000 + i, 367) for i in range(nq)]
    ctx.search_multi(db, qs, want_scores=False)
    t0 = time.perf_counter()
    got, _, st = ctx.search_multi(db, qs, want_scores=True)
    wall = time.perf_counter() - t0
    print("%d queries in one pass: fill %.3f ms, %.0f GCUPS (wall %.1f ms incl. %d MB of scores), K %d G %d long %d"
          % (nq, st["fill_ms"], st["cells"] / st["fill_ms"] / 1e6, wall * 1e3, got.nbytes >> 20, st["cols_per_wave"], st["group_lanes"], st["long_pairs"]))
    ctx.set_query(qs[-1])
    ref, _, _ = ctx.search(db)
    assert np.array_equal(got[-1], ref)
