import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, swg_loader
swg = swg_loader.load()
ctx = swg.Context(0)
sc = swg.load_scoring("PAM250")
q = swg.synth_query(0x5EED0002, 367)
ctx.set_scoring(sc, -2, -1); ctx.set_query(q); ctx.set_option("autotune", 0)
for maxlen in (5000, 3500, 2500, 1500):
    flat, off = swg.synth_db(0x5EED0002, 100000, max_len=maxlen)
    db = swg.Database(flat, off).upload(ctx)
    ctx.search(db, want_scores=False, k=100)
    fills = []
    for _ in range(12):
        _, _, st = ctx.search(db, want_scores=False, k=100)
        fills.append(st["fill_ms"])
    f = float(np.median(fills))
    print("max_len", maxlen, "residues", int(off[-1]), "fill %.3f ms" % f, "GCUPS(fill) %.0f" % (367 * int(off[-1]) / f / 1e6), "K", st["cols_per_wave"], "long", st["long_pairs"], st["long_cols_per_lane"])
    db.close()
