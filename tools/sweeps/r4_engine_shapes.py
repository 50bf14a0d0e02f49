"""Where the cost model's engine choice is right: uniform databases of short to medium sequences, both engines timed on
the same resident database, beside the model's pick (engine 0) and its two estimates."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, swg_loader
swg = swg_loader.load(); orc = swg_loader.oracle()
sc = swg.load_scoring("BLOSUM62")
ctx = swg.Context(0)
ctx.set_scoring(sc, -2, -1)
ctx.set_option("autotune", 0)
shapes = [("peptides 20-40", 1000000, dict(median=29.0, sigma_ln=0.25, min_len=20, max_len=40)),
          ("50-200, median 100", 500000, dict(median=100.0, sigma_ln=0.2, min_len=50, max_len=200)),
          ("100-370, median 250", 300000, dict(median=250.0, sigma_ln=0.3, min_len=100, max_len=370)),
          ("20-1000, median 150", 400000, dict(median=150.0, sigma_ln=0.6, min_len=20, max_len=1000))]
for name, n, kw in shapes:
    flat, off = swg.synth_db(0xABC, n, **kw)
    db = swg.Database(flat, off).upload(ctx)
    for lq in (64, 128, 367):
        q = swg.synth_query(0xABC + lq, lq)
        ctx.set_query(q)
        est = db.debug_engine(lq)
        out = {}
        base = None
        for e in (0, 1, 2):
            ctx.set_option("engine", e)
            got, _, st = ctx.search(db)
            if base is None: base = got
            assert np.array_equal(got, base)
            f = min(ctx.search(db, want_scores=False)[2]["fill_ms"] for _ in range(4))
            out[e] = (f, st["engine"], st["cell_form"], st["cols_per_wave"])
        pick = out[0][1]
        best = 1 if out[1][0] < out[2][0] else 2
        print("%-22s lq %3d: model picks engine %d (est diag %d us, systolic %d us); measured systolic %.3f ms (form %d K %d), lane groups %.3f ms (K %d) -> %s"
              % (name, lq, pick, est["diag_us"], est["systolic_us"], out[1][0], out[1][2], out[1][3], out[2][0], out[2][3],
                 "right" if pick == best else "WRONG by %.0f %%" % (100 * (out[pick][0] / out[best][0] - 1))), flush=True)
    ctx.set_option("engine", 0)
    db.close()
