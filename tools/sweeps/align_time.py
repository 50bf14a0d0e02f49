"""Time swg_align_hits on the top-100 of config 2 and of a long query (GPU box)."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
swg = importlib.import_module("seq-align-gpu_amd")
for lq, n, mat in ((367, 100000, "PAM250"), (3000, 20000, "BLOSUM62")):
    sc = swg.load_scoring(mat)
    q = swg.synth_query(0x5EED0002, lq)
    flat, off, _ = swg.synth_db(0x5EED0002, n, query=q, fraction=0.001, subst=0.05)
    ctx = swg.Context(0)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    db = swg.Database(flat, off).upload(ctx)
    _, hits, st = ctx.search(db, want_scores=False, k=100)
    for want_ops in (True, False):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            als = ctx.align_hits(db, hits, want_ops=want_ops)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("lq %d: %d hits, ops=%s: %s ms; fill %.2f ms; longest path %d, cells %.3g" %
              (lq, len(hits), want_ops, ["%.2f" % t for t in ts], st["fill_ms"], max(a["n_ops"] for a in als),
               sum(lq * (int(off[a["index"] + 1]) - int(off[a["index"]])) for a in als)), flush=True)
    assert [a["score"] for a in als] == [s for s, _ in hits]
    db.close(); ctx.close()
