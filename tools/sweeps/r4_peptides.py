"""Databases of very short sequences: 2 million peptides of 20-40 residues, queries of 128 and
30 columns -- a lane group needs a new pair every few token blocks, so the hand-out of pairs is what is timed.
usage: python tools/sweeps/r4_peptides.py [n_seqs] [opt=value ...]"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, swg_loader
swg = swg_loader.load(); orc = swg_loader.oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2000000
opts = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
sc = swg.load_scoring("BLOSUM62")
ctx = swg.Context(0)
flat, off = swg.synth_db(0xBEEF, n, median=29.0, sigma_ln=0.25, min_len=20, max_len=40)
lens = np.diff(off.astype(np.int64))
print("peptides:", n, "sequences, lengths", lens.min(), "..", lens.max(), "mean %.1f" % lens.mean(), flush=True)
ctx.set_scoring(sc, -2, -1)
for lq in (128, 30):
    q = swg.synth_query(0xBEEF + lq, lq)
    ctx.set_query(q)
    ctx.set_option("autotune", 0)
    for k, v in opts.items():
        ctx.set_option(k, int(v))
    db = swg.Database(flat, off).upload(ctx)
    got, hits, st = ctx.search(db, k=10)
    sample = np.linspace(0, n - 1, 4000).astype(np.int64)
    s_off = np.zeros(len(sample) + 1, dtype=np.uint64); s_off[1:] = np.cumsum(lens[sample])
    s_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in sample])
    ok = bool(np.array_equal(orc.score_db(q, s_flat, s_off, sc.table(), -2, -1), got[sample]))
    fills = []
    for _ in range(12):
        _, _, st = ctx.search(db, want_scores=False, k=10)
        fills.append(st["fill_ms"])
    f = float(np.median(fills[2:]))
    print("lq %d: fill %.3f ms  %.0f GCUPS  (K %d G %d W %d wgs %d form %d)  sample equals oracle: %s" % (
        lq, f, lq * float(off[-1]) / (f * 1e-3) / 1e9, st["cols_per_wave"], st["group_lanes"], st["waves"], st["workgroups"],
        st["cell_form"], ok), flush=True)
    assert ok
    db.close()
