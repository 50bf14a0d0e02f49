"""What would conflict-free profile reads be worth?  Config 3's database as it is, and with every residue replaced by the
same one ('A'): the lanes of an LDS access then read rows that differ only by their swizzle -- no two lanes of a 16-lane
group share a bank pair, lanes 16 apart read the same address (a broadcast) -- so the fill runs without bank conflicts,
on the same lengths, the same pairs and the same instruction stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import swg_loader
swg = swg_loader.load()
sc = swg.load_scoring("BLOSUM62")
q = swg.synth_query(0x5EED0003, 500)
flat, off = swg.synth_db(0x5EED0003, 570000)
ctx = swg.Context(0)
ctx.set_scoring(sc, -2, -1)
ctx.set_query(q)
ctx.set_option("autotune", 0)
for name, f in (("random residues", flat), ("one residue", np.full_like(flat, 1)), ("random residues", flat), ("one residue", np.full_like(flat, 1))):
    db = swg.Database(f, off).upload(ctx)
    for _ in range(6):
        ctx.search(db, k=10, want_scores=False)
    ms = []
    for _ in range(12):
        _, _, st = ctx.search(db, k=10, want_scores=False)
        ms.append(st["fill_ms"])
    print("%-16s fill %.3f ms (min %.3f)  %.0f GCUPS  K %d G %d form %d" % (name, np.mean(ms), np.min(ms), st["cells"] / np.mean(ms) / 1e6,
                                                                    st["cols_per_wave"], st["group_lanes"], st["cell_form"]), flush=True)
    db.close()
