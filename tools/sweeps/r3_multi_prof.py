"""swg_search_multi, 8 queries of config 2's shape in one pass: two queries per lane (qq=1) against two sequences per lane
(qq=0), for a counter run:  rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- python3 tools/sweeps/r3_multi_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import swg_loader
swg = swg_loader.load()
sc = swg.load_scoring("PAM250")
flat, off = swg.synth_db(0x5EED0002, 100000)
ctx = swg.Context(0)
ctx.set_scoring(sc, -2, -1)
ctx.set_option("autotune", 0)
db = swg.Database(flat, off).upload(ctx)
qs = [swg.synth_query(1000 + i, 367) for i in range(8)]
for qq in (1, 0):
    ctx.set_option("qq", qq)
    for _ in range(3):
        _, _, st = ctx.search_multi(db, qs, want_scores=False)
    print("qq", qq, "fill %.3f ms %.0f GCUPS form %d K %d G %d W %d wgs %d" % (st["fill_ms"], st["cells"] / st["fill_ms"] / 1e6, st["cell_form"], st["cols_per_wave"], st["group_lanes"], st["waves"], st["workgroups"]))
