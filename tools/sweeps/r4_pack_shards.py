"""swg_group_load's packing: all 8 shards of the 10M-sequence config-4 database from ONE global
sort (swg_db_pack_shards) against round 3's way -- swg_db_pack(whole, r, 8) for every r, i.e. eight global sorts one
after another.  Host only; run on the GPU box for its 16 CPUs."""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, swg_loader
swg = swg_loader.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
t0 = time.time(); flat, off = swg.synth_db(0x5EED0004, n); t_gen = time.time() - t0
print("database: %d sequences, %.2f G residues, generated in %.1f s; host threads %d" % (n, float(off[-1]) / 1e9, t_gen, swg.lib.swg_host_threads()), flush=True)
s0 = swg.lib.swg_debug_sort_count()
t0 = time.time(); shards = swg.Database.pack_shards(flat, off, 8); t_new = time.time() - t0
print("swg_db_pack_shards(8): %.2f s, %d global sort(s)" % (t_new, swg.lib.swg_debug_sort_count() - s0), flush=True)
res = [(d.count, d.residues) for d in shards]
for d in shards: d.close()
s0 = swg.lib.swg_debug_sort_count()
t0 = time.time()
old = []
for r in range(8):
    d = swg.Database(flat, off, r, 8); old.append((d.count, d.residues)); d.close()
t_old = time.time() - t0
print("8 x swg_db_pack(whole, r, 8): %.2f s, %d global sorts" % (t_old, swg.lib.swg_debug_sort_count() - s0), flush=True)
assert old == res
print("same shards; speed-up %.1fx" % (t_old / t_new))
