#!/usr/bin/env python3
"""Per-SIMD view of an SWG_TRACE file (last fill): which wavefronts shared a SIMD and how fast each ran."""
import sys, collections
import numpy as np
rows = []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        rows = []
        continue
    rows.append([int(x) for x in line.split()])
a = np.array(rows, dtype=np.uint64)
a = a[a[:, 4] > 0]
cls = a[:, 0].astype(int); wg = a[:, 1].astype(int); wv = a[:, 2].astype(int)
blocks = (a[:, 5] & np.uint64(0xFFFFFFFF)).astype(np.int64)
dur = (a[:, 4] - a[:, 3]).astype(np.int64) / 100.0
rate = dur / np.maximum(blocks, 1)
hw = a[:, 6]
hwid = (hw & np.uint64(0xFFFF)).astype(int); rank = ((hw >> np.uint64(32)) & np.uint64(0xFF)).astype(int); xcc = ((hw >> np.uint64(40)) & np.uint64(0xF)).astype(int)
slot = hwid & 15; simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = list(zip(xcc, se, sh, cu, simd))
print("distinct xcc", sorted(set(xcc)), "se", sorted(set(se)), "sh", sorted(set(sh)), "cu", sorted(set(cu)), "simd", sorted(set(simd)))
per = collections.defaultdict(list)
for i, k in enumerate(key):
    per[k].append((cls[i], rank[i], slot[i], wg[i], wv[i], round(rate[i], 1), blocks[i]))
print("SIMDs seen:", len(per), "waves per SIMD histogram:", np.bincount([len(v) for v in per.values()]))
for k in list(per)[:6]:
    print(k, sorted(per[k]))
for c in sorted(set(cls)):
    m = cls == c
    print("class", c, "rank histogram", np.bincount(rank[m]), "rate by rank", [round(float(np.median(rate[m & (rank == r)])), 2) for r in sorted(set(rank[m]))])
    print("   wave-in-WG -> simd:", [np.bincount(simd[m & (wv == v)], minlength=4).tolist() for v in sorted(set(wv[m]))])
