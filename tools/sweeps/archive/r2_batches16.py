"""Path A of INTEGRATION.md end to end: config 2's database handed over as the reference's own 16-lane
transposed batches (sorted by length, padded to each batch's longest: src/alignment_cmdline.c:429-452),
one swg_fill_batches16 call per macro-batch of 8192 batches like the reference's threads x 512; wall clock
includes re-packing, the upload over PCIe, the device-side layout build and the copy back of the scores."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import swg_loader
swg = swg_loader.load()
sc = swg.load_scoring("PAM250")
lq, n = 367, 100000
q = swg.synth_query(0x5EED0002, lq)
flat, off = swg.synth_db(0x5EED0002, n)
lens = np.diff(off).astype(np.int64)
order = np.argsort(-lens, kind="stable")
batches = []
t0 = time.perf_counter()
for b in range(0, n, 16):
    idx = order[b:b + 16]
    ml = int(lens[idx[0]])
    d = np.full((ml, 16), 31, dtype=np.int8)    # padded rows hold a real residue index and are computed as real rows, like the reference's
    for l, i in enumerate(idx):
        d[:lens[i], l] = flat[int(off[i]):int(off[i + 1])]
    batches.append((d, len(idx)))
print("built %d reference-shaped batches in %.2f s (python)" % (len(batches), time.perf_counter() - t0), flush=True)
ctx = swg.Context(0)
ctx.set_scoring(sc, -2, -1); ctx.set_query(q)
cells_real = int(lens.sum()) * lq
cells_padded = sum(d.shape[0] * 16 for d, _ in batches) * lq
import ctypes as C
arrs = []
for m in range(0, len(batches), 8192):           # the argument arrays, outside the timed region
    chunk = batches[m:m + 8192]
    arr = (swg.Batch16 * len(chunk))()
    outs = np.zeros((len(chunk), 16), dtype=np.int16)
    for i, (d, vs) in enumerate(chunk):
        arr[i].db_idx_t = d.ctypes.data; arr[i].max_len = d.shape[0]; arr[i].vector_size = vs
        arr[i].max_scores = outs[i].ctypes.data
    arrs.append((arr, len(chunk), outs))
for rep in range(4):
    t0 = time.perf_counter(); dev = 0.0
    for arr, nb, outs in arrs:
        secs = C.c_double(0)
        rc = swg.lib.swg_fill_batches16(ctx.handle, arr, nb, C.byref(secs)); assert rc == 0, rc
        dev += secs.value
    wall = time.perf_counter() - t0
    print("rep %d: wall %.1f ms (device %.1f ms) -> %.0f GCUPS on the real cells, %.0f on the cells the reference would compute (padded rows)"
          % (rep, wall * 1e3, dev * 1e3, cells_real / wall / 1e9, cells_padded / wall / 1e9), flush=True)
