#!/usr/bin/env python3
"""Randomised parity soak: random databases (tiny to long-tailed), query lengths, scoring tables,
gap scores and engine options through the C ABI, every score against the int32 oracle.
usage: python tests/fuzz_gpu.py [seconds] [seed]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import swg_loader

def main(budget=300.0, seed=1):
    swg = swg_loader.load(); orc = swg_loader.oracle()
    rng = np.random.default_rng(seed)
    ctx = swg.Context(0)
    mats = ["BLOSUM62", "PAM250", "BLOSUM45"]
    t_end = time.time() + budget
    cases = 0
    forms = {}   # searches by swg_stats.cell_form
    ks = [2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 16, 17, 20, 23, 24, 25, 28, 31, 32]
    while time.time() < t_end:
        shape = rng.integers(0, 5)
        n = int(rng.integers(1, 3000))
        if shape == 0:   lens = rng.integers(1, 8, size=n)
        elif shape == 1: lens = rng.integers(1, 400, size=n)
        elif shape == 2: lens = np.clip(rng.lognormal(5.0, 0.9, size=n), 1, 4000).astype(np.int64)
        elif shape == 3: lens = np.concatenate([rng.integers(1500, 6000, size=min(n, 3)), rng.integers(1, 120, size=max(0, n - 3))])
        else:            lens = np.full(n, int(rng.integers(1, 300)))
        lens = [int(v) for v in lens]
        lq = int(rng.choice([1, 2, 17, 64, 128, 200, 367, 368, 369, 500, 777, 1024, 1500, 2100, 3000]))
        if sum(lens) * lq > 6e9:
            continue
        sc = swg.load_scoring(str(rng.choice(mats)))
        # (the last three: positive gap INCREMENTS -- gap_open + gap_extend > 0 or gap_extend > 0 -- which only the
        # exact int32 form can express; the reference's CLI accepts them)
        go, ge = [(-2, -1), (-10, -1), (0, -1), (-3, 0), (-11, -2), (5, -1), (0, 1), (1, -3)][int(rng.integers(0, 8))]
        q = swg.synth_query(int(rng.integers(1, 1 << 30)), lq)
        seqs = [swg.synth_query(int(rng.integers(1, 1 << 30)), L) for L in lens]
        if rng.random() < 0.12:   # scores beyond int16 and beyond 65535: tryptophan-rich query, copies of its prefixes
            lq = int(rng.choice([2100, 3000, 4500]))
            q = swg.synth_query(int(rng.integers(1, 1 << 30)), lq)
            q[rng.random(lq) < 0.9] = 23
            sc = swg.load_scoring("PAM250")
            lens = [int(v) for v in rng.integers(1500, lq + 1, size=4)] + [int(v) for v in rng.integers(1, 200, size=int(rng.integers(1, 300)))]
            # (both 16-bit forms in one search: decoys either side of its length cut, and tryptophan runs that beat the
            # f16 ceiling from below the cut)
            lens += [int(v) for v in rng.integers(200, 700, size=int(rng.integers(0, 40)))]
            n_runs = int(rng.integers(0, 6))
            lens += [int(v) for v in rng.integers(200, 420, size=n_runs)]
            seqs = [q[:L].copy() if i < 4 else swg.synth_query(int(rng.integers(1, 1 << 30)), L) for i, L in enumerate(lens)]
            for s_ in seqs[len(seqs) - n_runs:]:
                s_[:] = 23
        if rng.random() < 0.3 and lq > 50:   # plant similar sequences
            for i in rng.integers(0, len(seqs), size=min(5, len(seqs))):
                L = len(seqs[i]); m = min(L, lq); seqs[i][:m] = q[:m]
        flat = np.concatenate(seqs); off = np.zeros(len(lens) + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
        want = orc.score_db(q, flat, off, sc.table(), go, ge)
        ctx.set_scoring(sc, go, ge); ctx.set_query(q)
        for k in ("force_bits", "engine", "cols_per_wave", "max_waves", "group_lanes", "long_split", "workgroups", "segment_blocks"):
            ctx.set_option(k, 0)
        for k in ("work_queue", "wide16", "autotune", "side_readout", "f16", "last_pass"):
            ctx.set_option(k, 1)
        ctx.set_option("long_helps", 0)
        ctx.set_option("batch", 8); ctx.set_option("batch_blocks", 0)
        opts = {}
        r = rng.random()
        if r < 0.25:
            opts = {"engine": 2, "cols_per_wave": int(rng.choice(ks)), "group_lanes": int(rng.choice([16, 32, 64])), "max_waves": 4}
        elif r < 0.35:
            opts = {"engine": 1}
        elif r < 0.45:
            opts = {"work_queue": 0}
        elif r < 0.55:
            opts = {"long_split": int(rng.choice([-1, 100, 500, 2000]))}
        elif r < 0.6:
            opts = {"force_bits": 32}
        elif r < 0.7:   # the int32 work-queue kernel at a forced geometry (round 3: these two families used to be exclusive)
            opts = {"force_bits": 32, "cols_per_wave": int(rng.choice(ks)), "group_lanes": int(rng.choice([16, 32, 64]))}
            if rng.random() < 0.5: opts["max_waves"] = 4
        if rng.random() < 0.2: opts["long_helps"] = 1
        if rng.random() < 0.3: opts["f16"] = int(rng.choice([0, 2]))   # int16 cells only / f16 cells whatever the score bound
        if rng.random() < 0.2: opts["autotune"] = 0
        if rng.random() < 0.2: opts["last_pass"] = 0   # every pass of a long query with the same columns per lane
        if rng.random() < 0.15: opts["wide16"] = 0
        # the work queue's batches: off, small, and thresholds from "no pair is short" to "every pair is" (a claim of
        # eight LONG pairs is legal, only slow)
        if rng.random() < 0.3: opts["batch"] = int(rng.choice([0, 2, 5, 8]))
        if rng.random() < 0.3: opts["batch_blocks"] = int(rng.choice([1, 4, 64, 100000]))
        if rng.random() < 0.25:   # multi-pass launches cut into segments of consecutive pairs (never shorter than a pair)
            opts["segment_blocks"] = int((max(lens) + 5) // 4 * rng.choice([1, 2, 7]) + rng.integers(0, 3))
        if "cols_per_wave" in opts and opts["cols_per_wave"] * opts["group_lanes"] * 64 > 150 * 1024:
            opts = {}
        for k, v in opts.items(): ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        try:
            got, hits, st = ctx.search(db, k=int(rng.integers(0, 20)))
        except swg.SwgError as e:
            if "no diagonal-engine geometry" in str(e):   # forced geometry that cannot run this query
                db.close(); continue
            raise
        if not np.array_equal(got, want):
            bad = np.nonzero(got != want)[0]
            print("MISMATCH case", cases, "n", n, "lq", lq, "shape", shape, "gaps", go, ge, "opts", opts, "stats", st)
            print("  first bad:", bad[:10], got[bad[:10]], want[bad[:10]])
            return 1
        db.close()
        cases += 1
        forms[int(st["cell_form"])] = forms.get(int(st["cell_form"]), 0) + 1
        if cases % 25 == 0:
            print("cases", cases, "last: n", n, "lq", lq, "opts", opts, "engine", st["engine"], "K", st["cols_per_wave"], "G", st["group_lanes"], "P", st["passes"], "wq", st["work_queue"], flush=True)
    print("OK", cases, "cases; by cell form", dict(sorted(forms.items())))
    return 0

if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
