#!/usr/bin/env python3
"""Randomised parity soak of swg_search_multi (batches of queries against one resident database): random batch
sizes and query lengths (equal, mixed, some beyond one pass, some high-scoring relatives of database sequences),
tables, gap scores, cell forms (option f16 0 / 1 / 2), two queries per lane on or off (option qq), with and without
the score array (hits only: the batch's top-K selected on the device).  Every score against the int32 oracle, every
hit list against the oracle's order.
usage: python tests/fuzz_multi_gpu.py [seconds] [seed]"""
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
    forms, sizes = {}, {}
    while time.time() < t_end:
        shape = rng.integers(0, 4)
        n = int(rng.integers(1, 2500))
        if shape == 0:   lens = rng.integers(1, 60, size=n)
        elif shape == 1: lens = rng.integers(1, 500, size=n)
        elif shape == 2: lens = np.clip(rng.lognormal(5.0, 0.8, size=n), 1, 3000).astype(np.int64)
        else:            lens = np.concatenate([rng.integers(800, 2500, size=min(n, 2)), rng.integers(1, 150, size=max(0, n - 2))])
        lens = [int(v) for v in lens]
        # odd batches either side of the launch's chunk of 256 queries too: with two queries per lane the last pair of
        # an odd batch (or of an odd last chunk: 257 = 256 + 1, 513 = 2 * 256 + 1) holds one query twice
        nq = int(rng.choice([1, 2, 3, 5, 8, 13, 32, 40, 255, 257, 513], p=[.12, .1, .14, .1, .1, .1, .1, .09, .05, .05, .05]))
        base = int(rng.choice([1, 7, 33, 128, 200, 367, 500, 900]))
        if nq > 200:     # (keeps the oracle's share of a case to seconds)
            base = int(rng.choice([1, 7, 33, 64]))
            n = min(n, 600)
            lens = lens[:n]
        mode = rng.integers(0, 3)
        if mode == 0:   qlens = [base] * nq
        elif mode == 1: qlens = [max(1, int(base * rng.uniform(0.3, 1.2))) for _ in range(nq)]
        elif nq > 200:  qlens = [int(rng.choice([1, 5, 33, 64, 90])) for _ in range(nq)]
        else:           qlens = [int(rng.choice([5, 64, 300, 1100, 2300])) for _ in range(nq)]   # some need several passes
        if sum(lens) * sum(qlens) > 4e9:
            continue
        sc = swg.load_scoring(str(rng.choice(mats)))
        go, ge = [(-2, -1), (-10, -1), (0, -1), (-3, 0), (-11, -2), (1, -3)][int(rng.integers(0, 6))]
        seqs = [swg.synth_query(int(rng.integers(1, 1 << 30)), L) for L in lens]
        queries = [swg.synth_query(int(rng.integers(1, 1 << 30)), L) for L in qlens]
        if rng.random() < 0.4:     # relatives: a query copied into a few sequences (scores far above the rest, some beyond 4096)
            for _ in range(int(rng.integers(1, 6))):
                qi, si = int(rng.integers(0, nq)), int(rng.integers(0, len(seqs)))
                m = min(len(queries[qi]), len(seqs[si]))
                seqs[si][:m] = queries[qi][:m]
        flat = np.concatenate(seqs); off = np.zeros(len(lens) + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
        ctx.set_scoring(sc, go, ge)
        for k_ in ("force_bits", "engine", "cols_per_wave", "max_waves", "group_lanes", "long_split", "workgroups", "segment_blocks"):
            ctx.set_option(k_, 0)
        for k_ in ("work_queue", "wide16", "autotune", "side_readout", "f16", "qq", "last_pass"):
            ctx.set_option(k_, 1)
        ctx.set_option("batch", 8); ctx.set_option("batch_blocks", 0)
        opts = {}
        if rng.random() < 0.25: opts["batch"] = int(rng.choice([0, 3, 8]))
        if rng.random() < 0.25: opts["batch_blocks"] = int(rng.choice([1, 64, 100000]))
        if rng.random() < 0.3: opts["qq"] = 0
        if rng.random() < 0.3: opts["f16"] = int(rng.choice([0, 2]))
        if rng.random() < 0.2: opts["autotune"] = 0
        if rng.random() < 0.15: opts["long_split"] = int(rng.choice([-1, 100, 500]))
        for k_, v in opts.items(): ctx.set_option(k_, v)
        ctx.set_query(queries[0])
        db = swg.Database(flat, off).upload(ctx)
        k = int(rng.choice([0, 1, 4, 30, 300]))
        want_scores = bool(rng.random() < 0.6) or k == 0
        got, hits, st = ctx.search_multi(db, queries, k=k, want_scores=want_scores)
        for i, q in enumerate(queries):
            want = orc.score_db(q, flat, off, sc.table(), go, ge)
            if want_scores and not np.array_equal(got[i], want):
                bad = np.nonzero(got[i] != want)[0]
                print("MISMATCH case", cases, "query", i, "of", nq, "lq", len(q), "n", n, "gaps", go, ge, "opts", opts, "stats", st)
                print("  first bad:", bad[:10], got[i][bad[:10]], want[bad[:10]])
                return 1
            if k and hits[i] != orc.topk(want, k):
                print("HITS DIFFER case", cases, "query", i, "of", nq, "lq", len(q), "n", n, "k", k, "scores asked", want_scores, "opts", opts, "stats", st)
                print("  got ", hits[i][:6], "\n  want", orc.topk(want, k)[:6])
                return 1
        db.close()
        cases += 1
        forms[int(st["cell_form"])] = forms.get(int(st["cell_form"]), 0) + 1
        sizes[nq] = sizes.get(nq, 0) + 1
        if cases % 20 == 0:
            print("cases", cases, "last: nq", nq, "qlens", qlens[:4], "n", n, "k", k, "scores", want_scores, "opts", opts, "form", st["cell_form"],
                  "K", st["cols_per_wave"], "G", st["group_lanes"], flush=True)
    print("OK", cases, "cases; by cell form", dict(sorted(forms.items())), "by batch size", dict(sorted(sizes.items())))
    return 0


if __name__ == "__main__":
    sys.exit(main(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
