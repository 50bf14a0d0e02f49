#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ (run in the container that has
/root/reference mounted; the outputs are committed, this script documents how).

Every expected score in the fixtures is produced by the REFERENCE's own
`alignment_fill_matrices` (src/alignment.c:47-187), compiled unmodified into
oracle/_ref/libswref.so by oracle/Makefile and driven 16 lanes at a time exactly
as its driver does (src/alignment_cmdline.c:429-509).  The int32 oracle's scores
are stored next to them; in the overflow fixture the two differ on purpose and
the reference's values are flagged invalid (it wraps in int16).

Fixture format: numpy .npz (arrays only, loadable with allow_pickle=False):
  sub[32][32] int8, gaps int32[2] (gap_open, gap_extend), query int8[lq],
  flat int8[...], offsets uint64[n+1]  -- database in reference batch order
  (n % 16 == 0, first of every 16 records the longest),
  lanes int32[n/16]                     -- real lanes per batch (<16: rest is '*' filler),
  ref16 int16[n]                        -- reference output per record,
  oracle32 int32[n]                     -- oracle/sw_oracle.c output per record,
  ref_valid uint8[1]                    -- 0 when the reference overflowed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import swg_loader  # noqa: E402

orc = swg_loader.oracle()
OUT = os.path.dirname(os.path.abspath(__file__))
AA20 = "ARNDCQEGHILKMFPSTWYV"
STAR = 31


def load_matrix(name):
    """Plain-Python reader of the NCBI-format files shipped in seq-align-gpu_amd/data."""
    sub = np.zeros((32, 32), dtype=np.int8)
    cols = None
    for line in open(os.path.join(ROOT, "seq-align-gpu_amd", "data", name + ".txt")):
        if line.startswith("#") or not line.strip():
            continue
        t = line.split()
        if cols is None:
            cols = t
            continue
        a = orc.letter_index(t[0])
        for c, v in zip(cols, t[1:]):
            sub[a, orc.letter_index(c)] = int(v)
    return sub


def idx(s):
    return np.array([orc.letter_index(c) for c in s], dtype=np.int8)


def rand_seq(rng, n, alphabet=AA20):
    return idx("".join(rng.choice(list(alphabet), size=n)))


def similar(rng, q, n, frac=0.33):
    """Random sequence of length n with runs copied from the query (real similarity)."""
    s = rand_seq(rng, n)
    pos = 0
    while pos < n:
        run = int(rng.integers(5, 40))
        if rng.random() < frac and len(q) > run:
            src = int(rng.integers(0, len(q) - run))
            m = min(run, n - pos)
            s[pos:pos + m] = q[src:src + m]
        pos += run
    return s


def build_case(rng, q, n_batches, lmin, lmax, lanes=None, seq_fn=None):
    seqs, lane_counts = [], []
    for b in range(n_batches):
        first = int(rng.integers(max(lmin, (lmin + lmax) // 2), lmax + 1))
        nl = 16 if lanes is None else lanes[b % len(lanes)]
        lens = [first] + [int(rng.integers(lmin, first + 1)) for _ in range(nl - 1)]
        for L in lens:
            seqs.append(seq_fn(rng, q, L) if seq_fn else similar(rng, q, L))
        for _ in range(16 - nl):  # filler lanes: all '*', as a harness must pass
            seqs.append(np.full(first, STAR, dtype=np.int8))
        lane_counts.append(nl)
    flat = np.concatenate(seqs)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return flat, offsets, np.array(lane_counts, dtype=np.int32)


def emit(name, sub, go, ge, q, flat, offsets, lanes, ref_valid=True):
    batches = orc.db_to_batches16(flat, offsets)
    ref = np.concatenate([orc.ref_batch16(q, b, sub, go, ge) for b in batches]).astype(np.int16)
    ora = orc.score_db(q, flat, offsets, sub, go, ge)
    if ref_valid:
        # filler lanes are '*' runs: the reference scores them too; compare everything
        bad = np.nonzero(ref.astype(np.int32) != ora)[0]
        assert bad.size == 0, (name, bad[:10], ref[bad[:10]], ora[bad[:10]])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), sub=sub, gaps=np.array([go, ge], dtype=np.int32),
                        query=q, flat=flat, offsets=offsets, lanes=lanes, ref16=ref, oracle32=ora,
                        ref_valid=np.array([1 if ref_valid else 0], dtype=np.uint8))
    print("%-28s n=%5d lq=%5d residues=%8d max=%6d ref==oracle:%s" % (
        name, len(offsets) - 1, len(q), len(flat), int(ora.max()), bool((ref == ora).all())))


def equal_length_batches(rng, q, lengths):
    """Batches whose 16 lanes all have the batch's length: nothing is padded, so the reference (which computes
    padded rows as real rows) and the per-pair oracle agree even when gap scores are positive and a
    gap through padding rows would keep gaining."""
    seqs = [similar(rng, q, L) for L in lengths for _ in range(16)]
    flat = np.concatenate(seqs)
    offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(s) for s in seqs])
    return flat, offsets, np.full(len(lengths), 16, dtype=np.int32)


def positive_gap_cases():
    """Round 2: gap scores whose INCREMENT is positive (gap_open + gap_extend > 0, or gap_extend > 0): the
    reference's CLI accepts them (src/alignment_cmdline.c:255-267), the packed int16 form cannot express them,
    so they pin the exact int32 kernel and the traceback."""
    rng = np.random.default_rng(20261004)
    b62 = load_matrix("BLOSUM62")
    q = rand_seq(rng, 150)
    emit("blosum62_gap_pos5_m1", b62, 5, -1, q, *equal_length_batches(rng, q, [120, 90, 64, 33, 7, 1]))
    emit("blosum62_gap_0_pos1", b62, 0, 1, q, *equal_length_batches(rng, q, [110, 80, 48, 16, 3]))


def mutated_prefix(rng, q, length, rate):
    """The query's first `length` residues with a fraction `rate` of them replaced at random."""
    s = q[:length].copy()
    hit = rng.random(length) < rate
    s[hit] = rand_seq(rng, int(hit.sum()))
    return s


def f16_boundary_cases():
    """Round 3: databases of close relatives of the query whose scores straddle 4096 -- the ceiling of the packed-f16
    cells (a score v is held as v - 2048, and f16 holds the integers of [-2048, 2048] exactly), from which on a
    sequence is flagged and re-scored in int32.  Batches of 16 records, the first the longest, as the reference's
    packer requires."""
    rng = np.random.default_rng(20261005)
    b62, pam = load_matrix("BLOSUM62"), load_matrix("PAM250")
    for name, sub, go, ge, lq in (("blosum62_f16_boundary", b62, -2, -1, 1150), ("blosum62_f16_boundary_gap11", b62, -11, -1, 1230),
                                  ("pam250_f16_boundary", pam, -2, -1, 1040)):
        q = rand_seq(rng, lq)
        seqs = []
        for b in range(6):
            first = lq - 10 * b
            lens = [first] + sorted((int(rng.integers(lq // 3, first + 1)) for _ in range(15)), reverse=True)
            for i, L in enumerate(lens):
                if b == 5 and i % 2:   # unrelated sequences between the relatives
                    seqs.append(rand_seq(rng, L))
                else:
                    seqs.append(mutated_prefix(rng, q, L, float(rng.choice([0.0, 0.03, 0.08, 0.15, 0.25, 0.4]))))
        flat = np.concatenate(seqs)
        offsets = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(x) for x in seqs])
        emit(name, sub, go, ge, q, flat, offsets, np.full(6, 16, dtype=np.int32))


def main():
    assert orc.have_ref(), "build oracle/_ref first (make -C oracle)"
    if "--positive-gaps" in sys.argv:          # only the fixtures added in round 2 (the others stay byte for byte)
        positive_gap_cases()
        return
    if "--f16-boundary" in sys.argv:           # only the fixtures added in round 3
        f16_boundary_cases()
        return
    rng = np.random.default_rng(20250523)
    pam, b62, b45 = load_matrix("PAM250"), load_matrix("BLOSUM62"), load_matrix("BLOSUM45")

    q = rand_seq(rng, 128)
    emit("pam250_lq128", pam, -2, -1, q, *build_case(rng, q, 40, 50, 450))
    q = rand_seq(rng, 367)
    emit("blosum62_lq367", b62, -2, -1, q, *build_case(rng, q, 40, 50, 450))
    q = rand_seq(rng, 200)
    emit("blosum45_lq200", b45, -2, -1, q, *build_case(rng, q, 24, 30, 400))
    # non-default gap scores, incl. gap_open 0 (linear gaps) and a steep open
    q = rand_seq(rng, 150)
    emit("blosum62_gap_10_1", b62, -10, -1, q, *build_case(rng, q, 16, 40, 300))
    emit("blosum62_gap_0_1", b62, 0, -1, q, *build_case(rng, q, 16, 40, 300))
    emit("pam250_gap_11_2", pam, -11, -2, q, *build_case(rng, q, 16, 40, 300))
    # gap scores outside the usual sign convention: exercised by the exact int32 form
    emit("blosum62_gap_pos1_m3", b62, 1, -3, q, *build_case(rng, q, 8, 30, 120))
    emit("blosum62_gap_m2_0", b62, -2, 0, q, *build_case(rng, q, 8, 30, 120))
    # query with ambiguity codes B/Z/X (lower case folds to the same indices)
    q = idx("mkvlaBZXxbzAGHWCYYNDEQBZXLLIVMFPSTWRK" * 3)
    emit("blosum62_query_bzx", b62, -2, -1, q,
         *build_case(rng, q, 16, 30, 200, seq_fn=lambda r, qq, L: rand_seq(r, L, AA20 + "BZX")))
    # fewer than 16 real lanes, replayed as 16 with '*' filler
    q = rand_seq(rng, 96)
    emit("pam250_partial_lanes", pam, -2, -1, q, *build_case(rng, q, 12, 20, 160, lanes=[1, 5, 15, 16, 9]))
    # long query
    q = rand_seq(rng, 3000)
    emit("blosum62_lq3000", b62, -2, -1, q, *build_case(rng, q, 2, 150, 400))
    # very short database sequences (1..8 residues) and a 1-residue query
    q = rand_seq(rng, 64)
    emit("blosum62_tiny_db", b62, -2, -1, q,
         *build_case(rng, q, 8, 1, 8, seq_fn=lambda r, qq, L: rand_seq(r, L)))
    q = rand_seq(rng, 1)
    emit("blosum62_lq1", b62, -2, -1, q, *build_case(rng, q, 4, 5, 60))

    # int16 overflow: PAM250 W:W = 17.  The reference wraps; truth is the int32 oracle.
    q = idx("W" * 3000)
    lens = [3000, 2000, 1928, 1927, 1000, 2500, 1950, 1900] + [1500] * 8
    seqs = [idx("W" * L) for L in lens]
    # first of the batch must be the longest (reference precondition)
    flat = np.concatenate(seqs)
    offsets = np.zeros(17, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    emit("pam250_overflow_w", pam, -2, -1, q, flat, offsets, np.array([16], dtype=np.int32), ref_valid=False)
    positive_gap_cases()
    f16_boundary_cases()


if __name__ == "__main__":
    main()
