"""CPU: the oracle (oracle/sw_oracle.c) against the committed golden vectors, and against the
reference build itself when oracle/_ref/libswref.so is present."""
import numpy as np
import pytest

from conftest import golden_names, load_golden


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_golden(orc, name):
    g = load_golden(name)
    go, ge = int(g["gaps"][0]), int(g["gaps"][1])
    got = orc.score_db(g["query"], g["flat"], g["offsets"], g["sub"], go, ge)
    assert np.array_equal(got, g["oracle32"])
    if g["ref_valid"][0]:
        # every expected value here came out of the reference's alignment_fill_matrices
        assert np.array_equal(got, g["ref16"].astype(np.int32))
    else:
        assert (got != g["ref16"].astype(np.int32)).any()


def test_wrap16_reproduces_reference_overflow(orc):
    """The reference's int16 lanes wrap; the wrap16 restatement reproduces
    what the reference returned on the overflow fixture, record by record."""
    g = load_golden("pam250_overflow_w")
    q, flat, off = g["query"], g["flat"], g["offsets"]
    # in the reference every lane walks max_len rows: shorter lanes see '*' padding
    max_len = int(off[1] - off[0])
    for i in range(len(off) - 1):
        d = np.full(max_len, 31, dtype=np.int8)
        s = flat[int(off[i]):int(off[i + 1])]
        d[:len(s)] = s
        assert orc.pair_wrap16(q, d, g["sub"], -2, -1) == int(g["ref16"][i])
    assert int(g["oracle32"][0]) == 51000 and int(g["ref16"][0]) == 32767


def test_letter_index_map(orc):
    for ch, want in (("A", 1), ("a", 1), ("Z", 26), ("z", 26), ("*", 31), ("X", 24), ("-", -1), ("1", -1)):
        assert orc.letter_index(ch) == want
    if orc.have_ref():
        for c in "ABCDEFGHIJKLMNOPQRSTUVWXYZabcxyz*":
            assert orc.rlib().swref_letters_to_index(ord(c)) == orc.letter_index(c)


def test_oracle_vs_reference_random(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref/libswref.so not built (reference not mounted)")
    rng = np.random.default_rng(7)
    g = load_golden("blosum62_lq367")
    sub = g["sub"]
    for trial in range(6):
        lq = int(rng.integers(1, 300))
        q = rng.integers(1, 26, size=lq).astype(np.int8)
        q[q == 10] = 1  # J/O/U are undefined in the matrices
        q[q == 15] = 1
        q[q == 21] = 1
        first = int(rng.integers(20, 200))
        seqs = []
        for l in range(16):
            L = first if l == 0 else int(rng.integers(1, first + 1))
            s = rng.integers(1, 26, size=L).astype(np.int8)
            s[(s == 10) | (s == 15) | (s == 21)] = 3
            seqs.append(s)
        go, ge = [(-2, -1), (-5, -2), (0, -1), (-3, 0), (2, -4), (-1, 1)][trial]
        ref = orc.ref_batch16(q, orc.make_batch16(seqs), sub, go, ge)
        for l in range(16):
            # what the reference computes: every lane walks `first` rows, '*'-padded
            # (src/alignment_cmdline.c:448-450) ...
            padded = np.full(first, 31, dtype=np.int8)
            padded[:len(seqs[l])] = seqs[l]
            assert orc.pair(q, padded, sub, go, ge) == int(ref[l]), (trial, l)
            # ... which equals the per-pair score whenever padding cannot score
            # (non-positive gap scores, S[q]['*'] <= 0: SURVEY A.3)
            if go <= 0 and ge <= 0:
                assert orc.pair(q, seqs[l], sub, go, ge) == int(ref[l]), (trial, l)


def test_oracle_edge_cases(orc):
    g = load_golden("blosum62_lq1")
    sub = g["sub"]
    a = np.array([1], dtype=np.int8)
    assert orc.pair(a, a, sub, -2, -1) == 4          # A:A in BLOSUM62
    assert orc.pair(a, np.array([18], dtype=np.int8), sub, -2, -1) == 0  # A:R = -1 -> floor 0
    assert orc.score_db(a, np.zeros(0, np.int8), np.zeros(1, np.uint64), sub, -2, -1).size == 0
    assert orc.topk(np.array([5, 9, 9, 1], dtype=np.int32), 3) == [(9, 1), (9, 2), (5, 0)]


def test_oracle_alignment_paths_score_what_the_reference_scored(orc):
    """sw_oracle_pair_trace: the path's substitution and gap scores add up to the score the
    reference's own fill produced for the pair (the path itself has no reference output: the fork
    removed the traceback), its coordinates bound the path, and it ends on a residue pair."""
    for name in ("pam250_lq128", "blosum62_gap_10_1", "blosum62_gap_0_1", "blosum62_gap_pos1_m3", "blosum62_gap_pos5_m1", "blosum62_gap_0_pos1", "blosum62_query_bzx",
                 "blosum62_tiny_db", "blosum62_lq1"):
        g = load_golden(name)
        go, ge = int(g["gaps"][0]), int(g["gaps"][1])
        off = g["offsets"]
        for i in range(0, len(off) - 1, 7):
            d = g["flat"][int(off[i]):int(off[i + 1])]
            sc, co, ops = orc.pair_trace(g["query"], d, g["sub"], go, ge)
            assert sc == int(g["oracle32"][i])
            if g["ref_valid"][0]:
                assert sc == int(g["ref16"][i])
            assert orc.path_score(g["query"], d, g["sub"], go, ge, co, ops) == sc
            assert (sc > 0) == (len(ops) > 0) and (not ops or ops[-1] == "M")
            assert co[1] - co[0] == ops.count("M") + ops.count("D") and co[3] - co[2] == ops.count("M") + ops.count("I")
