"""CPU: the arithmetic of the packed-f16 cells (seq-align-gpu_amd/csrc/swg_kernels.hip, CellsDiag FORM 2),
re-done in numpy float16 -- the same operations in the same order, one rounding each, as v_pk_add_f16 /
v_pk_maximum3_f16 perform them -- against the int32 oracle.  What the kernel's correctness rests on, checked without a
GPU: a pair whose computed best stays below +2048.0 (score 4096) has the oracle's score exactly, and a pair is flagged
(computed best >= 2048.0) exactly when its true score is 4096 or more.  Scores are pushed through the whole range, far
beyond the ceiling too (the values then run through the inexact part of f16 and must still come out flagged), with gap
magnitudes from 0 to 2048 and table entries over the whole int8 range."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401  (path set-up)
import swg_loader

F = np.float16


def f16_cells_best(q, d, sub, g, e):
    """Score of one pair as the f16 cells compute it: values held as v - 2048, floor F = -2048, G unfloored.
    Returns the computed best as a float16 (biased)."""
    lq = len(q)
    FLOOR = F(-2048.0)
    gg, ee = F(g), F(e)
    M = np.full(lq + 1, FLOOR, dtype=F)   # M[i] of the previous row; M[0] = column 0 = score 0
    A = np.full(lq + 1, FLOOR, dtype=F)
    G = (M - gg).astype(F)                # M - g of the previous row, unfloored
    best = FLOOR
    for r in range(len(d)):
        s = sub[q, d[r]].astype(F)        # profile row of this residue
        Mn = np.full(lq + 1, FLOOR, dtype=F)
        An = np.full(lq + 1, FLOOR, dtype=F)
        Gn = np.empty(lq + 1, dtype=F)
        Gn[0] = F(FLOOR - gg)
        gl, bl = Gn[0], FLOOR             # column 0: M = B = score 0
        for i in range(1, lq + 1):
            t = F(M[i - 1] + s[i - 1])
            a = max(G[i], F(A[i] - ee), FLOOR)
            b = max(gl, F(bl - ee), FLOOR)
            m = max(t, a, b)
            Mn[i], An[i] = m, a
            gl = Gn[i] = F(m - gg)
            bl = b
            if m > best:
                best = m
        M, A, G = Mn, An, Gn
    return best


@pytest.mark.parametrize("seed,gaps", [(1, (-2, -1)), (2, (-11, -1)), (3, (0, -1)), (4, (-3, 0)), (5, (-2048, 0)), (6, (-1000, -1048))])
def test_f16_cells_are_exact_below_their_ceiling_and_flag_everything_else(seed, gaps):
    orc = swg_loader.oracle()
    rng = np.random.default_rng(seed)
    go_open, go_ext = gaps
    g, e = -(go_open + go_ext), -go_ext            # magnitudes of the first and of every further gap position
    assert 0 <= e <= g <= 2048
    n_flagged = n_exact = n_edge = 0
    with np.errstate(over="ignore"):               # (+inf in the making, far above the flag level, is part of the test)
        for case in range(28):
            # a table whose diagonal decides how fast scores grow: from BLOSUM-like to the int8 maximum
            diag = int(rng.choice([6, 17, 60, 100, 127, 127]))
            sub = rng.integers(-8, 4, size=(32, 32)).astype(np.int8)
            sub[np.arange(32), np.arange(32)] = diag
            sub[0, :] = sub[:, 0] = 0
            lq = int(rng.integers(30, 90))
            q = rng.integers(1, 27, size=lq).astype(np.int8)
            # a relative of the query: a prefix with substitutions, an insertion and a deletion
            d = q[:int(rng.integers(lq // 2, lq + 1))].copy()
            hit = rng.random(len(d)) < rng.choice([0.0, 0.0, 0.1, 0.3])
            d[hit] = rng.integers(1, 27, size=int(hit.sum()))
            cut = int(rng.integers(1, len(d)))
            d = np.concatenate([d[:cut], rng.integers(1, 27, size=int(rng.integers(0, 4))).astype(np.int8), d[cut + int(rng.integers(0, 3)):]])
            off = np.array([0, len(d)], dtype=np.uint64)
            truth = int(orc.score_db(q, d, off, sub, go_open, go_ext)[0])
            best = f16_cells_best(q.astype(np.int64), d.astype(np.int64), sub.astype(np.int32), g, e)
            flagged = bool(best >= F(2048.0))
            assert flagged == (truth >= 4096), (case, truth, float(best))
            if flagged:
                n_flagged += 1
            else:
                assert int(best) + 2048 == truth, (case, truth, float(best))
                n_exact += 1
                n_edge += truth > 2048             # exact in the upper half of the range, where the bias matters
    assert n_flagged >= 3 and n_exact >= 8, (n_flagged, n_exact, n_edge)


def test_f16_holds_the_integers_the_argument_needs():
    """The facts the exactness argument uses: every integer of [-2048, 2048] is a float16; a sum of two of them rounds
    monotonically (>= 2048 above the range, <= -2048 below it); 65504 is the largest finite value and 65504 + 16 rounds to +inf."""
    ints = np.arange(-2048, 2049)
    assert np.array_equal(ints.astype(F).astype(np.int64), ints)
    a = np.arange(-2048, 2049, 7).astype(F)
    b = np.arange(-2048, 2049, 5).astype(F)
    with np.errstate(over="ignore"):
        s = (a[:, None] + b[None, :]).astype(F).astype(np.float64)
        exact = a.astype(np.float64)[:, None] + b.astype(np.float64)[None, :]
        inside = np.abs(exact) <= 2048
        assert np.array_equal(s[inside], exact[inside])
        assert (s[exact > 2048] >= 2048).all() and (s[exact < -2048] <= -2048).all()
        assert F(65504.0) + F(15.0) == F(65504.0) and np.isinf(F(65504.0) + F(16.0))
        assert np.isfinite(F(-2048.0) - F(2048.0)) and F(-2048.0) + F(-65504.0) == -np.inf   # the transient t of a padding column
