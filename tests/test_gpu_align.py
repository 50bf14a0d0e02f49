"""GPU (-m gpu): alignments of reported hits (swg_align_hits) through the C-ABI.

What pins what: the reference prints scores only (its fork removed the traceback), so an
alignment PATH has no reference output -- "parity unpinned" for the path itself.  Its SCORE is
pinned like every other: the path's substitution and gap scores must add up to the score the
reference's own alignment_fill_matrices produced for the pair (golden vectors), and the path and
its coordinates must equal the int32 oracle's, which follows the same documented tie rules."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _check(orc, q, flat, off, sub, go, ge, als, expect_scores=None):
    for a in als:
        d = flat[int(off[a["index"]]):int(off[a["index"] + 1])]
        sc, co, ops = orc.pair_trace(q, d, sub, go, ge)
        assert a["score"] == sc, a
        if expect_scores is not None:
            assert a["score"] == int(expect_scores[a["index"]]), a
        assert (a["q_begin"], a["q_end"], a["d_begin"], a["d_end"]) == co, (a, co)
        assert a["ops"] == ops, a["index"]
        assert a["n_ops"] == len(ops)
        assert orc.path_score(q, d, sub, go, ge, co, a["ops"]) == a["score"]
        if a["score"] > 0:
            assert a["ops"][-1] == "M"      # the score is a maximum of the match state (src/alignment.c:133)


@pytest.mark.parametrize("name", golden_names())
def test_alignments_of_the_golden_hits(swg, ctx, orc, name):
    """Top hits of every golden database: path = oracle's, and it re-scores to the value the
    reference's own fill produced for that pair (the int32 oracle's where the reference wraps)."""
    g = load_golden(name)
    go, ge = int(g["gaps"][0]), int(g["gaps"][1])
    ctx.set_scoring(g["sub"], go, ge)
    ctx.set_query(g["query"])
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    scores, hits, _ = ctx.search(db, k=12)
    als = ctx.align_hits(db, hits)
    assert [a["index"] for a in als] == [i for _, i in hits]
    assert [a["score"] for a in als] == [s for s, _ in hits]
    ref = g["ref16"].astype(np.int32) if g["ref_valid"][0] else g["oracle32"]
    _check(orc, g["query"], g["flat"], g["offsets"], g["sub"], go, ge, als, ref)
    db.close()


def test_alignments_with_planted_similarity(swg, ctx, orc):
    """Near-copies of the query with substitutions and indels: long paths with all three kinds of
    step, for every sequence of a small database (not only the top hits), plus the no-ops form."""
    rng = np.random.default_rng(77)
    sc = swg.load_scoring("BLOSUM62").table()
    q = swg.synth_query(5, 300)
    seqs = []
    for t in range(48):
        if t % 3 == 2:
            s = rng.integers(1, 21, size=int(rng.integers(1, 500))).astype(np.int8)
            s = np.array([b"ACDEFGHIKLMNPQRSTVWY"[v - 1] - 64 for v in s], dtype=np.int8)
        else:
            a = int(rng.integers(0, 150)); b = int(rng.integers(a + 40, 301))
            s = q[a:b].copy()
            for _ in range(int(rng.integers(0, 6))):     # indels
                p = int(rng.integers(1, len(s) - 1)); L = int(rng.integers(1, 5))
                if rng.random() < 0.5:
                    s = np.delete(s, slice(p, p + L))
                else:
                    s = np.insert(s, p, q[rng.integers(0, 300, size=L)])
            m = rng.random(len(s)) < 0.1
            s[m] = q[rng.integers(0, 300, size=int(m.sum()))]
            s = np.concatenate([q[rng.integers(0, 300, size=int(rng.integers(0, 30)))], s,
                                q[rng.integers(0, 300, size=int(rng.integers(0, 30)))]]).astype(np.int8)
        seqs.append(s)
    flat = np.concatenate(seqs); off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    for go, ge in ((-2, -1), (-11, -1), (0, -2)):
        ctx.set_scoring(sc, go, ge)
        ctx.set_query(q)
        db = swg.Database(flat, off).upload(ctx)
        scores, _, _ = ctx.search(db)
        every = [(int(scores[i]), i) for i in range(len(seqs))]
        als = ctx.align_hits(db, every)
        _check(orc, q, flat, off, sc, go, ge, als, scores)
        assert any("I" in a["ops"] for a in als) and any("D" in a["ops"] for a in als)
        bare = ctx.align_hits(db, every[:5], want_ops=False)
        assert [(b["score"], b["q_begin"], b["q_end"], b["d_begin"], b["d_end"], b["n_ops"]) for b in bare] == \
               [(a["score"], a["q_begin"], a["q_end"], a["d_begin"], a["d_end"], a["n_ops"]) for a in als[:5]]
        db.close()


def test_alignment_of_a_long_pair_and_a_wide_query(swg, ctx, orc):
    """More columns than one sweep of the workgroup covers (query 2000 > 256 threads) and the
    longest sequence of the database (up to 5000 residues): coordinates and path still equal the
    oracle's.  (Scores beyond int16: the pam250_overflow_w golden above.)"""
    sc = swg.load_scoring("PAM250").table()
    q = swg.synth_query(9, 2000)
    flat, off, _ = swg.synth_db(0x5EED0009, 64, query=q, fraction=0.2, subst=0.05, max_len=5000)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    db = swg.Database(flat, off).upload(ctx)
    scores, hits, _ = ctx.search(db, k=6)
    lens = np.diff(off.astype(np.int64))
    longest = int(np.argmax(lens))
    want = hits + [(int(scores[longest]), longest)]
    als = ctx.align_hits(db, want)
    _check(orc, q, flat, off, sc, -2, -1, als, scores)
    db.close()


def test_alignment_argument_errors(swg, ctx):
    sc = swg.load_scoring("BLOSUM62").table()
    q = swg.synth_query(1, 50)
    flat, off = swg.synth_db(3, 300)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    db = swg.Database(flat, off).upload(ctx)
    assert ctx.align_hits(db, []) == []
    with pytest.raises(swg.SwgError) as e:
        ctx.align_hits(db, [(0, 300)])            # no such sequence
    assert e.value.code == swg.SWG_ERR_ARG
    with pytest.raises(swg.SwgError) as e:
        ctx.align_hits(db, [(0, 0)], ops_stride=1)   # too short for any path
    assert e.value.code == swg.SWG_ERR_ARG
    # a shard holds only its own sequences
    half = swg.Database(flat, off, shard_rank=1, shard_count=2).upload(ctx)
    mine = set(int(i) for i in half.order() if i != 0xFFFFFFFF)
    other = next(i for i in range(300) if i not in mine)
    with pytest.raises(swg.SwgError):
        ctx.align_hits(half, [(0, other)])
    ok = ctx.align_hits(half, [(0, next(iter(mine)))])
    assert ok[0]["index"] in mine
    half.close()
    db.close()
