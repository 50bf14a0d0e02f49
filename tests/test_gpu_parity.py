"""GPU (-m gpu): the HIP path, called through the C-ABI, against the golden vectors (every
expected value produced by the reference's own alignment_fill_matrices) and against the int32
oracle on seeded inputs.  Integer work: the bar is bit-exact, per database sequence."""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _setup(ctx, g):
    ctx.set_scoring(g["sub"], int(g["gaps"][0]), int(g["gaps"][1]))
    ctx.set_query(g["query"])
    _reset_options(ctx)


OPTIONS = ("force_bits", "engine", "cols_per_wave", "max_waves", "group_lanes", "long_split", "workgroups", "segment_blocks")
DEFAULT_ON = ("work_queue", "wide16", "f16", "qq", "last_pass")
DEFAULT_OFF = ("long_helps",)


def _reset_options(ctx):
    ctx.set_option("batch", 8)
    ctx.set_option("batch_blocks", 0)
    for k in OPTIONS:
        ctx.set_option(k, 0)
    for k in DEFAULT_ON:
        ctx.set_option(k, 1)
    for k in DEFAULT_OFF:
        ctx.set_option(k, 0)


def _truth(g):
    return g["oracle32"]


# (the systolic engine has int16 cells only: its rows run on the library's choice)
@pytest.mark.parametrize("engine,cells", [(1, "auto"), (2, "auto"), (2, "int16"), (2, "f16")])
@pytest.mark.parametrize("name", golden_names())
def test_golden_through_search(swg, ctx, name, engine, cells):
    """cells: which cells the diagonal engine's 16-bit fill runs on -- the library's choice, packed int16 only
    (option f16 = 0), or the packed-f16 cells whenever the gap scores allow (f16 = 2: exact below 4096, every
    sequence that reaches it flagged and re-scored in int32; the *_f16_boundary fixtures straddle that ceiling)."""
    g = load_golden(name)
    _setup(ctx, g)
    gaps_ok = g["gaps"][0] <= 0 and g["gaps"][1] <= 0
    ctx.set_option("engine", engine)         # 1 systolic, 2 diagonal (both arithmetic widths)
    ctx.set_option("f16", {"auto": 1, "int16": 0, "f16": 2}[cells])
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    scores, hits, st = ctx.search(db, k=10)
    ctx.set_option("f16", 1)
    assert np.array_equal(scores, _truth(g)), (name, st)
    if engine == 1 and gaps_ok and cells != "int16":
        # the systolic engine takes the packed-f16 cells exactly where no score of the search can reach their ceiling
        # (it has no flag-and-re-run route): longest sequence x largest table entry, or the query's best total
        lens = np.diff(g["offsets"].astype(np.int64))
        bound = min(int(g["sub"][g["query"].astype(np.int64)].max(axis=1).clip(min=0).sum()),
                    min(len(g["query"]), (int(lens.max()) + 3) // 4 * 4) * int(g["sub"].max()))
        assert st["cell_form"] == (2 if bound < 4096 else 0), (bound, st)
        assert bound >= 4096 or st["n_rescored"] == 0
    elif cells == "int16" or engine == 1 or not gaps_ok:
        assert st["cell_form"] in (0, 1)
    elif cells == "f16":
        assert st["cell_form"] == 2 and st["n_rescored"] == int((_truth(g) >= 4096).sum()), st
    if g["ref_valid"][0]:
        assert np.array_equal(scores, g["ref16"].astype(np.int32))
    best = sorted(((-int(s), i) for i, s in enumerate(_truth(g))))[:10]
    assert hits == [(-s, i) for s, i in best]
    assert st["path_bits"] == (16 if gaps_ok else 32)
    assert st["engine"] == engine
    assert st["cells"] == len(g["query"]) * len(g["flat"])
    db.close()


@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("name", golden_names())
def test_golden_forced_int32(swg, ctx, name, engine):
    g = load_golden(name)
    _setup(ctx, g)
    ctx.set_option("force_bits", 32)
    ctx.set_option("engine", engine)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    scores, _, st = ctx.search(db)
    assert st["path_bits"] == 32
    assert np.array_equal(scores, _truth(g)), name
    db.close()


@pytest.mark.parametrize("name", ["pam250_lq128", "blosum62_lq367", "pam250_partial_lanes",
                                  "blosum62_gap_0_1", "blosum62_gap_pos1_m3", "blosum62_query_bzx",
                                  "blosum62_gap_pos5_m1", "blosum62_gap_0_pos1"])
@pytest.mark.parametrize("route", ["device", "host"])
def test_golden_through_reference_shaped_batches(swg, ctx, orc, name, route):
    """swg_fill_batches16 replays exactly what alignment_fill_matrices receives.  route "device": the batches are
    uploaded as they are and the pair tokens built from them on the device, in buffers the context keeps between
    calls (called three times: the second re-uses them, the third -- the batches in reverse order, so not sorted by
    length -- as well); "host" (work_queue = 0 takes it): un-transposed and packed on the host as in round 2."""
    g = load_golden(name)
    _setup(ctx, g)
    if route == "host":
        ctx.set_option("work_queue", 0)
    batches = orc.db_to_batches16(g["flat"], g["offsets"])
    lanes = [int(v) for v in g["lanes"]]
    for attempt in range(3):
        order = list(range(len(batches)))[::-1] if attempt == 2 else list(range(len(batches)))
        out, secs = ctx.fill_batches16([(batches[b], lanes[b]) for b in order])
        for o, b in zip(out, order):
            assert np.array_equal(o, g["ref16"][b * 16:b * 16 + lanes[b]]), (name, route, attempt, b)
        assert secs > 0
    _reset_options(ctx)


def test_overflow_is_detected_and_rescored(swg, ctx):
    g = load_golden("pam250_overflow_w")
    _setup(ctx, g)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    # default: the wide form of the diagonal engine (scores to 65535), int32 only beyond that
    scores, _, st = ctx.search(db)
    assert np.array_equal(scores, g["oracle32"])
    assert st["path_bits"] == 16 and st["n_rescored"] == int((g["oracle32"] >= 65535).sum())
    assert (scores != g["ref16"]).any()      # the reference itself is wrong here (wraps)
    # plain int16 + int32 re-score of everything that reached 32767
    ctx.set_option("wide16", 0)
    scores, _, st = ctx.search(db)
    assert np.array_equal(scores, g["oracle32"])
    assert st["path_bits"] == 16 and st["n_rescored"] == int((g["oracle32"] >= 32767).sum())
    ctx.set_option("wide16", 1)
    db.close()


@pytest.mark.parametrize("cols,group,waves", [(24, 16, 16), (12, 32, 16), (8, 64, 16), (12, 64, 8), (16, 16, 4),
                                              (32, 64, 12), (8, 16, 4), (8, 32, 8), (24, 64, 4),
                                              (6, 64, 16), (10, 32, 8), (6, 16, 4), (20, 16, 4), (28, 16, 4),
                                              (14, 16, 4), (18, 32, 8), (22, 16, 4), (4, 64, 4), (2, 64, 4), (2, 16, 4),
                                              (23, 16, 4), (21, 32, 8), (9, 64, 4), (31, 16, 4), (25, 32, 4), (7, 16, 4),
                                              (13, 64, 8), (17, 16, 16), (3, 64, 4), (5, 32, 4), (26, 16, 4), (30, 32, 4)])
def test_diagonal_geometry_does_not_change_scores(swg, ctx, cols, group, waves):
    """Columns per lane, lanes per sequence pair, occupancy and the number of query passes
    (1 .. 24 here) are invisible in the result of the diagonal engine."""
    for name in ("blosum62_lq367", "blosum62_lq3000", "pam250_partial_lanes", "blosum62_tiny_db",
                 "pam250_overflow_w", "blosum62_f16_boundary", "pam250_f16_boundary"):
        g = load_golden(name)
        _setup(ctx, g)
        ctx.set_option("engine", 2)
        # (every other geometry also with the f16 cells forced: pam250_overflow_w then runs scores to 51 000 through them)
        ctx.set_option("f16", 2 if (cols + group // 16) % 2 else 1)
        ctx.set_option("cols_per_wave", cols)
        ctx.set_option("group_lanes", group)
        ctx.set_option("max_waves", waves)
        db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
        scores, _, st = ctx.search(db)
        assert np.array_equal(scores, g["oracle32"]), (name, cols, group, waves, st)
        assert (st["engine"], st["cols_per_wave"], st["group_lanes"], st["waves"]) == (2, cols, group, waves)
        assert st["passes"] == -(-len(g["query"]) // (cols * group))
        db.close()


@pytest.mark.parametrize("cols,group,waves", [(24, 16, 16), (12, 32, 16), (8, 64, 16), (12, 64, 8), (16, 16, 4),
                                              (32, 64, 12), (8, 16, 4), (8, 32, 8), (6, 64, 16), (10, 32, 8), (6, 16, 4),
                                              (20, 16, 4), (28, 16, 4), (14, 16, 4), (18, 32, 8), (22, 16, 4), (4, 64, 4),
                                              (2, 64, 4), (2, 16, 4), (23, 16, 4), (21, 32, 8), (9, 64, 4), (31, 16, 4),
                                              (25, 32, 4), (7, 16, 4), (13, 64, 8), (17, 16, 16), (3, 64, 4), (5, 32, 4),
                                              (26, 16, 4), (30, 32, 4), (12, 32, 0), (23, 16, 0)])
def test_q32_geometry_does_not_change_scores(swg, ctx, cols, group, waves):
    """force_bits = 32 together with a forced lane-group geometry: swg_diag32q_kernel at every K and group width,
    one pass and several (its form with edges), two classes and one, whole multi-pass launches cut into
    segments.  A geometry whose int32 profile does not fit LDS is replaced by the library (fewest lanes that
    do); the scores do not depend on which one ran.  (Round 2 only ever ran this kernel at the planner's pick.)"""
    for name in ("blosum62_lq367", "blosum62_lq3000", "pam250_overflow_w", "blosum62_f16_boundary", "blosum62_tiny_db"):
        g = load_golden(name)
        _setup(ctx, g)
        ctx.set_option("engine", 2)
        ctx.set_option("force_bits", 32)
        ctx.set_option("cols_per_wave", cols)
        ctx.set_option("group_lanes", group)
        ctx.set_option("max_waves", waves)
        if (cols + group) % 3 == 0:
            ctx.set_option("segment_blocks", (int(np.diff(g["offsets"]).max()) + 5) // 4 * 2 + 1)
        db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
        scores, hits, st = ctx.search(db, k=5)
        assert np.array_equal(scores, g["oracle32"]), (name, cols, group, waves, st)
        assert st["path_bits"] == 32 and st["engine"] == 2 and st["work_queue"] == 1
        assert st["passes"] >= -(-len(g["query"]) // (st["cols_per_wave"] * st["group_lanes"]))
        db.close()
    _reset_options(ctx)


@pytest.mark.parametrize("gaps", [(5, -1), (0, 1), (1, -3), (2, 2), (-4, 3)])
@pytest.mark.parametrize("opts", [{}, {"cols_per_wave": 7, "group_lanes": 16}, {"cols_per_wave": 28, "group_lanes": 32},
                                  {"cols_per_wave": 16, "group_lanes": 64}, {"cols_per_wave": 3, "group_lanes": 64},
                                  {"work_queue": 0}])
def test_positive_gap_scores_through_the_work_queue(swg, ctx, orc, gaps, opts):
    """Gap scores of any sign (the reference's CLI accepts positive ones, src/alignment_cmdline.c:255-267) run on the
    exact cells of the int32 work-queue kernel: the reference's recurrence term by term, one pass or several (a
    third edge value between the passes), any lane-group geometry up to 28 columns per lane; work_queue = 0 is
    the older bin-based kernel.  Every score against the int32 oracle: two query lengths, sequences of 1 .. 700
    residues, planted copies of the query."""
    sc = swg.load_scoring("BLOSUM62")
    go, ge = gaps
    for lq, n in ((333, 700), (1700, 150)):
        q = swg.synth_query(900 + lq, lq)
        flat, off, _ = swg.synth_db(77 + lq, n, query=q, fraction=0.05, subst=0.2, min_len=1, max_len=700)
        want = orc.score_db(q, flat, off, sc.table(), go, ge)
        ctx.set_scoring(sc, go, ge)
        ctx.set_query(q)
        _reset_options(ctx)
        for k, v in opts.items():
            ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        got, hits, st = ctx.search(db, k=8)
        assert np.array_equal(got, want), (gaps, opts, lq, st)
        assert hits == orc.topk(want, 8) and st["path_bits"] == 32 and st["engine"] == 2
        assert st["work_queue"] == (0 if "work_queue" in opts else 1)
        if "cols_per_wave" in opts:
            assert (st["cols_per_wave"], st["group_lanes"]) == (opts["cols_per_wave"], opts["group_lanes"])
            assert st["passes"] == -(-lq // (opts["cols_per_wave"] * opts["group_lanes"]))
        db.close()
    _reset_options(ctx)


@pytest.mark.parametrize("cols,maxw", [(32, 0), (16, 0), (48, 0), (24, 0), (32, 1), (32, 2), (16, 3), (32, 5), (24, 2)])
def test_geometry_does_not_change_scores(swg, ctx, cols, maxw):
    """Strip width, wave count and the number of query passes are invisible in the result."""
    for name in ("blosum62_lq367", "blosum62_lq3000", "pam250_partial_lanes", "blosum62_tiny_db"):
        g = load_golden(name)
        _setup(ctx, g)
        ctx.set_option("engine", 1)
        ctx.set_option("cols_per_wave", cols)
        ctx.set_option("max_waves", maxw)
        db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
        scores, _, st = ctx.search(db)
        assert np.array_equal(scores, g["oracle32"]), (name, cols, maxw, st)
        assert st["cols_per_wave"] == cols
        if maxw:
            assert st["waves"] <= maxw and st["passes"] == -(-len(g["query"]) // (cols * st["waves"])) \
                or st["passes"] >= 1
        db.close()


def test_multipass_int32_and_few_workgroups(swg, ctx):
    g = load_golden("blosum62_lq3000")
    _setup(ctx, g)
    ctx.set_option("force_bits", 32)
    ctx.set_option("engine", 1)
    ctx.set_option("max_waves", 3)
    ctx.set_option("workgroups", 1)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    scores, _, st = ctx.search(db)
    assert st["passes"] > 1 and st["workgroups"] == 1
    assert np.array_equal(scores, g["oracle32"])
    db.close()


def test_random_database_matches_oracle(swg, ctx, orc):
    """Config-2-shaped (PAM250, lq 367) but small enough for the scalar oracle."""
    sc = swg.load_scoring("PAM250")
    q = swg.synth_query(0x5EED0002, 367)
    flat, off = swg.synth_db(0x5EED0002, 3000)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    db = swg.Database(flat, off).upload(ctx)
    scores, hits, st = ctx.search(db, k=100)
    assert np.array_equal(scores, want)
    assert hits == orc.topk(want, 100)
    for engine, few, split in ((1, 0, 0), (2, 0, 0), (2, 3, 0), (2, 0, -1), (2, 0, 1500), (2, 0, 700), (2, 0, 40)):
        ctx.set_option("engine", engine)
        ctx.set_option("workgroups", few)        # few workgroups: long streams of many pairs
        ctx.set_option("long_split", split)      # off / forced: longest pairs as their own class
        sc_e, _, st_e = ctx.search(db)
        assert st_e["engine"] == engine and np.array_equal(sc_e, want), st_e
        if split > 0:
            n_long = int((np.diff(off.astype(np.int64))[::2] + 2 > max(split, 64)).sum())
            assert st_e["long_pairs"] == (n_long if n_long * 4 <= 1500 else 0)   # never more than a quarter
    _reset_options(ctx)
    # unsorted input order must not matter: shuffle, search, compare per sequence
    rng = np.random.default_rng(5)
    perm = rng.permutation(len(off) - 1)
    lens = np.diff(off.astype(np.int64))
    off2 = np.zeros_like(off)
    off2[1:] = np.cumsum(lens[perm])
    flat2 = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in perm])
    db2 = swg.Database(flat2, off2).upload(ctx)
    scores2, _, _ = ctx.search(db2)
    assert np.array_equal(scores2, want[perm])
    # shards: union of per-shard results == whole, merged top-K == global top-K
    merged = np.zeros_like(want)
    keys = []
    for r in range(3):
        s = swg.Database(flat, off, r, 3).upload(ctx)
        sc_r, hits_r, _ = ctx.search(s, k=100)
        o = s.order()
        merged[o] = sc_r[o]
        keys += [swg.hit_key(a, b) for a, b in hits_r]
        s.close()
    assert np.array_equal(merged, want)
    assert swg.topk_merge_keys(np.array(keys, dtype=np.uint64), 100) == orc.topk(want, 100)
    db.close()
    db2.close()


def test_autotuned_geometry_matches_oracle(swg, ctx, orc):
    """>= 4096 sequences: the first search times the best-ranked geometries on the device and
    keeps the fastest; every timed run and every later search must give the oracle's scores."""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(77, 200)
    flat, off = swg.synth_db(77, 6000, max_len=1500)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    db = swg.Database(flat, off).upload(ctx)
    first, _, st1 = ctx.search(db)
    again, hits, st2 = ctx.search(db, k=25)
    assert np.array_equal(first, want) and np.array_equal(again, want)
    assert hits == orc.topk(want, 25)
    geom = lambda st: (st["cols_per_wave"], st["group_lanes"], st["waves"], st["long_pairs"])
    assert geom(st1) == geom(st2)                 # the tuned plan is kept
    ctx.set_option("autotune", 0)
    model, _, _ = ctx.search(db)
    ctx.set_option("autotune", 1)
    assert np.array_equal(model, want)
    import os, tempfile
    with tempfile.TemporaryDirectory() as d:      # a database loaded from its packed file searches the same
        db.save(os.path.join(d, "db.swgdb"))
        db2 = swg.Database(path=os.path.join(d, "db.swgdb")).upload(ctx)
        loaded, _, _ = ctx.search(db2)
        assert np.array_equal(loaded, want)
        db2.close()
    q2 = swg.synth_query(78, 333)                 # another query length: tuned separately
    ctx.set_query(q2)
    other, _, _ = ctx.search(db)
    assert np.array_equal(other, orc.score_db(q2, flat, off, sc.table(), -2, -1))
    db.close()


def test_high_similarity_rescore_matches_oracle(swg, ctx, orc):
    """Config-5-shaped: planted near-copies of a long query saturate int16 and are re-scored."""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(0x5EED0005, 8192)
    flat, off, planted = swg.synth_db(0x5EED0005, 600, query=q, fraction=0.02, subst=0.05)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    assert planted >= 3 and (want > 32767).sum() == planted
    db = swg.Database(flat, off).upload(ctx)
    # the wide form holds these scores (< 65535): nothing is left for the int32 re-score
    scores, hits, st = ctx.search(db, k=20)
    assert np.array_equal(scores, want)
    assert st["n_rescored"] == 0 and st["passes"] > 1 and st["engine"] == 2
    assert hits == orc.topk(want, 20)
    ctx.set_option("wide16", 0)
    scores0, _, st0 = ctx.search(db)
    assert np.array_equal(scores0, want) and st0["n_rescored"] == planted
    ctx.set_option("wide16", 1)
    ctx.set_option("engine", 1)
    scores1, _, st1 = ctx.search(db)
    assert st1["engine"] == 1 and np.array_equal(scores1, want) and st1["n_rescored"] == planted
    db.close()


@pytest.mark.parametrize("f16", [1, 2])
@pytest.mark.parametrize("geom", [{}, {"cols_per_wave": 8, "group_lanes": 16, "max_waves": 4},
                                  {"cols_per_wave": 24, "group_lanes": 16, "max_waves": 8, "long_split": 900},
                                  {"cols_per_wave": 6, "group_lanes": 64, "max_waves": 4},
                                  {"cols_per_wave": 32, "group_lanes": 32, "max_waves": 4, "long_split": -1}])
def test_wide16_range_and_beyond(swg, ctx, orc, geom, f16):
    """Scores in every range at once: below 32767, between 32767 and 65535 (wide form, exact) and
    above 65535 (flagged by the wide form, re-scored in int32): a tryptophan-rich query (17 per
    match in PAM250) against copies of its prefixes, short decoys, and a one-pass geometry.
    f16 = 2 forces the packed-f16 cells on it: the copies' cells run past 65504 into +inf there, which must
    neither reach the pairs that follow them in their lane groups nor change any score (everything from 4096
    up is re-scored in int32)."""
    sc = swg.load_scoring("PAM250")
    rng = np.random.default_rng(33)
    w = swg.synth_query(1, 1)
    w[:] = 23  # 'W'
    q = np.concatenate([np.repeat(w, 4400), swg.synth_query(77, 100)])
    lens = [4500, 4200, 3000, 2500, 1900, 1000] + [int(v) for v in rng.integers(1, 400, size=500)]
    seqs = [q[:L].copy() if i < 6 else swg.synth_query(2000 + i, L) for i, L in enumerate(lens)]
    flat = np.concatenate(seqs)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    assert (want >= 65535).sum() >= 2 and ((want >= 32767) & (want < 65535)).sum() >= 2 and (want < 32767).sum() > 400
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    for k, v in geom.items():
        ctx.set_option(k, v)
    ctx.set_option("f16", f16)
    db = swg.Database(flat, off).upload(ctx)
    got, hits, st = ctx.search(db, k=10)
    assert np.array_equal(got, want), (geom, st)
    assert st["cell_form"] == 2 if f16 == 2 else st["cell_form"] in (1, 4)
    if st["cell_form"] == 4:    # (one class, default f16: the sequences under split_rows rows on the f16 cells)
        lens_a = np.asarray(lens)
        assert 0 < st["split_rows"] < 400 and st["cells_f16"] == len(q) * int(lens_a[_f16_part(lens_a, st["split_rows"])].sum())
        assert st["n_rescored"] == int((want >= 65535).sum()) + int((want[_f16_part(lens_a, st["split_rows"])] >= 4096).sum())
    else:
        assert st["n_rescored"] == int((want >= (4096 if f16 == 2 else 65535)).sum())
    assert st["engine"] == 2
    assert hits == orc.topk(want, 10)
    db.close()
    _reset_options(ctx)


def _f16_part(lens, split_rows):
    """Which sequences of a database the both-forms search (cell_form 4) ran on the f16 cells: by sorted rank (length
    descending, stable) the pairs (2i, 2i+1) whose longer member is under split_rows rows."""
    lens = np.asarray(lens)
    order = np.argsort(-lens.astype(np.int64), kind="stable")
    n_long = int((lens >= split_rows).sum())
    first = (n_long + 1) // 2 * 2
    part = np.zeros(len(lens), dtype=bool)
    part[order[first:]] = True
    return part


@pytest.mark.parametrize("lq,opts", [(1300, {}), (1300, {"segment_blocks": 700}), (2600, {"cols_per_wave": 16, "group_lanes": 16}),
                                     (900, {"long_split": 300})])
def test_f16_flagged_pairs_rerun_few_and_many(swg, ctx, orc, lq, opts):
    """What the f16 cells flag (scores from 4096 up) is run again on the int16 cells by the same work-queue kernel
    in list mode.  The re-run's geometry is a guess from the previous search of the database: 64 lanes per pair
    for a few flagged pairs (the first search knows nothing: that plan), the main fill's own for many (the second
    search, which has seen that more than a thousand pairs were flagged).  Both must give the oracle's scores;
    multi-pass re-runs cut into segments and two-class main fills included."""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(0xF16, lq)
    flat, off, planted = swg.synth_db(0xF16, 4200, query=q, fraction=0.7, subst=0.08, max_len=900)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    n_flag = int((want >= 4096).sum())
    assert (n_flag > 2200 or lq < 1000) and (want < 4096).sum() > 500, (n_flag, planted)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("f16", 2)       # (also past the veto: this database flags most of its rows)
    ctx.set_option("autotune", 0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    db = swg.Database(flat, off).upload(ctx)
    for attempt in range(2):
        got, hits, st = ctx.search(db, k=25)
        assert np.array_equal(got, want), (attempt, opts, st)
        assert st["cell_form"] == 2 and st["n_rescored"] == n_flag and hits == orc.topk(want, 25)
    db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


@pytest.mark.parametrize("name", ["pam250_lq128", "blosum62_lq367", "blosum62_tiny_db", "blosum62_lq1",
                                  "pam250_overflow_w", "blosum62_lq3000"])
def test_device_topk_matches_oracle_order(swg, ctx, orc, name):
    """Top-K selected on the device (scores stay in HBM): same hits, same tie order as the
    oracle's full sort, for K below, around and above the number of sequences, with heavy
    ties (tiny scores) and with scores beyond the histogram range (host fall-back)."""
    g = load_golden(name)
    _setup(ctx, g)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    n = len(g["offsets"]) - 1
    for k in (1, 7, 100, n - 1, n, n + 50, 4000):
        scores, hits, st = ctx.search(db, want_scores=False, k=k)
        assert scores is None
        assert hits == orc.topk(g["oracle32"], k), (name, k)
    db.close()


def test_searches_in_flight(swg, ctx, orc):
    """swg_search_begin/end: four searches queued back to back (two databases, full scores and
    device top-K mixed) each deliver their own results; a fifth begin is refused."""
    g1, g2 = load_golden("blosum62_lq367"), load_golden("pam250_partial_lanes")
    _setup(ctx, g1)
    db1 = swg.Database(g1["flat"], g1["offsets"]).upload(ctx)
    # same scoring/query for everything in flight; second database scored against g1's query
    db2 = swg.Database(g2["flat"], g2["offsets"]).upload(ctx)
    want2 = orc.score_db(g1["query"], g2["flat"], g2["offsets"], g1["sub"], -2, -1)
    t = [ctx.search_begin(db1, k=10), ctx.search_begin(db2, k=0, want_scores=True),
         ctx.search_begin(db1, k=5, want_scores=True), ctx.search_begin(db2, k=50)]
    with pytest.raises(swg.SwgError) as e:
        ctx.search_begin(db1, k=1)
    assert e.value.code == swg.SWG_ERR_STATE
    s3, h3, _ = ctx.search_end(t[2])          # out of order on purpose
    s1, h1, _ = ctx.search_end(t[0])
    s2, h2, _ = ctx.search_end(t[1])
    keys4, _ = ctx.search_end_keys(t[3])
    assert s1 is None and h1 == orc.topk(g1["oracle32"], 10)
    assert np.array_equal(s2, want2) and h2 == []
    assert np.array_equal(s3, g1["oracle32"]) and h3 == orc.topk(g1["oracle32"], 5)
    assert swg.topk_merge_keys(keys4, 50) == orc.topk(want2, 50)
    with pytest.raises(swg.SwgError):
        ctx.search_end(t[0])                  # ticket already redeemed
    again, _, _ = ctx.search(db1)             # slots are free again
    assert np.array_equal(again, g1["oracle32"])
    db1.close()
    db2.close()


def test_errors_are_codes_not_crashes(swg, ctx):
    g = load_golden("blosum62_lq1")
    _setup(ctx, g)
    with pytest.raises(swg.SwgError) as e:
        ctx.set_query(np.array([1, 0, 3], dtype=np.int8))
    assert e.value.code == swg.SWG_ERR_RESIDUE
    with pytest.raises(swg.SwgError):
        ctx.set_option("nonsense", 1)
    db = swg.Database(g["flat"], g["offsets"])          # packed but never uploaded
    with pytest.raises(swg.SwgError) as e:
        ctx.search(db)
    assert e.value.code == swg.SWG_ERR_STATE
    ctx.set_option("cols_per_wave", 33)                   # no such instantiation
    db.upload(ctx)
    with pytest.raises(swg.SwgError):
        ctx.search(db)
    ctx.set_option("cols_per_wave", 0)
    empty = swg.Database(np.zeros(0, np.int8), np.zeros(1, np.uint64)).upload(ctx)
    scores, hits, st = ctx.search(empty, k=5)
    assert scores.size == 0 and hits == [] and st["cells"] == 0
    # a segment of the multi-pass fill shorter than one pair of sequences cannot be launched: an error, not a hang
    g3 = load_golden("blosum62_lq3000")
    _setup(ctx, g3)
    db3 = swg.Database(g3["flat"], g3["offsets"]).upload(ctx)
    ctx.set_option("engine", 2)
    ctx.set_option("segment_blocks", 8)
    with pytest.raises(swg.SwgError) as e:
        ctx.search(db3)
    assert e.value.code == swg.SWG_ERR_ARG
    with pytest.raises(swg.SwgError):
        ctx.set_option("segment_blocks", 1 << 27)
    ctx.set_option("segment_blocks", 0)
    scores, _, _ = ctx.search(db3)
    assert np.array_equal(scores, g3["oracle32"])
    db3.close()
    _reset_options(ctx)


def test_full_size_config2_properties(swg, ctx, orc):
    """BASELINE config 2 at full size (367 aa vs 100k sequences, PAM250) is too big for the scalar
    oracle, so the HIP path is pinned by size-independent properties of the recurrence:
      * three independent code paths agree per sequence (diagonal int16, systolic int16, exact int32);
      * doubling every substitution score and both gap scores doubles every score;
      * reversing the query and every database sequence leaves every score unchanged;
      * a database copy of the query scores sum(S[q_i][q_i]);
      * top-K equals the best K of the full score vector (ties by lower index);
    plus the oracle itself on an evenly spaced sample of 600 sequences."""
    sc = swg.load_scoring("PAM250")
    tab = sc.table()
    q = swg.synth_query(0x5EED0002, 367)
    flat, off = swg.synth_db(0x5EED0002, 100000)
    lens = np.diff(off.astype(np.int64))
    # append the query itself as one more database sequence
    flat = np.concatenate([flat, q])
    off = np.concatenate([off, [off[-1] + len(q)]]).astype(np.uint64)
    n = len(off) - 1
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    db = swg.Database(flat, off).upload(ctx)
    base, hits, st = ctx.search(db, k=100)
    assert st["engine"] == 2 and st["path_bits"] == 16
    # the long pairs are a class of their own, launched on a second stream: measured to have run BESIDE the bulk
    assert st["long_pairs"] > 0 and st["classes_overlapped"] == 1
    # independent paths
    ctx.set_option("engine", 1)
    assert np.array_equal(ctx.search(db)[0], base)
    ctx.set_option("engine", 0)
    ctx.set_option("force_bits", 32)
    assert np.array_equal(ctx.search(db)[0], base)
    ctx.set_option("force_bits", 0)
    # self score and top-K
    assert base[n - 1] == sum(int(tab[a, a]) for a in q) and hits[0] == (int(base[n - 1]), n - 1)
    order = np.lexsort((np.arange(n), -base.astype(np.int64)))[:100]
    assert hits == [(int(base[i]), int(i)) for i in order]
    # oracle on a sample
    sample = np.linspace(0, n - 1, 600).astype(np.int64)
    s_off = np.zeros(len(sample) + 1, dtype=np.uint64)
    s_off[1:] = np.cumsum([int(off[i + 1] - off[i]) for i in sample])
    s_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in sample])
    assert np.array_equal(orc.score_db(q, s_flat, s_off, tab, -2, -1), base[sample])
    # linearity: 2*S, 2*gaps -> 2*scores
    ctx.set_scoring((tab.astype(np.int16) * 2).astype(np.int8), -4, -2)
    assert np.array_equal(ctx.search(db)[0], 2 * base)
    # reversal symmetry
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q[::-1].copy())
    rflat = np.concatenate([flat[int(off[i]):int(off[i + 1])][::-1] for i in range(n)])
    rdb = swg.Database(rflat, off).upload(ctx)
    assert np.array_equal(ctx.search(rdb)[0], base)
    rdb.close()
    db.close()


@pytest.mark.parametrize("cfg", ["config2", "config3", "config4_share", "config5"])
def test_full_size_databases_equal_the_reference_itself(swg, ctx, orc, cfg):
    """BASELINE configs 2 and 3 at full size, and one GPU's eighth of config 4, score for score against
    the REFERENCE's own alignment_fill_matrices (oracle/_ref: its alignment.c compiled from its
    sources, run under its OpenMP dispatch on the host; config 4: every 8th 16-record batch, the GPU
    still searches all 1.25 million sequences).  Config 5 (8192-aa query, 1 % near-copies scoring
    about 40 000): the reference's int16 lanes wrap above 32767, so every sequence the
    GPU scores at most 32767 must equal the reference and every third of the others the int32 oracle.
    The synthetic databases are emitted sorted by length and in multiples of 16, which is what the
    reference's packer requires."""
    if not orc.have_ref():
        pytest.skip("oracle/_ref was not built (needs the reference sources at build time)")
    lq, n, mat, seed, every = {"config2": (367, 100000, "PAM250", 0x5EED0002, 1),
                               "config3": (500, 570000, "BLOSUM62", 0x5EED0003, 1),
                               "config4_share": (3000, 1250000, "BLOSUM62", 0x5EED0004, 8),
                               "config5": (8192, 100000, "BLOSUM62", 0x5EED0005, 1)}[cfg]
    sc = swg.load_scoring(mat)
    tab = sc.table()
    q = swg.synth_query(seed, lq)
    if cfg == "config5":
        flat, off, planted = swg.synth_db(seed, n, query=q, fraction=0.01, subst=0.05)
        assert planted > 0
    else:
        flat, off = swg.synth_db(seed, n)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    db = swg.Database(flat, off).upload(ctx)
    scores, _, st = ctx.search(db)
    # (f16 cells: exact below 4096, the few sequences that reach it are re-scored)
    assert st["engine"] == 2 and st["path_bits"] == 16 and (st["n_rescored"] == 0 or cfg == "config5" or st["cell_form"] == 2)
    if cfg == "config5":
        # the same search on the path the configuration is named after: plain int16 (sticks at 32767), every
        # flagged sequence re-scored in int32 -- at full size, all 100 000 scores against the wide form's
        ctx.set_option("wide16", 0)
        scores16, _, st16 = ctx.search(db)
        ctx.set_option("wide16", 1)
        assert st["n_rescored"] == 0                        # 8192 columns cannot pass 65535: the wide form is exact
        assert st16["n_rescored"] == planted == int((scores > 32767).sum())
        assert np.array_equal(scores16, scores)
    db.close()
    groups = np.arange(0, n // 16, every)
    lens = np.diff(off.astype(np.int64))
    assert all(lens[g * 16] == lens[g * 16:g * 16 + 16].max() for g in groups[:: max(1, len(groups) // 2000)])
    batches = []
    for g in groups:
        o = off[g * 16:g * 16 + 17].astype(np.int64)
        b = np.full((int(o[1] - o[0]), 16), 31, dtype=np.int8)          # '*' filler, src/alignment_cmdline.c:444-450
        for l in range(16):
            b[:int(o[l + 1] - o[l]), l] = flat[int(o[l]):int(o[l + 1])]
        batches.append(b)
    ref, _ = orc.ref_batches(q, batches, tab, -2, -1, threads=int(swg.lib.swg_host_threads()))
    idx = (groups[:, None] * 16 + np.arange(16)[None, :]).ravel()
    if cfg != "config5":
        assert np.array_equal(ref.astype(np.int32).ravel(), scores[idx]), cfg
        return
    small = scores[idx] <= 32767
    assert np.array_equal(ref.astype(np.int32).ravel()[small], scores[idx][small])
    big = idx[~small]
    assert 0 < len(big) <= 2 * planted and scores[big].min() > 32767
    big = big[::3]          # the scalar oracle needs 0.25 s per 8192 x 8192 pair and thread
    b_off = np.zeros(len(big) + 1, dtype=np.uint64)
    b_off[1:] = np.cumsum(lens[big])
    b_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in big])
    assert np.array_equal(orc.score_db(q, b_flat, b_off, tab, -2, -1), scores[big])


def test_config4_share_with_relatives_takes_the_flag_and_rerun_route(swg, ctx, orc):
    """bench.py's block "4_relatives": one GPU's share of config 4 with a seeded 0.5 % of the sequences relatives of
    the 3000-aa query at 30-70 % identity -- scores about 3 800 .. 10 000, above the f16 cells' ceiling (4096) and
    below int16's.  The f16 fill flags them, the flagged pairs run again on int16 cells (list mode); first search and
    steady state return the same scores, every planted sequence's batch and every 16th other batch equal the
    REFERENCE's own alignment_fill_matrices (scores below 32767: its int16 lanes are exact there), and the cells the
    fill ran on are reported."""
    if not orc.have_ref():
        pytest.skip("oracle/_ref was not built (needs the reference sources at build time)")
    lq, n, seed = 3000, 1250000, 0x5EED0007
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table()
    q = swg.synth_query(seed, lq)
    flat, off, planted = swg.synth_db(seed, n, query=q, fraction=0.005, subst=0.3, subst_hi=0.7)
    assert 5000 < planted < 7500
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    db = swg.Database(flat, off).upload(ctx)
    first, hits1, st1 = ctx.search(db, k=50)
    assert st1["path_bits"] == 16 and st1["cell_form"] == 2            # nothing known about this database yet: the f16 cells
    n_hi = int((first >= 4096).sum())
    assert st1["n_rescored"] == n_hi and 0.8 * planted < n_hi <= planted + 16   # (a few relatives at 30 % stay below 4096)
    assert first.max() < 32767
    again, hits2, st2 = ctx.search(db, k=50)                           # steady state: whatever the library learnt
    assert np.array_equal(again, first) and hits2 == hits1
    assert st2["cell_form"] in (0, 2) and (st2["cell_form"] == 0 or st2["n_rescored"] == n_hi)
    ctx.set_option("f16", 0)                                           # the int16 cells alone, as a third opinion on all of it
    plain, _, st3 = ctx.search(db)
    assert st3["cell_form"] == 0 and np.array_equal(plain, first)
    db.close()
    _reset_options(ctx)
    lens = np.diff(off.astype(np.int64))
    hot = np.unique(np.nonzero(first >= 4096)[0] // 16)
    groups = np.unique(np.concatenate([hot, np.arange(0, n // 16, 16)]))
    batches = []
    for g in groups:
        o = off[g * 16:g * 16 + 17].astype(np.int64)
        b = np.full((int(o[1] - o[0]), 16), 31, dtype=np.int8)          # '*' filler, src/alignment_cmdline.c:444-450
        for l in range(16):
            b[:int(o[l + 1] - o[l]), l] = flat[int(o[l]):int(o[l + 1])]
        batches.append(b)
    ref, _ = orc.ref_batches(q, batches, tab, -2, -1, threads=int(swg.lib.swg_host_threads()))
    idx = (groups[:, None] * 16 + np.arange(16)[None, :]).ravel()
    assert np.array_equal(ref.astype(np.int32).ravel(), first[idx])
    assert hits1 == orc.topk(first, 50) and hits1[0][0] > 9000


def test_failed_host_allocation_in_search_end_is_a_status(swg, ctx, orc):
    """include/swg.h:9-12 on the GPU: the fall-back key vector of swg_search_end (taken when the device top-K cannot be
    used: here k beyond its candidate capacity) throws std::bad_alloc through the test hook -- the call returns
    SWG_ERR_NOMEM, the ticket is spent, and the context goes on working."""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(77, 120)
    flat, off = swg.synth_db(77, 9000, max_len=400)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    db = swg.Database(flat, off).upload(ctx)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    t = ctx.search_begin(db, 6000)                  # k > half the candidate capacity: the host selects from all scores
    swg.lib.swg_debug_fail_alloc(3)
    with pytest.raises(swg.SwgError) as e:
        ctx.search_end(t)
    swg.lib.swg_debug_fail_alloc(0)
    assert e.value.code == swg.SWG_ERR_NOMEM and "swg_search_end" in str(e.value)
    with pytest.raises(swg.SwgError):               # the ticket is spent
        ctx.search_end(t)
    got, hits, _ = ctx.search(db, k=6000)
    assert np.array_equal(got, want) and hits == orc.topk(want, 6000)
    db.close()


def test_long_tail_database(swg, ctx, orc):
    """Swiss-Prot-like tail: a 35,000-residue sequence among short ones (one very long pair,
    odd sequence count, lengths 1 and 2 present), query longer than one pass of some geometries."""
    sc = swg.load_scoring("BLOSUM62")
    rng = np.random.default_rng(9)
    q = swg.synth_query(5, 500)
    lens = [35000, 1, 2, 7000] + [int(v) for v in rng.integers(20, 900, size=297)]
    seqs = [swg.synth_query(100 + i, L) for i, L in enumerate(lens)]
    flat = np.concatenate(seqs)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    for opts in ({}, {"engine": 1}, {"force_bits": 32}, {"cols_per_wave": 8, "group_lanes": 16, "max_waves": 4},
                 {"long_split": 3000}, {"long_split": -1, "cols_per_wave": 6, "group_lanes": 64, "max_waves": 8}):
        _reset_options(ctx)
        for k, v in opts.items():
            ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        got, hits, st = ctx.search(db, k=5)
        assert np.array_equal(got, want), (opts, st)
        assert hits == orc.topk(want, 5)
        db.close()
    _reset_options(ctx)


def test_work_queue_variants_agree(swg, ctx, orc):
    """The diagonal engine with pairs off the work queue (default), with the long class going on
    with the bulk's pairs or not, and with static streams: same scores.  The database mixes very
    short sequences (pairs of one or two token blocks, shorter than a lane group is wide), a long
    tail, and fewer / more pairs than lane groups and queue shards."""
    sc = swg.load_scoring("PAM250")
    rng = np.random.default_rng(21)
    q = swg.synth_query(8, 367)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    shapes = {
        "tiny": [1, 2, 3],                                           # 2 pairs: most queue shards empty
        "short": [int(v) for v in rng.integers(1, 12, size=3000)],   # pairs far shorter than 16..64 rows
        "tiniest": [1, 2] * 30000,     # every pair is one token block: up to 18 pairs inside a 64-lane group
        "mixed": [4000, 2500, 1, 1, 2] + [int(v) for v in rng.integers(5, 700, size=6000)],
    }
    for name, lens in shapes.items():
        seqs = [swg.synth_query(1000 + i, L) for i, L in enumerate(lens)]
        flat = np.concatenate(seqs)
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        want = orc.score_db(q, flat, off, sc.table(), -2, -1)
        for opts in ({}, {"work_queue": 0}, {"long_helps": 1}, {"long_split": 300}, {"long_split": 300, "long_helps": 1},
                     {"long_split": -1}, {"cols_per_wave": 6, "group_lanes": 64, "max_waves": 4},
                     {"cols_per_wave": 12, "group_lanes": 32, "max_waves": 8, "long_split": 500},
                     {"cols_per_wave": 24, "group_lanes": 16, "max_waves": 4, "workgroups": 3}):
            _reset_options(ctx)
            ctx.set_option("engine", 2)
            for k, v in opts.items():
                ctx.set_option(k, v)
            db = swg.Database(flat, off).upload(ctx)
            got, hits, st = ctx.search(db, k=7)
            assert np.array_equal(got, want), (name, opts, st)
            assert hits == orc.topk(want, 7)
            assert st["work_queue"] == opts.get("work_queue", 1) and st["passes"] == 1
            # a second search on the same resident database re-arms the queue counters
            got2, _, _ = ctx.search(db, k=0)
            assert np.array_equal(got2, want), (name, opts, "second search")
            db.close()
    _reset_options(ctx)


def test_pairs_claimed_by_the_batch(swg, ctx, orc):
    """Round 4: where pairs are short one queue request claims `batch` consecutive pairs (three zones per shard: single
    long pairs, whole batches, the last pairs single again).  Batch sizes, thresholds from "no pair counts as short" to
    "every pair does", a long class in front (the range does not begin at pair 0), several passes cut into segments
    (ranges that begin and end anywhere), fewer pairs than one batch per shard, and batches of queries: always the
    oracle's scores.  Peptide-like and tiny pairs, odd counts."""
    sc = swg.load_scoring("BLOSUM62")
    rng = np.random.default_rng(77)
    shapes = {
        "peptides": [int(v) for v in rng.integers(20, 41, size=30001)],
        "tiniest": [1, 2, 3] * 9000 + [1],
        "few": [int(v) for v in rng.integers(5, 30, size=37)],              # less than one batch per shard
        "mixed": [3000, 1800, 900] + [int(v) for v in rng.integers(1, 400, size=9000)],
    }
    ctx.set_scoring(sc, -2, -1)
    for name, lens in shapes.items():
        seqs = [swg.synth_query(5000 + i, L) for i, L in enumerate(lens)]
        flat = np.concatenate(seqs)
        off = np.zeros(len(lens) + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        for lq in (30, 700):
            q = swg.synth_query(9 + lq, lq)
            ctx.set_query(q)
            want = orc.score_db(q, flat, off, sc.table(), -2, -1)
            for opts in ({}, {"batch": 0}, {"batch": 2}, {"batch": 5, "batch_blocks": 3}, {"batch_blocks": 100000},
                         {"batch_blocks": 100000, "long_split": 200}, {"batch_blocks": 64, "long_split": 200, "long_helps": 1},
                         {"cols_per_wave": 8, "group_lanes": 16, "max_waves": 4, "segment_blocks": 4000, "batch_blocks": 100000},
                         {"cols_per_wave": 4, "group_lanes": 64, "max_waves": 4, "batch": 8, "batch_blocks": 12},
                         {"f16": 0, "batch_blocks": 9}):
                if lq == 30 and "segment_blocks" in opts:
                    continue                                                     # (several passes need a query beyond G * K columns)
                _reset_options(ctx)
                ctx.set_option("autotune", 0)
                ctx.set_option("engine", 2)        # (left alone the cost model gives peptides to the systolic engine)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                db = swg.Database(flat, off).upload(ctx)
                got, hits, st = ctx.search(db, k=5)
                assert st["engine"] == 2 and st["work_queue"] == 1
                assert np.array_equal(got, want), (name, lq, opts, st)
                assert hits == orc.topk(want, 5)
                db.close()
        if name in ("peptides", "tiniest"):                                      # batches of queries share the zones
            _reset_options(ctx)
            qs = [swg.synth_query(600 + i, L) for i, L in enumerate((30, 64, 17))]
            db = swg.Database(flat, off).upload(ctx)
            for opts in ({"engine": 2}, {"engine": 2, "batch_blocks": 100000}, {"engine": 2, "qq": 0, "batch": 3}, {}):
                # (engine 2: the batch on the lane groups; left alone such a database goes query by query to the systolic engine)
                _reset_options(ctx)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                got, _, st = ctx.search_multi(db, qs)
                assert not opts or st["engine"] == 2, (name, opts, st)
                for i, q in enumerate(qs):
                    assert np.array_equal(got[i], orc.score_db(q, flat, off, sc.table(), -2, -1)), (name, opts, i, st)
            db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


def test_cost_model_gives_peptides_to_the_systolic_engine(swg, ctx, orc):
    """Round 4: a database of short sequences of near-equal length is what the systolic engine is good at (no reset rows,
    no flags, nothing per pair), and the cost model -- not only the autotuner -- says so: with autotune off, 300 000
    peptides of 20-40 residues run on engine 1 for short queries and on the lane groups for a long one, BASELINE's length
    distribution stays on the lane groups, and the scores are the oracle's either way.  The systolic fill is at least a
    quarter faster than the lane groups' on the same resident database."""
    sc = swg.load_scoring("BLOSUM62")
    ctx.set_scoring(sc, -2, -1)
    flat, off = swg.synth_db(0xBEEF, 300000, median=29.0, sigma_ln=0.25, min_len=20, max_len=40)
    lens = np.diff(off.astype(np.int64))
    sample = np.linspace(0, len(lens) - 1, 6000).astype(np.int64)
    s_off = np.zeros(len(sample) + 1, dtype=np.uint64)
    s_off[1:] = np.cumsum(lens[sample])
    s_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in sample])
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    db = swg.Database(flat, off).upload(ctx)
    for lq, engine in ((30, 1), (128, 1), (600, 2)):
        q = swg.synth_query(40 + lq, lq)
        ctx.set_query(q)
        want = orc.score_db(q, s_flat, s_off, sc.table(), -2, -1)
        ctx.set_option("engine", 0)
        got, hits, st = ctx.search(db, k=20)
        assert st["engine"] == engine, (lq, st)
        assert np.array_equal(got[sample], want), (lq, st)
        order = np.lexsort((np.arange(len(got)), -got.astype(np.int64)))[:20]
        assert hits == [(int(got[i]), int(i)) for i in order]
        if engine == 1:
            fills = {}
            for e in (1, 2):
                ctx.set_option("engine", e)
                other, _, st2 = ctx.search(db)
                assert st2["engine"] == e and np.array_equal(other, got), (lq, e)
                fills[e] = min(ctx.search(db, want_scores=False)[2]["fill_ms"] for _ in range(3))
            assert fills[1] < 0.8 * fills[2], (lq, fills)
    db.close()
    flat, off = swg.synth_db(0x5EED0002, 20000)
    db = swg.Database(flat, off).upload(ctx)
    ctx.set_option("engine", 0)
    ctx.set_query(swg.synth_query(1, 128))
    assert ctx.search(db)[2]["engine"] == 2
    db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


def test_multipass_through_the_work_queue(swg, ctx, orc):
    """A query of several passes: one work-queue launch per pass, edges handed from launch to
    launch.  Same scores as fixed streams and as the systolic engine, for every lane-group width,
    also when pairs are much shorter than a lane group (edges of idle and reset rows)."""
    g = load_golden("blosum62_lq3000")
    _setup(ctx, g)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    for opts in ({}, {"cols_per_wave": 16, "group_lanes": 16, "max_waves": 4},
                 {"cols_per_wave": 12, "group_lanes": 32, "max_waves": 8},
                 {"cols_per_wave": 8, "group_lanes": 64, "max_waves": 4},
                 {"cols_per_wave": 32, "group_lanes": 16, "max_waves": 4},
                 {"cols_per_wave": 6, "group_lanes": 16, "max_waves": 4},
                 # launches cut into segments of consecutive pairs (what a database of more than 2^26
                 # token blocks gets), also with the long pairs as their own class
                 {"cols_per_wave": 32, "group_lanes": 16, "max_waves": 4, "segment_blocks": 300},
                 {"cols_per_wave": 12, "group_lanes": 32, "max_waves": 8, "segment_blocks": 150, "long_split": 1},
                 {"cols_per_wave": 8, "group_lanes": 64, "max_waves": 4, "segment_blocks": 200}):
        _reset_options(ctx)
        ctx.set_option("engine", 2)
        for k, v in opts.items():
            ctx.set_option(k, v)
        scores, _, st = ctx.search(db)
        assert np.array_equal(scores, g["oracle32"]), (opts, st)
        assert st["work_queue"] == 1 and st["passes"] > 1
        # one launch per pass, times the segments when the database is cut into them
        assert (st["fill_launches"] > st["passes"]) if "segment_blocks" in opts else (st["fill_launches"] == st["passes"]), st
        assert st["fill_launches"] % st["passes"] == 0
        ctx.set_option("work_queue", 0)
        scores0, _, st0 = ctx.search(db)
        assert np.array_equal(scores0, g["oracle32"]) and st0["work_queue"] == 0 and st0["passes"] == st["passes"]
    db.close()
    # short and tiny sequences against a long query
    sc = swg.load_scoring("BLOSUM62")
    rng = np.random.default_rng(5)
    q = swg.synth_query(91, 1500)
    lens = [3000, 1, 2, 3] + [int(v) for v in rng.integers(1, 60, size=4000)]
    seqs = [swg.synth_query(500 + i, L) for i, L in enumerate(lens)]
    flat = np.concatenate(seqs)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    for opts in ({"cols_per_wave": 8, "group_lanes": 16, "max_waves": 4}, {"cols_per_wave": 6, "group_lanes": 64, "max_waves": 4},
                 {"cols_per_wave": 8, "group_lanes": 16, "max_waves": 4, "segment_blocks": 800}):
        _reset_options(ctx)
        ctx.set_option("engine", 2)
        for k, v in opts.items():
            ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        got, _, st = ctx.search(db)
        assert np.array_equal(got, want), (opts, st)
        assert st["work_queue"] == 1 and st["passes"] > 1
        db.close()
    _reset_options(ctx)


def test_group_of_several_contexts_on_one_device(swg, orc):
    """the in-process multi-GPU path had never run with n > 1.  A group that names this box's one
    device two, four and five times holds that many contexts and shards: `swg_group_load` sorts the database ONCE and cuts,
    builds and uploads the shards side by side (one host thread each), `swg_group_search` queues every shard's search
    before it awaits any, scores land by original index, the top-K lists are merged, alignments are routed to the shard
    that holds the sequence.  (RCCL refuses a device named twice, so the keys are merged on the host here -- which is what
    the max-all-reduce of disjoint segments computes; the collective itself runs in test_group_with_rccl_merge.)"""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(33, 220)
    flat, off = swg.synth_db(33, 3001, max_len=900)           # 24 bins: uneven shares for 5 shards, an odd sequence count
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    for n in (2, 4, 5):
        sorts = swg.lib.swg_debug_sort_count()
        grp = swg.Group([0] * n)
        grp.set_option("autotune", 0)
        grp.set_scoring(sc, -2, -1)
        grp.set_query(q)
        grp.load(flat, off)
        assert swg.lib.swg_debug_sort_count() == sorts + 1
        scores, hits, stats = grp.search(k=60)
        assert np.array_equal(scores, want) and hits == orc.topk(want, 60), n
        assert len(stats) == n and sum(st["cells"] for st in stats) == len(q) * len(flat)
        assert all(st["cells"] > 0 for st in stats)
        none, hits2, _ = grp.search(want_scores=False, k=9)     # a second search on the resident shards, hits only
        assert none is None and hits2 == orc.topk(want, 9)
        als = grp.align_hits(hits[:8])                          # the eight best hits live on several shards
        for a, (s_, i_) in zip(als, hits[:8]):
            sc_, co, ops = orc.pair_trace(q, flat[int(off[i_]):int(off[i_ + 1])], sc.table(), -2, -1)
            assert (a["score"], a["index"], a["ops"]) == (s_, i_, ops) and sc_ == s_
        # another query against the same resident shards
        q2 = swg.synth_query(34, 90)
        grp.set_query(q2)
        scores2, hits3, _ = grp.search(k=5)
        want2 = orc.score_db(q2, flat, off, sc.table(), -2, -1)
        assert np.array_equal(scores2, want2) and hits3 == orc.topk(want2, 5)
        grp.set_query(q)
        grp.close()
    # fewer bins than contexts: some shards are empty
    flat1, off1 = swg.synth_db(35, 200, max_len=300)            # 2 bins, 4 contexts
    grp = swg.Group([0, 0, 0, 0])
    grp.set_scoring(sc, -2, -1)
    grp.set_query(q)
    grp.load(flat1, off1)
    scores, hits, stats = grp.search(k=10)
    want1 = orc.score_db(q, flat1, off1, sc.table(), -2, -1)
    assert np.array_equal(scores, want1) and hits == orc.topk(want1, 10)
    assert sorted(st["cells"] > 0 for st in stats) == [False, False, True, True]
    grp.close()


def test_group_with_rccl_merge(swg, orc):
    """swg_group on this box's one GPU with the collective forced: shard packing, concurrent
    begin/end, the RCCL max-all-reduce of the hit keys and the final merge all run; results are
    the oracle's.  (More devices only add segments to the same all-reduce.)"""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(21, 150)
    flat, off = swg.synth_db(21, 1500, max_len=700)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    for force in (False, True):
        grp = swg.Group([0], force_collective=force)
        grp.set_scoring(sc, -2, -1)
        grp.set_query(q)
        grp.load(flat, off)
        scores, hits, stats = grp.search(k=40)
        assert np.array_equal(scores, want) and hits == orc.topk(want, 40)
        none, hits2, _ = grp.search(want_scores=False, k=7)
        assert none is None and hits2 == orc.topk(want, 7)
        assert len(stats) == 1 and stats[0]["cells"] == len(q) * len(flat)
        als = grp.align_hits(hits[:6])
        for a, (s_, i_) in zip(als, hits[:6]):
            sc_, co, ops = orc.pair_trace(q, flat[int(off[i_]):int(off[i_ + 1])], sc.table(), -2, -1)
            assert (a["score"], a["index"], a["ops"]) == (s_, i_, ops) and sc_ == s_
            assert (a["q_begin"], a["q_end"], a["d_begin"], a["d_end"]) == co
        with pytest.raises(swg.SwgError):
            grp.align_hits([(0, len(off) - 1)])
        grp.close()


def test_queries_streamed_against_resident_database(swg, ctx, orc):
    """Many queries, one resident database: each query is set and queued while the previous
    search is still in flight (many-to-many use); every result is the oracle's."""
    sc = swg.load_scoring("BLOSUM45")
    flat, off = swg.synth_db(31, 2500, max_len=900)
    ctx.set_scoring(sc, -3, -1)
    _reset_options(ctx)
    db = swg.Database(flat, off).upload(ctx)
    queries = [swg.synth_query(40 + i, L) for i, L in enumerate((33, 700, 128, 1, 257, 64, 1500, 90))]
    pending, got = None, []
    for q in queries:
        ctx.set_query(q)
        t = ctx.search_begin(db, k=10, want_scores=True)
        if pending is not None:
            got.append(ctx.search_end(pending))
        pending = t
    got.append(ctx.search_end(pending))
    for q, (scores, hits, st) in zip(queries, got):
        want = orc.score_db(q, flat, off, sc.table(), -3, -1)
        assert np.array_equal(scores, want), len(q)
        assert hits == orc.topk(want, 10) and st["cells"] == len(q) * len(flat)
    db.close()


def test_randomised_soak():
    """tests/fuzz_gpu.py for half a minute: random databases, query lengths, tables, gap scores and
    engine options through the C ABI against the int32 oracle (851 cases passed in the 7-minute run
    of round 1)."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "tests", "fuzz_gpu.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(30.0, 11) == 0


def test_randomised_soak_of_query_batches():
    """tests/fuzz_multi_gpu.py for twenty seconds: random batches of queries (equal, mixed and several-pass lengths,
    relatives planted, cell forms and two-queries-per-lane drawn, with and without the score array) through
    swg_search_multi, every score and hit list against the int32 oracle."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_multi_gpu", os.path.join(ROOT, "tests", "fuzz_multi_gpu.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.main(20.0, 3) == 0


def _db_with_empties(swg, seed, n, n_empty, max_len):
    """Random database with n_empty zero-length records mixed in at seeded positions."""
    flat, off = swg.synth_db(seed, n, min_len=1, max_len=max_len)
    lens = np.diff(off.astype(np.int64))
    rng = np.random.default_rng(seed)
    lens[rng.choice(n, size=n_empty, replace=False)] = 0
    perm = rng.permutation(n)                              # and not in sorted order either
    seqs = [flat[int(off[i]):int(off[i]) + int(lens[i])] for i in perm]
    off2 = np.zeros(n + 1, dtype=np.uint64)
    off2[1:] = np.cumsum([len(s) for s in seqs])
    flat2 = np.concatenate(seqs) if int(off2[-1]) else np.zeros(0, np.int8)
    return flat2, off2


@pytest.mark.parametrize("n,n_empty,max_len", [(2000, 0, 700), (1999, 37, 300), (257, 200, 9), (5, 5, 5), (1, 0, 3)])
def test_device_built_tokens_equal_the_host_builder(swg, ctx, n, n_empty, max_len):
    """The pair tokens are built on the device from the uploaded residue bytes; the host restatement of
    the same layout (swg_diag_host.cpp: write_pair_tokens) must give the same image bit for bit --
    odd counts (a pair without a second sequence), empty sequences and empty pairs included."""
    flat, off = _db_with_empties(swg, 1234 + n, n, n_empty, max_len)
    db = swg.Database(flat, off).upload(ctx)
    dev = db.debug_pair_tokens(ctx, from_host=False)
    host = db.debug_pair_tokens(ctx, from_host=True)
    assert dev.size == host.size and dev.size > 0
    assert np.array_equal(dev, host), int(np.nonzero(dev != host)[0][0])
    # every pair carries exactly one last-row flag: the tail lane pops one pair id per flag
    assert int(((dev & 0x20000) != 0).sum()) == (db.count + 1) // 2
    assert ((dev & 0xFFF40707) == 0).all()                 # residue bytes are index << 3, flags are bits 16, 17 and 19
    # two reset rows open every pair; the second one says so (bit 19)
    assert int(((dev & 0x10000) != 0).sum()) == 2 * ((db.count + 1) // 2)
    assert np.array_equal((dev & 0x80000) != 0, np.roll((dev & 0x90000) == 0x10000, 1))
    db.close()


def test_empty_records_mixed_in(swg, ctx, orc):
    """Zero-length records score 0 and must not disturb their neighbours' pair ids (an empty pair still
    hands its id to the tail lane), in every engine and through the long class."""
    sc = swg.load_scoring("BLOSUM62")
    q = swg.synth_query(5, 90)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    for seed, n, n_empty, max_len in ((1, 3000, 301, 400), (2, 700, 650, 40), (3, 130, 129, 2000)):
        flat, off = _db_with_empties(swg, seed, n, n_empty, max_len)
        want = orc.score_db(q, flat, off, sc.table(), -2, -1)
        assert (want[np.diff(off.astype(np.int64)) == 0] == 0).all()
        db = swg.Database(flat, off).upload(ctx)
        for opts in ({}, {"engine": 1}, {"work_queue": 0}, {"long_split": 64}, {"force_bits": 32},
                     {"group_lanes": 64, "cols_per_wave": 2}):
            _reset_options(ctx)
            for k, v in opts.items():
                ctx.set_option(k, v)
            got, hits, st = ctx.search(db, k=20)
            assert np.array_equal(got, want), (seed, opts, st)
            assert hits == orc.topk(want, 20)
        db.close()
    _reset_options(ctx)


def test_config5_every_sequence_similar(swg, ctx, orc):
    """SURVEY 8d's stress variant of config 5: 100 000 sequences, ALL of them full-length copies of the 8192-aa
    query with 5 % point substitutions (scores about 40 000): every sequence saturates plain int16.  The wide
    form (exact to 65535) and the int16 + flagged int32 re-score path must agree on all 100 000 scores; a
    seeded sample is compared with the int32 oracle (the whole set is 6.7e12 cells: minutes on the host)."""
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table()
    lq, n = 8192, 100000
    q = swg.synth_query(0x5EED0005, lq)
    flat, off, planted = swg.synth_db(0x5EED0005, n, query=q, fraction=1.0, subst=0.05)
    assert planted == n and int(off[-1]) == n * lq
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    db = swg.Database(flat, off).upload(ctx)
    wide, hits, st = ctx.search(db, k=100)
    assert st["path_bits"] == 16 and st["n_rescored"] == 0 and wide.min() > 32767 and wide.max() < 65535
    ctx.set_option("wide16", 0)
    plain, _, st16 = ctx.search(db)
    ctx.set_option("wide16", 1)
    assert st16["n_rescored"] == n and np.array_equal(plain, wide)
    order = np.lexsort((np.arange(n), -wide.astype(np.int64)))[:100]
    assert hits == [(int(wide[i]), int(i)) for i in order]
    sample = np.random.default_rng(55).choice(n, size=48, replace=False)
    s_off = np.arange(len(sample) + 1, dtype=np.uint64) * lq
    s_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in sample])
    assert np.array_equal(orc.score_db(q, s_flat, s_off, tab, -2, -1), wide[sample])
    db.close()


@pytest.mark.parametrize("opts", [{}, {"segment_blocks": 3000}, {"cols_per_wave": 24, "group_lanes": 32, "max_waves": 4, "long_split": -1}])
@pytest.mark.parametrize("long_ones,wide16", [(True, 1), (False, 1), (True, 0)])
def test_both_16bit_forms_in_one_search(swg, ctx, orc, opts, long_ones, wide16):
    """A query that can score beyond 32767 (wide form) against a database of mostly short sequences: those under
    4096 * lq / qbound rows run on the f16 cells, the longer ones on the wide form, in launches of their own per pass
    (swg_stats.cell_form 4).  The threshold is an expectation, so the database also holds what defeats it: runs of
    400-500 tryptophans (11 per match, 4400-5500 for fewer rows than the threshold), which the f16 cells must flag and the
    wide form score again; near-copies of the whole query (beyond 32767, on the wide form from the start).  Without any
    long sequence (long_ones False) the whole search runs on the f16 cells and what they flag on the wide form.  All
    scores against the oracle, twice (the second search plans the re-run from what the first one saw), also with the
    passes cut into segments and with a forced geometry.  wide16 = 0: the long part on plain int16 cells, everything
    from 32767 up -- the f16 part's flagged pairs that saturate their int16 re-run too -- re-scored in int32."""
    sc = swg.load_scoring("BLOSUM62")
    # (an entry no sequence here uses, so that "longest sequence x largest entry" does not cap the score bound below
    # 32767 when every sequence is short: the search without long ones must still plan for the wide range)
    sc.sub[21][21] = 127
    rng = np.random.default_rng(404)
    w = swg.synth_query(1, 1)
    w[:] = 23  # 'W'
    q = np.concatenate([swg.synth_query(91, 1250), np.repeat(w, 2000), swg.synth_query(92, 1250)])
    assert not (q == 21).any()
    lens = [int(v) for v in rng.integers(20, 500, size=900)]
    seqs = [swg.synth_query(3000 + i, L) for i, L in enumerate(lens)]
    for i, L in enumerate((400, 430, 470, 500, 380)):                   # tryptophan runs: short, but beyond the f16 ceiling
        seqs[50 + 7 * i] = np.repeat(w, L)
        lens[50 + 7 * i] = L
    if long_ones:
        for L in (4500, 4400, 3000, 2200, 1500, 900, 700, 650):         # (prefixes of the query: the W run is inside from 1250 on)
            seqs.append(q[:L].copy())
            lens.append(L)
        for i, L in enumerate(int(v) for v in rng.integers(560, 1400, size=40)):
            seqs.append(swg.synth_query(5000 + i, L))
            lens.append(L)
    flat = np.concatenate(seqs)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    assert ((want >= 4096) & (np.asarray(lens) <= 500)).sum() >= 4 and (not long_ones or (want > 32767).sum() >= 2)
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    ctx.set_option("wide16", wide16)
    for k, v in opts.items():
        ctx.set_option(k, v)
    db = swg.Database(flat, off).upload(ctx)
    for attempt in range(2):
        got, hits, st = ctx.search(db, k=20)
        assert np.array_equal(got, want), (attempt, opts, st, np.nonzero(got != want)[0][:8])
        assert hits == orc.topk(want, 20) and st["passes"] > 1
        if long_ones:
            part = _f16_part(np.asarray(lens), st["split_rows"])
            assert st["cell_form"] == (4 if wide16 else 5) and 500 < st["split_rows"] < 560 and st["cells_f16"] == len(q) * int(np.asarray(lens)[part].sum()), st
            assert st["n_rescored"] == int((want[part] >= 4096).sum()) + int((want >= (65535 if wide16 else 32767)).sum())
        else:
            assert st["cell_form"] == 2 and st["n_rescored"] == int((want >= 4096).sum()), st
    db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


@pytest.mark.parametrize("geom,last", [((32, 16), 28), ((23, 32), 2), ((16, 64), 15), ((31, 16), 2), ((25, 16), 13)])
@pytest.mark.parametrize("f16", [0, 2])
def test_last_pass_has_its_own_geometry(swg, ctx, geom, last, f16):
    """A query of several passes: the last pass runs the instantiation with the fewest columns per lane that cover
    what is left of the query (3000 columns as 5 x 512 + 28 x 16, as 4 x 736 + 2 x 32, as 7 x 400 + 13 x 16 ...) -- a
    different kernel and profile layout from the other passes', the same scores (reference-produced golden, lq 3000),
    with the option off too."""
    g = load_golden("blosum62_lq3000")
    _setup(ctx, g)
    K, G = geom
    lq = len(g["query"])
    npass = -(-lq // (K * G))
    want_last = -(-(lq - (npass - 1) * K * G) // G)
    want_last = max(2, want_last)
    ctx.set_option("autotune", 0)
    ctx.set_option("engine", 2)
    ctx.set_option("cols_per_wave", K)
    ctx.set_option("group_lanes", G)
    ctx.set_option("max_waves", 4)
    ctx.set_option("long_split", -1)
    ctx.set_option("f16", f16)
    db = swg.Database(g["flat"], g["offsets"]).upload(ctx)
    for on in (1, 0):
        ctx.set_option("last_pass", on)
        scores, _, st = ctx.search(db, k=5)
        assert np.array_equal(scores, _truth(g)) and np.array_equal(scores, g["ref16"].astype(np.int32)), (geom, on, st)
        assert st["passes"] == npass and st["cols_per_wave"] == K and st["group_lanes"] == G, st
        assert st["last_pass_cols"] == (want_last if on and want_last < K else 0), (st, want_last)
        assert not on or st["last_pass_cols"] == last
    db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


@pytest.mark.parametrize("wide16", [1, 0])
def test_titin_sized_query(swg, ctx, orc, wide16):
    """The longest proteins known are about 35 000 residues: a 36 000-aa query (18 passes of 2048 columns) against
    prefixes of itself -- scores far beyond 65535, so the 16-bit forms flag them and the int32 work-queue kernel scores
    them over its own passes --, sequences either side of the both-forms cut, and one sequence longer than the query.
    wide16 = 0: plain int16 cells first, everything from 32767 up re-scored."""
    sc = swg.load_scoring("BLOSUM62")
    lq = 36000
    q = swg.synth_query(0x717, lq)
    rng = np.random.default_rng(717)
    lens = [36000, 30011, 14000, 9000, 6500, 2000] + [int(v) for v in rng.integers(1, 1500, size=150)] + [40000]
    seqs = [q[:L].copy() if i < 6 else swg.synth_query(7000 + i, L) for i, L in enumerate(lens)]
    seqs[3][::7] = 1                                  # (a relative with substitutions, not a copy)
    flat = np.concatenate(seqs)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    want = orc.score_db(q, flat, off, sc.table(), -2, -1)
    assert (want >= 65535).sum() >= 2 and ((want >= 32767) & (want < 65535)).sum() >= 1 and (want < 4096).sum() > 100
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    ctx.set_option("wide16", wide16)
    db = swg.Database(flat, off).upload(ctx)
    for attempt in range(2):
        got, hits, st = ctx.search(db, k=8)
        assert np.array_equal(got, want), (attempt, st, np.nonzero(got != want)[0][:8])
        assert hits == orc.topk(want, 8) and st["passes"] >= 18 and st["path_bits"] == 16
        assert st["cell_form"] == (4 if wide16 else 5), st   # (both forms; the long part on the wide form / on plain int16 cells)
    db.close()
    _reset_options(ctx)
    ctx.set_option("autotune", 1)


def test_config5_one_gpu_share_with_flagged_rescore(swg, ctx, orc):
    """Config 5 as BASELINE names it ("forcing 16->32-bit rescore") at one GPU's share of SURVEY 8d's shape: 1.25
    million sequences, 1 % of them near-copies of the 8192-aa query (what bench.py's config-5 block and its `rescore`
    leg run).  The wide int16 form (exact to 65535) and plain int16 + the flagged int32 re-score must agree on all
    1 250 000 scores, the re-score must have run on exactly the planted sequences, and seeded samples are compared
    with the REFERENCE (oracle/_ref: 16-record batches that hold no planted sequence -- its int16 lanes wrap above
    32767) and with the int32 oracle (planted ones)."""
    lq, n = 8192, 1250000
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table()
    q = swg.synth_query(0x5EED0005, lq)
    flat, off, planted = swg.synth_db(0x5EED0005, n, query=q, fraction=0.01, subst=0.05)
    assert 11000 < planted < 14000
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    db = swg.Database(flat, off).upload(ctx)
    wide, hits, st = ctx.search(db, k=100)
    # (default: both forms -- the near-copies and everything else of 4096 * lq / qbound rows or more on the wide form,
    # the rest on the f16 cells, which flag nothing here)
    assert st["path_bits"] == 16 and st["cell_form"] == 4 and st["n_rescored"] == 0 and st["passes"] > 1
    lens_all = np.diff(off.astype(np.int64))
    assert 600 < st["split_rows"] < 900 and st["cells_f16"] == lq * int(lens_all[_f16_part(lens_all, st["split_rows"])].sum())
    assert 0.5 * st["cells"] < st["cells_f16"] < 0.8 * st["cells"] and 0 < st["fill_f16_ms"] < st["fill_ms"]
    ctx.set_option("f16", 0)
    only_wide, hits_w, st_w = ctx.search(db, k=100)
    assert st_w["cell_form"] == 1 and st_w["n_rescored"] == 0 and np.array_equal(only_wide, wide) and hits_w == hits
    ctx.set_option("wide16", 0)
    plain, hits16, st16 = ctx.search(db, k=100)
    ctx.set_option("f16", 1)
    ctx.set_option("wide16", 1)
    ctx.set_option("autotune", 1)
    db.close()
    big = wide > 32767
    assert int(big.sum()) == planted and st16["n_rescored"] == planted and st16["rescore_ms"] > 0 and st16["cell_form"] == 0
    assert np.array_equal(plain, wide) and hits16 == hits
    order = np.lexsort((np.arange(n), -wide.astype(np.int64)))[:100]
    assert hits == [(int(wide[i]), int(i)) for i in order]
    rng = np.random.default_rng(5)
    sample = rng.choice(np.nonzero(big)[0], size=24, replace=False)
    s_off = np.zeros(len(sample) + 1, dtype=np.uint64)
    s_off[1:] = np.cumsum([int(off[i + 1] - off[i]) for i in sample])
    s_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for i in sample])
    assert np.array_equal(orc.score_db(q, s_flat, s_off, tab, -2, -1), wide[sample])
    if not orc.have_ref():
        pytest.skip("oracle/_ref was not built (needs the reference sources at build time): planted sample checked only")
    lens = np.diff(off.astype(np.int64))
    srt = np.argsort(-lens, kind="stable")                  # the reference wants its input sorted by length (A.7-5)
    groups = [g for g in rng.choice(n // 16, size=400, replace=False) if not big[srt[g * 16:g * 16 + 16]].any()][:200]
    batches = []
    for g in groups:
        ids = srt[g * 16:g * 16 + 16]
        b = np.full((int(lens[ids[0]]), 16), 31, dtype=np.int8)
        for l, i in enumerate(ids):
            b[:int(lens[i]), l] = flat[int(off[i]):int(off[i + 1])]
        batches.append(b)
    ref, _ = orc.ref_batches(q, batches, tab, -2, -1, threads=int(swg.lib.swg_host_threads()))
    idx = np.concatenate([srt[g * 16:g * 16 + 16] for g in groups])
    assert len(groups) >= 100 and np.array_equal(ref.astype(np.int32).ravel(), wide[idx])


def test_config4_whole_database_on_one_gpu(swg, ctx, orc):
    """Config 4 as BASELINE names it: ONE 10M-sequence database (3000-aa query), all of it on this GPU (what
    bench.py's scaling reference runs).  A seeded sample of 16-record batches spread over the whole length
    range against the REFERENCE's alignment_fill_matrices (oracle/_ref), per entry as test/tests.py:105-133
    compares; the top-100 against the int32 oracle on the candidates and against the full score vector."""
    lq, n = 3000, 10000000
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table()
    q = swg.synth_query(0x5EED0004, lq)
    sh = swg.synth_db_shard(0x5EED0004, n, 0, 1)           # the generator bench.py --gpus N uses, one shard
    flat, off = sh["flat"], sh["offsets"]
    assert np.array_equal(sh["index"], np.arange(n, dtype=np.uint32)) and sh["residues_total"] == int(off[-1])
    ctx.set_scoring(sc, -2, -1)
    ctx.set_query(q)
    _reset_options(ctx)
    ctx.set_option("autotune", 0)
    db = swg.Database(flat, off, index=sh["index"], n_total=n).upload(ctx)
    scores, hits, st = ctx.search(db, k=100)
    ctx.set_option("autotune", 1)
    db.close()
    assert st["path_bits"] == 16 and st["cells"] == lq * int(off[-1])
    assert st["n_rescored"] == (int((scores >= 4096).sum()) if st["cell_form"] == 2 else 0)
    # top-100: consistent with the score vector, and every candidate's score equal to the oracle's
    order = np.lexsort((np.arange(n), -scores.astype(np.int64)))[:100]
    assert hits == [(int(scores[i]), int(i)) for i in order]
    c_off = np.zeros(101, dtype=np.uint64)
    c_off[1:] = np.cumsum([int(off[i + 1] - off[i]) for _, i in hits])
    c_flat = np.concatenate([flat[int(off[i]):int(off[i + 1])] for _, i in hits])
    assert np.array_equal(orc.score_db(q, c_flat, c_off, tab, -2, -1), np.array([s for s, _ in hits], dtype=np.int32))
    if not orc.have_ref():
        pytest.skip("oracle/_ref was not built (needs the reference sources at build time): candidates checked only")
    groups = np.unique(np.concatenate([np.random.default_rng(4).choice(n // 16, size=500, replace=False),
                                       np.arange(0, 40), np.arange(n // 16 - 40, n // 16)]))      # and both ends
    batches = []
    for g in groups:
        o = off[g * 16:g * 16 + 17].astype(np.int64)
        assert o[1] - o[0] == np.diff(o).max()               # sorted: the first of 16 is the longest
        b = np.full((int(o[1] - o[0]), 16), 31, dtype=np.int8)
        for l in range(16):
            b[:int(o[l + 1] - o[l]), l] = flat[int(o[l]):int(o[l + 1])]
        batches.append(b)
    ref, _ = orc.ref_batches(q, batches, tab, -2, -1, threads=int(swg.lib.swg_host_threads()))
    idx = (groups[:, None] * 16 + np.arange(16)[None, :]).ravel()
    assert np.array_equal(ref.astype(np.int32).ravel(), scores[idx])


def test_many_queries_in_one_pass(swg, ctx, orc):
    """swg_search_multi: a batch of queries against one resident database in one launch per class must give,
    query by query, what swg_search gives (and the oracle) -- small database (one query cannot fill the GPU),
    queries of different lengths in one batch, a long class, more queries than fit one launch, and the
    batches that fall back to one search after another (a query of several passes; scores that can pass
    32767; positive gap scores)."""
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table()
    ctx.set_scoring(sc, -2, -1)
    _reset_options(ctx)
    ctx.set_query(swg.synth_query(1, 50))                              # the context's own query must survive
    flat, off = swg.synth_db(0x5EED0001, 1024)                         # config 1's database
    db = swg.Database(flat, off).upload(ctx)
    own, _, _ = ctx.search(db)
    qs = [swg.synth_query(100 + i, 128) for i in range(64)]
    got, hits, st = ctx.search_multi(db, qs, k=10)
    assert st["engine"] == 2 and st["work_queue"] == 1 and st["passes"] == 1 and st["cells"] == 64 * 128 * len(flat)
    for i in (0, 1, 31, 63):
        want = orc.score_db(qs[i], flat, off, tab, -2, -1)
        assert np.array_equal(got[i], want), i
        assert hits[i] == orc.topk(want, 10)
    for i, q in enumerate(qs):                                         # every query against the single-query path
        ctx.set_query(q)
        one, h1, _ = ctx.search(db, k=10)
        assert np.array_equal(got[i], one) and hits[i] == h1, i
    ctx.set_query(swg.synth_query(1, 50))
    assert np.array_equal(ctx.search(db)[0], own)
    # mixed lengths (the geometry is the longest query's), 300 queries (two launches of at most 256)
    rng = np.random.default_rng(9)
    qs = [swg.synth_query(500 + i, int(rng.integers(1, 200))) for i in range(300)]
    got, hits, st = ctx.search_multi(db, qs, k=3)
    for i in (0, 7, 255, 256, 299):
        want = orc.score_db(qs[i], flat, off, tab, -2, -1)
        assert np.array_equal(got[i], want) and hits[i] == orc.topk(want, 3), (i, len(qs[i]))
    db.close()
    # a larger database with a long tail (bulk and long class side by side), 8 queries
    flat, off = swg.synth_db(31, 20000)
    db = swg.Database(flat, off).upload(ctx)
    qs = [swg.synth_query(900 + i, 367) for i in range(8)]
    got, _, st = ctx.search_multi(db, qs)
    for i in (0, 7):
        assert np.array_equal(got[i], orc.score_db(qs[i], flat, off, tab, -2, -1)), i
    # fall-backs: several passes, possible saturation, positive gap scores -- same results, one by one
    flat2, off2 = swg.synth_db(32, 600, max_len=400)
    db2 = swg.Database(flat2, off2).upload(ctx)
    long_qs = [swg.synth_query(950 + i, 2500) for i in range(2)] + [swg.synth_query(960, 90)]
    got, hits, st = ctx.search_multi(db2, long_qs, k=5)
    for i, q in enumerate(long_qs):
        want = orc.score_db(q, flat2, off2, tab, -2, -1)
        assert np.array_equal(got[i], want) and hits[i] == orc.topk(want, 5), i
    ctx.set_scoring(sc, 1, -2)
    got, _, st = ctx.search_multi(db2, [long_qs[2], qs[0][:60]])
    assert st["path_bits"] == 32
    assert np.array_equal(got[0], orc.score_db(long_qs[2], flat2, off2, tab, 1, -2))
    assert np.array_equal(got[1], orc.score_db(qs[0][:60], flat2, off2, tab, 1, -2))
    ctx.set_scoring(sc, -2, -1)
    with pytest.raises(swg.SwgError):
        ctx.search_multi(db2, [qs[0], np.zeros(0, np.int8)])            # an empty query is an error, not a crash
    db.close()
    db2.close()


@pytest.mark.parametrize("opts,form", [({}, 3), ({"qq": 0}, 2), ({"f16": 0}, 0)])
def test_many_queries_cell_forms(swg, ctx, orc, opts, form):
    """The three forms a batch of queries can run on -- two QUERIES per lane on the f16 cells (the default where no
    query of the batch can score 4096), two sequences per lane on the f16 cells, on the int16 cells -- give the
    oracle's scores: odd numbers of queries and of sequences, a 1-residue query, a database with a long class."""
    sc = swg.load_scoring("PAM250")
    tab = sc.table()
    ctx.set_scoring(sc, -3, -1)
    ctx.set_query(swg.synth_query(1, 50))
    for n, max_len, lens in ((1023, 900, (128, 1, 77, 300, 299, 45, 128)), (9001, 5000, (367, 200, 366))):
        flat, off = swg.synth_db(70 + n, n, max_len=max_len)
        _reset_options(ctx)
        for k, v in opts.items():
            ctx.set_option(k, v)
        db = swg.Database(flat, off).upload(ctx)
        qs = [swg.synth_query(300 + i, L) for i, L in enumerate(lens)]
        got, hits, st = ctx.search_multi(db, qs, k=4)
        # (two queries per lane need twice the LDS per column: where the planner's wide lane groups for a small
        # database leave no room for that, the batch stays on two sequences per lane)
        assert st["cell_form"] == form or (form == 3 and n == 1023 and st["cell_form"] == 2), st
        for i, q in enumerate(qs):
            want = orc.score_db(q, flat, off, tab, -3, -1)
            assert np.array_equal(got[i], want) and hits[i] == orc.topk(want, 4), (opts, n, i)
        # hits only: the batch's top-K is selected on the device (three launches for all queries) -- the same lists,
        # also where many sequences tie at the k-th score (k = 300 reaches deep into a 1-residue query's ties)
        for k in (1, 4, 300):
            none, hits_dev, _ = ctx.search_multi(db, qs, k=k, want_scores=False)
            assert none is None
            for i in range(len(qs)):
                assert hits_dev[i] == orc.topk(got[i], k), (opts, n, i, k)
        db.close()
    _reset_options(ctx)


def test_many_queries_fill_the_gpu_where_one_cannot(swg, ctx):
    """Config 1's shape (128 aa vs 1024 sequences) is one 5000-row chain per launch: about 80 GCUPS.  64 such
    queries in one pass must run at least ten times that."""
    sc = swg.load_scoring("BLOSUM62")
    ctx.set_scoring(sc, -2, -1)
    _reset_options(ctx)
    flat, off = swg.synth_db(0x5EED0001, 1024)
    db = swg.Database(flat, off).upload(ctx)
    qs = [swg.synth_query(100 + i, 128) for i in range(64)]
    ctx.search_multi(db, qs, want_scores=False)                          # warm-up: tokens, code objects
    best = 0.0
    for _ in range(3):
        _, _, st = ctx.search_multi(db, qs, want_scores=False)
        best = max(best, st["cells"] / (st["fill_ms"] * 1e-3) / 1e9)
    ctx.set_query(qs[0])
    ctx.search(db, want_scores=False)
    _, _, one = ctx.search(db, want_scores=False)
    single = one["cells"] / (one["fill_ms"] * 1e-3) / 1e9
    print("one query %.1f GCUPS, 64 queries in one pass %.1f GCUPS" % (single, best))
    assert best >= 800.0 and best >= 8.0 * single
    db.close()
