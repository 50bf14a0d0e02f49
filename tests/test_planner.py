"""The cost model's choices for the BASELINE shapes, on the host (swg_debug_plan: no device involved).
bench.py relies on the model alone -- every rank of a multi-GPU run must arrive at the same geometry
without timing anything -- so the geometries DESIGN.md section 6 reports are pinned here: a change to the
planner that moves one of them has to be a decision, not an accident."""
import numpy as np
import pytest

import swg_loader


@pytest.fixture(scope="module")
def swg():
    return swg_loader.load()


def _plan(swg, seed, n, lq, **kw):
    if kw:
        q = swg.synth_query(seed, lq)
        flat, off, _ = swg.synth_db(seed, n, query=q, **kw)
    else:
        flat, off = swg.synth_db(seed, n)
    db = swg.Database(flat, off)
    try:
        return db.debug_plan(lq)
    finally:
        db.close()


def test_config2_bulk_and_long_class(swg):
    p = _plan(swg, 0x5EED0002, 100000, 367)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"]) == (2, 23, 16, 4, 1), p
    assert (p["long_K"], p["long_G"], p["long_W"]) == (6, 64, 4), p
    assert 1500 <= p["long_pairs"] <= 3000 and p["long_workgroups"] == 256, p      # one long-class wavefront per SIMD


def test_config3_headline(swg):
    p = _plan(swg, 0x5EED0003, 570000, 500)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"], p["workgroups"]) == (1, 32, 16, 4, 1, 768), p


def test_config4_share_six_passes(swg):
    p = _plan(swg, 0x5EED0004, 1250000, 3000)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"], p["workgroups"]) == (1, 32, 16, 4, 6, 768), p
    assert p["last_pass_cols"] == 28, p         # 3000 = 5 x 512 + 440: the last pass with 28 columns per lane (448)


def test_config5_long_query_near_copies(swg):
    # the planted 8 200-row pairs are the chain of every pass: 64 lanes each, two wavefronts per SIMD
    p = _plan(swg, 0x5EED0005, 100000, 8192, fraction=0.01, subst=0.05)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"]) == (1, 32, 64, 8, 4), p
    assert p["last_pass_cols"] == 0, p          # 8192 = 4 x 2048: nothing left over


def test_last_pass_columns(swg):
    """The last pass of a query of several passes covers what is left with the fewest columns per lane (a single pass
    and a two-class plan have none of their own)."""
    flat, off = swg.synth_db(0x5EED0003, 200000)
    db = swg.Database(flat, off)
    try:
        for lq in (367, 500, 2000, 2300, 3000, 4000, 5000, 6000, 8192, 8200):
            p = db.debug_plan(lq)
            cover = p["K"] * p["G"]
            if p["passes"] == 1 or p["classes"] == 2:
                assert p["last_pass_cols"] == 0, (lq, p)
                continue
            rest = lq - (p["passes"] - 1) * cover
            assert 0 < rest <= cover, (lq, p)
            need = max(2, -(-rest // p["G"]))
            assert p["last_pass_cols"] == (need if need < p["K"] else 0), (lq, p, need)
    finally:
        db.close()


def test_both_forms_cut(swg):
    """The length from which a sequence takes the int16 cells in a search that runs both forms: 4096 * lq / qbound rows
    (qbound: the query's best possible total), and where it cuts the sorted order -- pairs (2i, 2i+1) by rank, a pair
    with one long member being a long pair."""
    sc = swg.load_scoring("BLOSUM62")
    tab = sc.table().astype(np.int64)
    lq = 8192
    q = swg.synth_query(0x5EED0005, lq)
    flat, off, planted = swg.synth_db(0x5EED0005, 20000, query=q, fraction=0.01, subst=0.05)
    qbound = int(tab[q.astype(np.int64)][:, 1:].max(axis=1).sum())
    db = swg.Database(flat, off)
    try:
        cut = db.debug_split(lq, qbound)
    finally:
        db.close()
    rows = -(-4096 * lq // qbound)
    assert cut["rows"] == rows and 700 < rows < 900, (cut, qbound)
    lens = np.sort(np.diff(off.astype(np.int64)))[::-1]
    n_long = int((lens >= rows).sum())
    first = (n_long + 1) // 2
    assert planted <= n_long < len(lens) // 2 and cut["first_pair"] == first and cut["residues_f16"] == int(lens[2 * first:].sum()), cut


@pytest.mark.parametrize("lq,want", [(600, (1, 19, 32, 4, 1)), (800, (1, 25, 32, 12, 1)), (1000, (1, 32, 32, 12, 1)),
                                     (1200, (1, 19, 64, 12, 1)), (1500, (1, 24, 64, 12, 1)), (2000, (1, 32, 64, 12, 1))])
def test_single_pass_lengths_between_the_configs(swg, lq, want):
    """What the autotuner picked on the device when the model was last compared with it:
    no long class for a tie, 12 wavefronts per workgroup where LDS keeps four from the same occupancy."""
    p = _plan(swg, 0x5EED0003, 200000, lq)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"]) == want, p


@pytest.mark.parametrize("lq,want", [(2500, (1, 27, 32, 12, 3)), (4000, (1, 32, 64, 12, 2)), (6000, (1, 32, 64, 12, 3))])
def test_several_passes_with_wide_groups(swg, lq, want):
    """Few passes of wide groups (12 wavefronts per workgroup) beat many of narrow ones when no pair is long
    against a pass: the autotuner's picks on the device, 3 to 4 % above what the model chose before."""
    p = _plan(swg, 0x5EED0003, 200000, lq)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"]) == want, p


def test_tiny_database_takes_the_widest_groups(swg):
    p = _plan(swg, 0x5EED0001, 1024, 128)
    assert p["G"] == 64 and p["passes"] == 1 and p["K"] * 64 >= 128, p


def test_few_pairs_do_not_share_their_simd(swg):
    """A database of few pairs does not fill the wave slots its geometry allows: the wavefront of the longest chain has
    its SIMD to itself whatever the workgroup's LDS size, so the fewest columns per lane win (round 3: 64 lanes x 17
    columns used to rank first for a 400-column query here -- one wavefront per SIMD by its LDS size -- and ran at half
    the speed of 64 x 7 on the device)."""
    lens = [300000, 120000, 7] + [int(v) for v in np.random.default_rng(3).integers(1, 600, size=400)]
    seqs = [swg.synth_query(1000 + i, L) for i, L in enumerate(lens)]
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    db = swg.Database(np.concatenate(seqs), off)
    try:
        p = db.debug_plan(400)
        assert (p["classes"], p["K"], p["G"], p["passes"]) == (1, 7, 64, 1), p
        p = db.debug_plan(128)
        assert (p["K"], p["G"], p["passes"]) == (2, 64, 1), p
    finally:
        db.close()
    flat, off = swg.synth_db(0x5EED0001, 1024)            # config 1's shape: the estimate is the longest pair's chain
    db = swg.Database(flat, off)
    try:
        p = db.debug_plan(128)
        assert (p["classes"], p["K"], p["G"], p["passes"]) == (1, 2, 64, 1) and 250 < p["est_us"] < 700, p
    finally:
        db.close()


def test_engine_choice_short_sequences_go_systolic(swg):
    """Round 4: the cost model compares both engines (swg_debug_engine).  Peptides -- short sequences of near-equal
    length -- go to the systolic engine for short queries (measured on 2 M peptides: lq 128 7 220 GCUPS against the lane
    groups' 4 875, lq 30 5 030 against 1 930); a query that needs more columns than one systolic pass holds, BASELINE's
    length distribution (a 5 000-row bin is a serial chain there) and small databases stay on the lane groups.  The
    estimates themselves are pinned to the measured fills within 15 %."""
    flat, off = swg.synth_db(0xBEEF, 2000000, median=29.0, sigma_ln=0.25, min_len=20, max_len=40)
    db = swg.Database(flat, off)
    e30, e128, e367, e600 = (db.debug_engine(lq) for lq in (30, 128, 367, 600))
    db.close()
    assert e30["systolic"] and e128["systolic"] and not e367["systolic"] and not e600["systolic"]
    assert e600["systolic_K"] == 0                                  # 600 columns: no single systolic pass
    assert abs(e128["diag_us"] - 1562) < 0.15 * 1562 and abs(e128["systolic_us"] - 1047) < 0.15 * 1047   # measured fills, us (systolic: on its f16 cells)
    assert abs(e30["diag_us"] - 920) < 0.15 * 920 and e30["systolic_us"] < 400
    for seed, n, lqs in ((0x5EED0002, 100000, (30, 128, 367)), (0x5EED0003, 570000, (128, 367, 500)), (0x5EED0001, 1024, (128,))):
        flat, off = swg.synth_db(seed, n)
        db = swg.Database(flat, off)
        for lq in lqs:
            assert not db.debug_engine(lq)["systolic"], (n, lq)
        db.close()


def test_list_rerun_takes_a_full_size_workgroup(swg):
    """The int16 re-run of what the f16 cells flagged: 64 lanes per pair from 512 columns up, and -- one workgroup per
    CU is all that fits beside such a profile -- a workgroup of as many wavefronts as the kernel was compiled for (the
    kernel trims it to the list's length, which only the device knows).  Until the end of round 4 it had four: one
    wavefront per SIMD whatever the list (config 4's share with relatives: 15.4 -> 10.7 ms)."""
    assert swg.debug_list_plan(3000, 1) == {"K": 24, "G": 64, "W": 16, "passes": 2}
    assert swg.debug_list_plan(3000, 3100) == {"K": 24, "G": 64, "W": 16, "passes": 2}       # (no guess involved)
    p = swg.debug_list_plan(8192, 1)
    assert (p["G"], p["passes"]) == (64, 4) and p["K"] == 32 and p["W"] == 12, p               # K = 32 is compiled for 12
    p = swg.debug_list_plan(600, 1)
    assert (p["K"], p["G"], p["W"], p["passes"]) == (10, 64, 16, 1), p
    # a short query's long list keeps the main fill's geometry
    assert swg.debug_list_plan(367, 5000, main=(23, 16, 4)) == {"K": 23, "G": 16, "W": 4, "passes": 1}
    assert swg.debug_list_plan(367, 100, main=(23, 16, 4))["G"] == 64
