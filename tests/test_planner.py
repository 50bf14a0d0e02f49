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


def test_config5_long_query_near_copies(swg):
    # the planted 8 200-row pairs are the chain of every pass: 64 lanes each, two wavefronts per SIMD
    p = _plan(swg, 0x5EED0005, 100000, 8192, fraction=0.01, subst=0.05)
    assert (p["classes"], p["K"], p["G"], p["W"], p["passes"]) == (1, 32, 64, 8, 4), p


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
