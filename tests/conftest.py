import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import swg_loader  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    # incremental: a no-op when libswg.so, the CLI and the oracle are newer than their sources
    swg_loader.build_module().build(verbose=False)
    orc = swg_loader.oracle()
    if not os.path.exists(orc.ORACLE_SO):
        orc.build()


@pytest.fixture(scope="session")
def swg():
    _ensure_built()
    return swg_loader.load()


@pytest.fixture(scope="session")
def orc():
    _ensure_built()
    return swg_loader.oracle()


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def ctx(swg):
    """One GPU context for the whole session (GPU tests only)."""
    c = swg.Context(0)
    yield c
    c.close()
