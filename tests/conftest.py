import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_tables():
    from hifihr_amd.mano_tables import synthetic_mano_tables
    return synthetic_mano_tables(0)


def mano_pkl_path():
    """A user-supplied MANO file (never shipped).  In the build container the reference mounts one."""
    for p in (os.environ.get("HIFIHR_MANO_PKL"), "/root/reference/data/MANO_RIGHT.pkl"):
        if p and os.path.exists(p):
            return p
    return None
