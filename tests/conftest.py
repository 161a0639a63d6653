import os
import sys

import numpy as np
import pytest

# OpenMP threads that find nothing to do sleep instead of spinning (must be set before the first OpenMP runtime of the process starts).  The C
# oracle's loops over 25 million pairs share the box with whatever the subprocess-based tests before them have left running; spinning teams on
# oversubscribed cores are the one known way for a 9-s test to take minutes (r05: one such run, never located: docs/HISTORY.md 0).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sample():
    """The reference's bundled snp_sample alignment (states, POS) + oracle outputs committed as golden data."""
    g = np.load(os.path.join(GOLDEN, "snp_sample_states.npz"))
    o = np.load(os.path.join(GOLDEN, "snp_sample_oracle.npz"))
    d = {k: o[k] for k in o.files}
    d["states"], d["POS"] = g["states"], g["POS"]
    d["g"] = float(d["g"])
    return d


@pytest.fixture(scope="session")
def synth():
    s = np.load(os.path.join(GOLDEN, "synth_c2slice.npz"))
    d = {k: s[k] for k in s.files}
    d["g"] = float(d["g"])
    return d


@pytest.fixture(scope="session")
def kat():
    s = np.load(os.path.join(GOLDEN, "kat_small.npz"))
    return {k: s[k] for k in s.files}


@pytest.fixture(scope="session")
def engine():
    from ldweaver_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()
