import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def tune_env(**tunables):
    """Environment of a test subprocess whose library gets the given tunables (srs_tuning_set via the mirror's SRS_TEST_TUNING hook) plus
    any UPPER-CASE extra variables, e.g. tune_env(msm_sort=2, commit_chunks=3, N="100")."""
    extra = {k: str(v) for k, v in tunables.items() if k.isupper()}
    spec = ",".join(f"{k}={int(v)}" for k, v in tunables.items() if not k.isupper())
    return dict(os.environ, SRS_TEST_TUNING=spec, **extra)


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def h2i(x):
    return int(x, 16)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def srs():
    """The product package, loaded on the real HIP library (fails loudly when it is missing)."""
    import sirius_amd as S
    from sirius_amd import _lib
    _lib.load()
    return S


def seeded_scalars(O, curve, n, seed, kind="uniform"):
    """(n,4) Montgomery scalars of `curve`'s scalar field; value distributions of SURVEY.md 8d."""
    from oracle import pyref as P
    q = P.CURVES[curve].q
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)          # < 2^253 < q: canonical without rejection
    if kind == "trace":
        u = rng.random(n)
        raw[u < 0.55] = 0                                       # 55 % zero
        m = (u >= 0.55) & (u < 0.75)
        raw[m] = 0
        raw[m, 0] = rng.integers(0, 2, size=int(m.sum()), dtype=np.uint64)   # 20 % bits
        m = (u >= 0.75) & (u < 0.90)
        raw[m, 1:] = 0                                          # 15 % < 2^64
    sf = O.SCALAR_FIELD[curve]
    return O.to_mont(sf, raw)
