import os

import numpy as np
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def near_tie_windows(N, seed):
    """Windows whose correlation has two isolated peaks of ALMOST equal height: x_i = one impulse, x_j = two impulses with
    random phases and an amplitude ratio 1 + k * 2^-23 (k = -4 ... 4), the HIGHER 'full' index carrying the larger one for
    k > 0.  |c| differs by about k ulp between the two lags, |c|^2 by about 2k ulp."""
    rng = np.random.default_rng(seed)
    ks = np.arange(-4, 5)
    e = np.zeros((len(ks), 2, N), np.complex64)
    for w, k in enumerate(ks):
        a, b = int(rng.integers(3, N // 4)), int(rng.integers(N // 2, N - 3))
        ph = np.exp(2j * np.pi * rng.random(3))
        amp = float(rng.uniform(20.0, 90.0))
        e[w, 0, 7] = amp * ph[0]
        e[w, 1, a] = np.complex64(amp * ph[1])
        e[w, 1, b] = np.complex64(amp * ph[2]) * np.float32(1.0 + k * 2.0 ** -23)
    return e
