"""The two alternative builds of the fused N = 4096 kernel (k_win8, k_winp): parity of a -DRMX_EXPERIMENTS library.
Not part of the product's suite (tests/): the default library does not contain these kernels.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
          -DRMX_EXPERIMENTS -o /tmp/librmx_exp.so radio-mapper_amd/csrc/rmx_hip.hip
    RMX_LIBRARY=/tmp/librmx_exp.so python -m pytest tools/experiments -q -p no:cacheprovider
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import radio_mapper_amd as rm                    # noqa: E402
from oracle import xcorr_ref as orc              # noqa: E402
from test_gpu_parity import TOL, _assert_parity  # noqa: E402


@pytest.fixture(scope="module")
def xc():
    from radio_mapper_amd import xcorr
    if xcorr.device_count() == 0:
        pytest.skip("no GPU")
    return xcorr


@pytest.fixture(scope="module")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("opt", ["win8", "pk"])
def test_alternative_window_kernels(xc, golden_dir, opt):
    """The two alternative builds of the fused N = 4096 kernel -- k_win8 (8 points x 1024 threads, radix-8,
    4 waves per SIMD) and k_winp (k_win on packed fp32) -- against the reference-generated fixture, the
    exact-tie construction and the default kernel on a few hundred windows (complex64 and raw uint8)."""
    g = np.load(os.path.join(golden_dir, "xcorr_b8_n4096.npz"))
    iq = orc.decode_u8_iq(g["raw_u8"])
    W, B, N = iq.shape
    with xc.XcorrEngine(B, N, W) as eng:
        try:
            eng.set_option(opt, 1)
        except xc.RmxError as e:
            assert e.code == -5 and "RMX_EXPERIMENTS" in str(e)     # RMX_E_UNSUPPORTED: the default build has neither kernel
            pytest.skip("this library was built without -DRMX_EXPERIMENTS (set RMX_LIBRARY to the experiments build)")
        li, lf, pk = eng.correlate(iq)
        _assert_parity(li, lf, pk, g["lag_int"], g["lag_frac"], g["peak"], g["margin"])
        li8, lf8, pk8 = eng.correlate(g["raw_u8"])
        assert np.array_equal(li, li8) and np.array_equal(lf, lf8) and np.array_equal(pk, pk8)
    e = np.zeros((1, 2, N), np.complex64)
    a = 777
    e[0, 0, 0] = 1.0; e[0, 0, N // 2] = 1.0
    e[0, 1, a] = 1.0; e[0, 1, a + N // 2] = -1.0
    with xc.XcorrEngine(2, N, 1) as eng:
        eng.set_option(opt, 1)
        li, lf, pk = eng.correlate(e)
    assert li[0, 0] == a - N // 2 and abs(pk[0, 0] - 1.0) < 1e-5      # exact tie -> lowest 'full' index
    for nb, nw in ((3, 5), (8, 300), (16, 3)):
        x, _ = rm.synth.make_windows(nw, nb, N, 10e6, seed=77 + nb)
        with xc.XcorrEngine(nb, N, nw) as eng:
            l0, f0, p0 = eng.correlate(x)
            eng.set_option(opt, 1)
            l1, f1, p1 = eng.correlate(x)
        assert np.array_equal(l0, l1)
        assert np.allclose(l0 + f0, l1 + f1, atol=TOL) and np.allclose(p0, p1, rtol=1e-5)
