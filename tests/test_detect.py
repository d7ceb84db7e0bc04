"""Spectral detection: oracle = the reference's own library calls
(oracle/detect_ref.py); the HIP kernels through rmx_detect_batch against it (gpu).

The two FFTs (pocketfft / ours) agree to ~1e-6 relative, so dB values agree to ~1e-5 dB; a peak set
can only differ where two competing dB values are closer than that (a local-maximum test, a
highest-first removal between neighbours, the 0.3 confidence cut).  The tests require identical peak
sets on windows whose smallest such margin is above 1e-3 dB and report the rest."""
import numpy as np
import pytest

from oracle import detect_ref as dr


def make_windows(W, N, seed, tones=3, u8=False):
    rng = np.random.default_rng(seed)
    n = np.arange(N)
    x = (rng.standard_normal((W, N)) + 1j * rng.standard_normal((W, N))) * 6.0
    for w in range(W):
        for _ in range(tones):
            f = rng.uniform(-0.45, 0.45)
            x[w] += rng.uniform(10, 60) * np.exp(2j * np.pi * (f * n + rng.uniform()))
    if u8:
        re = np.clip(np.floor(x.real + 128.0), 0, 255).astype(np.uint8)
        im = np.clip(np.floor(x.imag + 128.0), 0, 255).astype(np.uint8)
        raw = np.empty((W, 2 * N), np.uint8)
        raw[:, 0::2], raw[:, 1::2] = re, im
        x = (re.astype(np.float32) - 127.5) + 1j * (im.astype(np.float32) - 127.5)
        return x.astype(np.complex64), raw
    return x.astype(np.complex64), None


def test_oracle_is_the_reference_call_sequence():
    """detect_one == the literal statements of buoy_node.py:401-433 on one window."""
    import scipy.signal
    from scipy.fft import fft
    x, _ = make_windows(1, 4096, 1)
    iq = x[0]
    p = 20 * np.log10(np.abs(fft(iq)) + 1e-12)
    peaks, _ = scipy.signal.find_peaks(p, height=-70, distance=10)
    floor = np.median(p)
    fs, fc = 2.4e6, 100e6
    freqs = np.fft.fftfreq(len(iq), 1.0 / fs) + fc
    want = []
    for k in peaks:
        if abs(freqs[k] - fc) < 10000:
            continue
        conf = min(max((p[k] - floor) / 20.0, 0.0), 1.0)
        if conf < 0.3:
            continue
        want.append(k)
    bins, pw, snr, conf, fl = dr.detect_one(iq, dc_exclude_bins=10e3 * len(iq) / fs)
    assert list(bins) == want and fl == floor
    assert np.all(conf >= 0.3) and np.all(conf <= 1.0) and np.allclose(snr, pw - fl)


def test_oracle_edge_cases():
    z = np.zeros(1024, np.complex64)
    bins, pw, snr, conf, fl = dr.detect_one(z)                 # flat spectrum: no local maximum
    assert len(bins) == 0 and abs(fl + 240.0) < 1e-3
    t = np.exp(2j * np.pi * 64 * np.arange(1024) / 1024).astype(np.complex64) * 100
    bins, pw, snr, conf, fl = dr.detect_one(t)                 # one on-bin tone
    assert 64 in bins and conf[list(bins).index(64)] == 1.0


def _margin(p, floor, dist, min_conf):
    """smallest dB gap that decides a peak set: neighbours of local maxima, competitors within dist,
    distance of the confidence to the cut"""
    import scipy.signal
    cand = scipy.signal.argrelextrema(p, np.greater_equal)[0]
    m = np.inf
    d1 = np.abs(np.diff(p))
    m = min(m, d1.min())
    for i, k in enumerate(cand):
        near = cand[(np.abs(cand - k) < dist) & (cand != k)]
        if len(near):
            m = min(m, np.abs(p[near] - p[k]).min())
    conf = (p[cand] - floor) / 20.0
    m = min(m, (np.abs(conf - min_conf) * 20.0).min())
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("N,W,u8", [(16384, 6, False), (8192, 6, False), (1024, 12, False), (16384, 4, True), (256, 8, False)])
def test_gpu_detect_matches_oracle(N, W, u8):
    import __graft_entry__ as g
    g.build()
    from radio_mapper_amd import xcorr
    x, raw = make_windows(W, N, seed=100 + N + (7 if u8 else 0), u8=u8)
    dc = 10e3 * N / 2.4e6
    ref = dr.detect_batch(x, dc_exclude_bins=dc)
    with xcorr.XcorrEngine(2, 4096, 1) as eng:
        got = eng.detect(raw if u8 else x, dc_exclude_bins=dc, max_peaks=N // 4)
    n_exact = 0
    for w in range(W):
        rb, rp, rs, rc, rf = ref[w]
        gb, gp, gs, gc, gf = got[w]
        assert abs(gf - rf) < 2e-4, (gf, rf)
        if np.array_equal(gb, rb):
            n_exact += 1
            assert np.abs(gp - rp).max() < 2e-4 and np.abs(gs - rs).max() < 4e-4 and np.abs(gc - rc).max() < 2e-5
        else:
            # differing sets must be explained by a near-tie somewhere in the window
            p = dr.power_spectrum_db(x[w])
            assert _margin(p, rf, 10, 0.3) < 1e-3, f"window {w}: sets differ without a near-tie"
            common = np.intersect1d(gb, rb)
            assert len(common) >= 0.98 * max(len(rb), 1)
    assert n_exact >= W - 2


@pytest.mark.gpu
def test_gpu_detect_edge_cases():
    import __graft_entry__ as g
    g.build()
    from radio_mapper_amd import xcorr
    N = 1024
    z = np.zeros((3, N), np.complex64)
    z[1] = 100 * np.exp(2j * np.pi * 64 * np.arange(N) / N)           # on-bin tone: one peak, confidence 1
    z[2] = 50 * np.exp(2j * np.pi * (-200) * np.arange(N) / N) + 50 * np.exp(2j * np.pi * (-195) * np.arange(N) / N)
    with xcorr.XcorrEngine(2, 4096, 1) as eng:
        got = eng.detect(z, max_peaks=64)
        ref = dr.detect_batch(z)
        assert len(got[0][0]) == 0 and abs(got[0][4] + 240.0) < 1e-3        # flat spectrum: nothing
        # pure tones: everything else is rounding noise of the FFT at hand; compare the tones
        assert np.array_equal(got[1][0][got[1][1] > 20.0], ref[1][0][ref[1][1] > 20.0])
        assert 64 in got[1][0] and got[1][3][list(got[1][0]).index(64)] == 1.0
        # two equal tones 5 bins apart (< distance): exactly one survives; which one is decided by the
        # last bit of the two magnitudes, i.e. by the FFT's rounding
        strong = got[2][0][got[2][1] > 20.0]
        assert len(strong) == 1 and strong[0] in (N - 200, N - 195)
        # capacity: count is reported in full, arrays are cut at max_peaks
        x, _ = make_windows(1, 4096, 5)
        full = eng.detect(x, max_peaks=4096)[0]
        cut = eng.detect(x, max_peaks=8)[0]
        assert len(cut[0]) == 8 and np.array_equal(cut[0], full[0][:8])
        with pytest.raises(xcorr.RmxError):
            eng.detect(np.zeros((1, 1000), np.complex64))                   # not a power of two
