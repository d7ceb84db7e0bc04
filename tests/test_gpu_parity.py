"""Parity of the HIP path (through the C ABI) against the oracle and the golden fixtures.
Bars: integer lag bit-exact; |lag - lag_ref| <= 1e-5 * max(|lag_ref|, 1); peak rtol 1e-5.

Where the oracle's own two largest magnitudes are closer than 1e-5 relative the integer argmax is decided by the
float32 rounding of the FFT used (pocketfft vs ours): such a pair-window is accepted only if the GPU's lag IS the
oracle's second candidate (`_assert_parity(..., margin, second)`; the seeded sweeps compute both with
`oracle.peak_top2`).  The committed fixtures get no such exception: their smallest margin is 2.3e-3."""
import os

import numpy as np
import pytest

import radio_mapper_amd as rm
from conftest import near_tie_windows
from oracle import xcorr_ref as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture
def opts(xc):
    """kernel-selection defaults for the engines a test creates (rmx_set_default_option), cleared afterwards"""
    yield xc.set_default_option
    xc.clear_default_options()


@pytest.fixture(scope="module")
def xc():
    import __graft_entry__ as g
    g.build()
    from radio_mapper_amd import xcorr
    assert xcorr.device_count() > 0, "no MI355X visible"
    return xcorr


def _top2(iq, plist):
    """(margin [W][P], lag of the oracle's second-largest magnitude sample [W][P]) for _assert_parity"""
    W = iq.shape[0]
    m = np.zeros((W, len(plist)))
    second = np.zeros((W, len(plist)), np.int64)
    for w in range(W):
        for q, (i, j) in enumerate(plist):
            m[w, q], _, second[w, q] = orc.peak_top2(iq[w, i], iq[w, j])
    return m, second


def _assert_parity(li, lf, pk, ri, rf, rp, margin=None, second=None):
    """Integer lags bit-exact.  The one accepted exception: the oracle's own two largest magnitudes are within 1e-5
    of each other (`margin`) AND the GPU's lag is the oracle's second candidate (`second`) -- then two correct float32
    FFTs may order the pair differently.  Without `second` (the committed fixtures, whose smallest margin is 2.3e-3)
    there is no exception at all."""
    ref = ri + rf
    got = li + lf.astype(np.float64)
    bad = li != ri
    if margin is not None and second is not None:
        excused = bad & (margin <= TOL) & (li == second)
        assert not np.any(bad & ~excused), f"{int(np.sum(bad & ~excused))} integer lags differ"
    else:
        assert not bad.any(), f"{int(bad.sum())} integer lags differ"
    ok = ~bad
    assert np.all(np.abs(got[ok] - ref[ok]) <= TOL * np.maximum(np.abs(ref[ok]), 1.0)), \
        "fractional lag outside 1e-5: max %.3e" % np.abs(got[ok] - ref[ok]).max()
    assert np.allclose(pk[ok], rp[ok], rtol=1e-5, atol=0)


@pytest.mark.parametrize("name", ["xcorr_b3_n4096", "xcorr_b8_n4096"])
def test_golden_fixtures(xc, golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    iq = orc.decode_u8_iq(g["raw_u8"])
    W, B, N = iq.shape
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
        _assert_parity(li, lf, pk, g["lag_int"], g["lag_frac"], g["peak"], g["margin"])
        # raw uint8 ingest (decode fused in the first kernel) must give identical results
        li8, lf8, pk8 = eng.correlate(g["raw_u8"])
        assert np.array_equal(li, li8) and np.array_equal(lf, lf8) and np.array_equal(pk, pk8)


@pytest.mark.parametrize("name", ["xcorr_b4_n256", "xcorr_b3_n1024", "xcorr_b3_n16384", "xcorr_b3_n8192"])
def test_golden_fixtures_other_lengths(xc, golden_dir, name, opts):
    """Window lengths other than 4096 run the generic path: same definition, same bar, against fixtures generated from
    the reference module (xcorr_b3_n8192 and xcorr_b3_n16384 are the reference's own capture lengths,
    iq_stream_client.py:459 and buoy_node.py:364).  Twice: with the dispatch a batch of this size gets (few windows: the
    per-transform / four-step kernels), and with the whole-window kernels forced (option wscr = 2: g_win_scr,
    g_win_scr14, g_win_eo15 -- what a batch that fills the chip runs); no exception for these fixtures in either."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    iq = orc.decode_u8_iq(g["raw_u8"])
    W, B, N = iq.shape
    for force in (False, True) + (("k16",) if N == 16384 else ()):
        if force == "k16":            # N = 16384 from about a hundred transforms per batch on: k16_fwd + k16_pairs (kwin16k.hpp)
            xc.clear_default_options()
            opts("kwin16k", 2)
        elif force:
            opts("wscr", 2)
        with xc.XcorrEngine(B, N, W) as eng:
            li, lf, pk = eng.correlate(iq)
            _assert_parity(li, lf, pk, g["lag_int"], g["lag_frac"], g["peak"], g["margin"])
            li8, lf8, pk8 = eng.correlate(g["raw_u8"])
            assert np.array_equal(li, li8) and np.array_equal(lf, lf8) and np.array_equal(pk, pk8)


def test_edge_cases_n256_fixture(xc, golden_dir):
    g = np.load(os.path.join(golden_dir, "xcorr_edge_n256.npz"))
    iq = g["iq"]
    n = iq.shape[-1]
    with xc.XcorrEngine(2, n, iq.shape[0]) as eng:
        li, lf, pk = eng.correlate(iq)
    assert li[0, 0] == -(n - 1) and lf[0, 0] == 0.0 and pk[0, 0] == 0.0
    assert li[1, 0] == 20 and li[2, 0] == n - 1 and li[3, 0] == -(n - 1)
    assert lf[2, 0] == 0.0 and lf[3, 0] == 0.0
    # window 4: two impulses of equal height -> |r| has two maxima of 1.0 at lags -7 and +9 that are equal only up to
    # the last bit of whichever FFT computed them (the fixture, i.e. pocketfft, records -7).  Which of two such
    # values is larger is not part of the path's definition; the tie RULE (lowest 'full' index) is, and it is
    # enforced where the kernel reproduces a tie exactly: test_exact_tie_resolves_to_lowest_index.
    # Written through the oracle like every seeded test (VERDICT r04 #7a): the oracle's own two largest magnitudes must be
    # within 1e-5 of each other, and the GPU's lag must be the oracle's first or its second candidate -- nothing else.
    margin4, first4, second4 = orc.peak_top2(iq[4, 0], iq[4, 1])
    assert int(g["lag_int"][4, 0]) == first4 == -7 and second4 == 9 and margin4 <= TOL
    assert li[4, 0] == first4 or (margin4 <= TOL and li[4, 0] == second4)
    assert li[5, 0] == 0
    assert np.allclose(pk[1:], g["peak"][1:], rtol=1e-5)
    # fractional lag of every window against the reference-generated values (windows 0, 2, 3: exactly 0
    # by the edge / flat-top rule; 1, 4: isolated impulses, the neighbour taps are round-off of a peak of
    # height 1..2 -> |frac| ~ 1e-8; 5: the flat triangular top, conditioned as in test_edge_cases_n4096)
    ref = g["lag_frac"]
    assert lf[0, 0] == ref[0, 0] == 0.0
    for w in (1, 2, 3, 4):
        assert abs(lf[w, 0] - ref[w, 0]) <= TOL, (w, lf[w, 0], ref[w, 0])
    cond = n / 2.0
    assert abs(lf[5, 0] - ref[5, 0]) <= max(TOL, 4 * 6e-8 * cond)


@pytest.mark.parametrize("name", ["xcorr_cfg1_n262144", "xcorr_b3_n1048576"])
def test_large_windows_seeded(xc, golden_dir, name):
    """BASELINE configs[0]/[1] shapes (N = 2^18, 2^20): inputs regenerated from the recorded seed and
    checked against the recorded checksum, outputs against the reference-generated values."""
    import hashlib
    import json
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    kw = json.loads(str(g["gen"]))
    iq, _ = rm.synth.make_windows(**kw)
    assert hashlib.sha256(np.ascontiguousarray(iq).tobytes()).hexdigest() == str(g["input_sha256"])
    W, B, N = iq.shape
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
    _assert_parity(li, lf, pk, g["lag_int"], g["lag_frac"], g["peak"], g["margin"])


def test_edge_cases_n4096(xc):
    """Zeros (all-tie), impulses, peaks on both edges, an exact two-peak tie, constant inputs."""
    N = 4096
    e = np.zeros((6, 2, N), np.complex64)
    e[1, 0, 5] = 1.0; e[1, 1, 25] = 2.0 - 1.0j
    e[2, 0, 0] = 3.0; e[2, 1, N - 1] = 1.0j
    e[3, 0, N - 1] = 1.0; e[3, 1, 0] = -2.0
    e[4, 0, 100] = 1.0; e[4, 1, 93] = 1.0; e[4, 1, 109] = 1.0
    e[5, 0, :] = 4.5 - 2.5j; e[5, 1, :] = -1.5 + 0.5j
    ri, rf, rp = orc.xcorr_batch_literal(e)
    with xc.XcorrEngine(2, N, 6) as eng:
        li, lf, pk = eng.correlate(e)
    assert li[0, 0] == -(N - 1) and lf[0, 0] == 0.0 and pk[0, 0] == 0.0      # all ties -> lowest index
    assert li[1, 0] == ri[1, 0] == 20
    assert li[2, 0] == ri[2, 0] == N - 1 and lf[2, 0] == 0.0
    assert li[3, 0] == ri[3, 0] == -(N - 1) and lf[3, 0] == 0.0
    # a tie only up to each FFT's last bit (see test_edge_cases_n256_fixture): the oracle's first or second candidate
    margin4, first4, second4 = orc.peak_top2(e[4, 0], e[4, 1])
    assert ri[4, 0] == first4 and {first4, second4} == {-7, 9} and margin4 <= TOL
    assert li[4, 0] == first4 or li[4, 0] == second4
    # constant inputs -> triangular |r|: the parabola's curvature is 2/N of its height, so one float32
    # ulp of tap asymmetry moves the vertex by eps32 * N/4 = 1.2e-4 samples.  The bar for such a
    # flat top is the 1e-5 of the spec OR 4 ulp of tap error through that conditioning.
    cond = float(rp[5, 0]) / abs(2.0 * float(rp[5, 0]) / N)
    assert li[5, 0] == ri[5, 0] == 0 and abs(lf[5, 0] - rf[5, 0]) <= max(TOL, 4 * 6e-8 * cond)
    assert np.allclose(pk[1:], rp[1:], rtol=1e-5)


def test_exact_tie_resolves_to_lowest_index(xc):
    """numpy's argmax rule (ties -> lowest 'full' index, i.e. the most negative lag) on a tie that the
    HIP transform reproduces EXACTLY, so the rule is tested rather than the rounding.

    x_i = d[0] + d[N/2],  x_j = d[a] - d[a + N/2]  (d = unit impulse, a < N/2).  The true correlation
    has its two peaks at lags a - N/2 (+1) and a + N/2 (-1), N apart, and cancels at lag a.  In the
    kernel every even bin of the 2N-point product is an exact zero (the N-point spectrum of x_i is 2 on
    even and exactly 0 on odd bins, that of x_j exactly 0 on even bins: the butterflies only add and
    subtract equal values there, twiddles multiply zeros), so the even half-transform e[n] is exactly 0
    and the last radix-2 gives r[n] = +o[n], r[n + N] = -o[n]: the two peaks have bit-identical
    magnitudes.  The lower 'full' index, lag a - N/2, must win, with the same peak value either way;
    the all-pairs kernel and the custom-pair kernels (same FFT blocks) agree."""
    N = 4096
    for a in (5, 777, 2047):
        e = np.zeros((1, 2, N), np.complex64)
        e[0, 0, 0] = 1.0; e[0, 0, N // 2] = 1.0
        e[0, 1, a] = 1.0; e[0, 1, a + N // 2] = -1.0
        ri, rf, rp = orc.xcorr_batch_literal(e)
        assert ri[0, 0] in (a - N // 2, a + N // 2)              # the oracle's FFT decides its own tie by rounding
        with xc.XcorrEngine(2, N, 1) as eng:
            li, lf, pk = eng.correlate(e)
            lc, fc, pc = eng.correlate(e, pairs=np.array([[0, 1], [1, 0]], np.int32))
        assert li[0, 0] == a - N // 2, (a, li)
        assert lc[0, 0] == a - N // 2
        # reversed pair: peaks at -(a - N/2) and -(a + N/2); the lower index is -(a + N/2)
        assert lc[0, 1] == -(a + N // 2)
        assert abs(pk[0, 0] - 1.0) < 1e-5 and abs(pc[0, 0] - 1.0) < 1e-5 and abs(pc[0, 1] - 1.0) < 1e-5
        assert abs(lf[0, 0]) <= TOL


def test_exact_tie_in_uint8_windows_of_16_samples(xc):
    """Case 405 of the round-5 soak (seed 1001): 7 windows of 16 buoys x 16 raw uint8 samples at 3 dB.  In window 3 the lags 4
    and 10 of pair (9, 13) tie EXACTLY (half-integer samples: every product sum is exact in float32): `scipy.signal.correlate`
    -- the reference's primitive, oracle.xcorr_batch_literal -- and the engine both report the lower index, while the oracle's
    fast form (one FFT per buoy, float32) rounds the tie the other way.  The engine must equal the literal oracle on every
    pair-window of the batch, from raw bytes and from the decoded complex64."""
    out = rm.synth.make_windows(7, 16, 16, 20e6, seed=539948125, snr_db=3.0, return_u8=True)
    iq, raw = out[0], out[2]
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    pl = orc.pair_list(16)
    q = [k for k in range(len(pl)) if tuple(pl[k]) == (9, 13)][0]
    margin, first, second = orc.peak_top2(iq[3, 9], iq[3, 13])
    assert margin == 0.0 and {first, second} == {4, 10} and ri[3, q] == 4
    with xc.XcorrEngine(16, 16, 7) as eng:
        li, lf, pk = eng.correlate(raw)
        lc, fc, pc = eng.correlate(iq)
    assert np.array_equal(li, ri) and np.array_equal(lc, ri)
    assert np.array_equal(lf, fc) and np.array_equal(pk, pc)
    _assert_parity(li, lf, pk, ri, rf, rp)


def test_custom_pairs_and_antisymmetry(xc):
    iq, _ = rm.synth.make_windows(16, 5, 4096, 10e6, seed=21)
    fwd = np.array([(0, 1), (0, 4), (2, 3), (1, 4)], np.int32)
    rev = fwd[:, ::-1].copy()
    with xc.XcorrEngine(5, 4096, 16) as eng:
        li, lf, pk = eng.correlate(iq, fwd)
        ri, rf, rpk = eng.correlate(iq, rev)
        lall, fall, pall = eng.correlate(iq)
    # r_ji[tau] = conj(r_ij[-tau]): mirrored lag, same peak (different rounding order: 1e-5)
    assert np.array_equal(li, -ri)
    assert np.all(np.abs(lf + rf) <= TOL) and np.allclose(pk, rpk, rtol=1e-5)
    allp = [tuple(p) for p in xc.pair_list(5)]
    cols = [allp.index(tuple(p)) for p in fwd]
    # a custom list runs the two-kernel path, the default list the fused window kernel: same lags,
    # last-bit differences in the fraction allowed
    assert np.array_equal(li, lall[:, cols]) and np.all(np.abs(lf - fall[:, cols]) <= TOL)
    assert np.allclose(pk, pall[:, cols], rtol=1e-5)
    # against the oracle with the same custom list
    oi, of_, op = orc.xcorr_batch_literal(iq, fwd)
    _assert_parity(li, lf, pk, oi, of_, op)


def test_cfg3_full_size_properties_and_subset_parity(xc):
    """BASELINE configs[2]: B=8, N=4096, W=4096.  Size-independent properties on the full batch
    (results do not depend on chunking / work split / window order), oracle parity on 256 windows."""
    import torch
    W, B, N = 4096, 8, 4096
    iq_small, delays = rm.synth.make_windows(256, B, N, 10e6, seed=1003)
    rng = np.random.default_rng(0)
    reps = W // 256
    # full batch = 16 differently rotated copies of the 256 seeded windows (rotation of every buoy by
    # the same amount keeps the lag structure but changes all samples)
    iq = np.empty((W, B, N), np.complex64)
    for r in range(reps):
        iq[r * 256:(r + 1) * 256] = iq_small if r == 0 else np.roll(iq_small, 97 * r, axis=2)
    P = B * (B - 1) // 2
    with xc.XcorrEngine(B, N, W) as eng:
        a = eng.correlate(iq)
        eng.set_option("chunk_windows", 136)
        eng.set_option("pairs_per_block", 5)
        b = eng.correlate(iq)
        perm = rng.permutation(W)
        c = eng.correlate(iq[perm])
        eng.set_option("resident", 0)                    # the 2-workgroups-per-CU kernel variant
        d = eng.correlate(iq[:512])
    for x, y in zip(a, b):
        assert np.array_equal(x, y)                      # work decomposition never changes a bit
    _assert_parity(d[0], d[1], d[2], a[0][:512], a[1][:512].astype(np.float64), a[2][:512])
    for x, y in zip(a, c):
        assert np.array_equal(x[perm], y)                # windows are independent
    ri, rf, rp = orc.xcorr_batch_fast(iq[:256], workers=16)
    _assert_parity(a[0][:256], a[1][:256], a[2][:256], ri, rf, rp)
    pairs = orc.pair_list(B)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    assert np.all(np.abs(a[0][:256] + a[1][:256] - true) < 0.5)   # and the lags are the simulated ones
    # closure on the integer-free quantity: lag(0,1) + lag(1,2) - lag(0,2) ~ 0 for a common source
    lag = a[0][:256] + a[1][:256].astype(np.float64)
    cols = {tuple(p): k for k, p in enumerate(pairs)}
    clos = lag[:, cols[(0, 1)]] + lag[:, cols[(1, 2)]] - lag[:, cols[(0, 2)]]
    assert np.all(np.abs(clos) < 1.0)


def test_device_pointer_entry_and_timing(xc):
    import torch
    W, B, N = 64, 8, 4096
    iq, _ = rm.synth.make_windows(W, B, N, 10e6, seed=5)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(iq.view(np.float32).reshape(W, B, N, 2)).to(dev)
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, P), dtype=torch.float32, device=dev)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_option("timing", 1)
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        torch.cuda.synchronize()
        tm = eng.last_timing()
        host = eng.correlate(iq)
    assert tm["pair_launches"] >= 1 and tm["pair_ms"] > 0
    assert np.array_equal(lag.cpu().numpy(), host[0])
    assert np.array_equal(frac.cpu().numpy(), host[1])
    assert np.array_equal(peak.cpu().numpy(), host[2])


def test_tdoa_processor_with_iq(xc):
    """Drop-in API: detections that carry IQ get time differences from the GPU lag (S7)."""
    from radio_mapper_amd import tdoa_processor as tp
    fs = 10e6
    iq, d = rm.synth.make_windows(1, 4, 4096, fs, seed=77)
    proc = tp.TDoAProcessor()
    names = ["N", "E", "S", "W"]
    coords = [(35.50, -97.50), (35.47, -97.45), (35.44, -97.50), (35.47, -97.55)]
    for n, (la, lo) in zip(names, coords):
        proc.register_buoy(tp.BuoyPosition(n, la, lo, 0.0, 100))
    t0 = 1_700_000_000_000_000_000
    dets = [tp.SignalDetection(n, 121.5, -50.0, "t", t0, la, lo, 0.9, "emergency", iq[0, k], fs)
            for k, (n, (la, lo)) in enumerate(zip(names, coords))]
    meas = proc.tdoa_calculator.calculate_tdoa_measurements(dets, proc.buoy_positions)
    oi, of_, _ = orc.xcorr_batch_literal(iq)
    assert len(meas) == 6
    for q, m in enumerate(meas):
        ns, metres = orc.lag_to_tdoa(oi[0, q] + of_[0, q], fs)
        assert abs(m.time_difference_ns - ns) <= 1
        assert abs(m.distance_difference_m - m.time_difference_ns / 1e9 * 299792458.0) < 1e-9


@pytest.mark.parametrize("name", ["caf_b3_n4096", "caf_b3_n1024", "caf_b4_n2048_d21"])
def test_caf_golden(xc, golden_dir, name):
    """rmx_caf_batch against the Doppler-grid fixtures generated with the reference primitive:
    winning hypothesis and integer lag exact, fractional lag / peak to 1e-5; per-bin peaks of the
    fixture give the margin between the best and the second-best hypothesis (>= 1e-2 here)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    iq = orc.decode_u8_iq(g["raw_u8"])
    W, B, N = iq.shape
    bp = np.sort(g["bin_peak"].astype(np.float64), axis=-1)
    assert ((bp[..., -1] - bp[..., -2]) / bp[..., -1]).min() > 1e-3
    with xc.XcorrEngine(B, N, W) as eng:
        dop, li, lf, pk = eng.caf(iq, g["doppler_cps"])
        assert np.array_equal(dop, g["dop_idx"])
        _assert_parity(li, lf, pk, g["lag_int"], g["lag_frac"], g["peak"])
        dop8, li8, lf8, pk8 = eng.caf(g["raw_u8"], g["doppler_cps"])
        assert np.array_equal(dop, dop8) and np.array_equal(li, li8) and np.array_equal(lf, lf8)
        # a single zero-Doppler hypothesis is the plain path
        d0, l0, f0, p0 = eng.caf(iq, [0.0])
        l1, f1, p1 = eng.correlate(iq)
        assert not d0.any() and np.array_equal(l0, l1)
        assert np.allclose(l0 + f0, l1 + f1, atol=TOL) and np.allclose(p0, p1, rtol=1e-5)
        # custom pair list
        sel = np.array([[1, 2], [0, 2]], np.int32)
        at = [[tuple(p) for p in g["pairs"].tolist()].index(tuple(q)) for q in sel.tolist()]
        ds, ls, fs_, ps = eng.caf(iq, g["doppler_cps"], sel)
        assert np.array_equal(ds, dop[:, at]) and np.array_equal(ls, li[:, at])


@pytest.mark.parametrize("n_buoys,n_windows", [(2, 5), (3, 4), (5, 3), (7, 3), (16, 2), (32, 1)])
def test_buoy_counts_n4096(xc, n_buoys, n_windows):
    """The fused window kernel's schedule depends on the buoy count (anchor runs, alternating stream
    direction, batches of 7 resolved peaks, cfg4's 16 buoys = 120 pairs); every count against the
    oracle on seeded windows, and more windows than one persistent workgroup pass would need."""
    fs = 10e6
    iq, delays = rm.synth.make_windows(n_windows, n_buoys, 4096, fs, seed=400 + n_buoys)
    ri, rf, rp = orc.xcorr_batch_fast(iq, workers=4)
    with xc.XcorrEngine(n_buoys, 4096, n_windows) as eng:
        li, lf, pk = eng.correlate(iq)
    assert li.shape == (n_windows, n_buoys * (n_buoys - 1) // 2)
    margin, second = _top2(iq[:n_windows], orc.pair_list(n_buoys))
    _assert_parity(li, lf, pk, ri, rf, rp, margin, second)
    true = delays[:, orc.pair_list(n_buoys)[:, 1]] - delays[:, orc.pair_list(n_buoys)[:, 0]]
    assert np.all(np.abs(li + lf - true) < 0.5)


def test_more_windows_than_compute_units(xc):
    """Persistent workgroups loop over windows: 600 windows > 256 CUs exercises the second and third
    pass of a workgroup (tables kept, record ring and exchange images reused across windows)."""
    iq, _ = rm.synth.make_windows(600, 3, 4096, 2.4e6, seed=77)
    ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
    with xc.XcorrEngine(3, 4096, 600) as eng:
        li, lf, pk = eng.correlate(iq)
        assert np.array_equal(li, ri)
        ref = ri + rf
        assert np.all(np.abs((li + lf.astype(np.float64)) - ref) <= TOL * np.maximum(np.abs(ref), 1.0))
        eng.set_option("chunk_windows", 100)          # several launches, last one partial
        li2, lf2, pk2 = eng.correlate(iq)
        assert np.array_equal(li, li2) and np.array_equal(lf, lf2) and np.array_equal(pk, pk2)


def test_caf_four_step_length(xc):
    """cfg5's shape in small: the Doppler grid on a window length that runs the four-step path
    (N = 16384), against the oracle; one buoy carries a 2-bin Doppler offset."""
    N, fs = 16384, 20e6
    step = 0.5 / N
    grid = (np.arange(5) - 2) * step
    offs = np.array([0.0, 2.0, -1.0]) * step
    iq, delays = rm.synth.make_windows(1, 3, N, fs, seed=55, doppler_cps=offs)
    rd, ri, rf, rp = orc.caf_batch(iq, grid)
    with xc.XcorrEngine(3, N, 1) as eng:
        dop, li, lf, pk = eng.caf(iq, grid)
    assert np.array_equal(dop, rd) and np.array_equal(dop[0], [4, 1, 0])
    _assert_parity(li, lf, pk, ri, rf, rp)


def test_error_codes_on_device(xc):
    """The C functions return a negative code and a message, never abort (the reference's 'log and
    return' convention, tdoa_processor.py:151-153, restated at the C boundary)."""
    import ctypes as C
    lib = xc.load_library()
    with xc.XcorrEngine(3, 4096, 4) as eng:
        iq = np.zeros((5, 3, 4096), np.complex64)
        with pytest.raises(xc.RmxError) as e:
            eng.correlate(iq)                                    # n_windows > max_windows
        assert e.value.code == -1 and "max_windows" in str(e.value)
        with pytest.raises(xc.RmxError):
            eng.correlate(iq[:2], pairs=np.array([[0, 3]], np.int32))   # buoy index out of range
        with pytest.raises(ValueError):
            eng.correlate(np.zeros((2, 3, 1000), np.complex64))  # wrong window length (binding check)
        assert lib.rmx_xcorr_batch(eng._ctx, None, 1, None, 3, None, None, None, 0) == -1
        with pytest.raises(xc.RmxError):
            eng.caf(iq[:2], [])                                  # empty Doppler grid
        with pytest.raises(xc.RmxError):
            eng.set_option("no_such_option", 1)
        with pytest.raises(xc.RmxError) as e:
            eng.set_option("dbg", 2)                             # ablation masks exist only under -DRMX_ABLATE
        assert e.value.code == -5
        with pytest.raises(ValueError):
            eng.caf(np.zeros((2, 4, 4096), np.complex64), [0.0])  # wrong buoy count: refused before C reads it
        with pytest.raises(xc.RmxError):
            eng.solve(np.zeros((3, 3)), np.zeros((2, 3), np.int32), np.zeros((2, 3), np.float32), 0.0)   # fs = 0
        # the engine is still usable after errors
        li, lf, pk = eng.correlate(iq[:2])
        assert li.shape == (2, 3) and np.all(li == -(4096 - 1)) and np.all(pk == 0.0)   # all-zero windows
    with pytest.raises(xc.RmxError) as e:
        xc.XcorrEngine(3, 4096, 4, device=99)
    assert e.value.code == -1


@pytest.mark.parametrize("shape", [(8, 4096, 300), (3, 4096, 5), (3, 1024, 40), (3, 65536, 6)])
def test_mixed_pointer_flags_through_the_raw_abi(xc, shape):
    """VERDICT r04 #7c: rmx_xcorr_batch accepts RMX_IN_DEVICE alone (device windows, host results) and RMX_OUT_DEVICE alone
    (host windows, device results); the binding only ever passes both or neither.  Raw ctypes calls of all four forms on
    one ctx -- fused N = 4096 with a partial round, the small-batch route, a whole-window kernel, the four-step path --
    must give identical arrays, equal to the oracle's."""
    import ctypes as C
    import torch
    B, N, W = shape
    iq = rm.synth.make_windows(W, B, N, 10e6, seed=sum(shape))[0]
    ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
    P = B * (B - 1) // 2
    lib = xc.load_library()
    dev = torch.device("cuda", 0)
    x_d = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32).reshape(W, B, N, 2)).to(dev)
    got = {}
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for flags in (0, xc.RMX_IN_DEVICE, xc.RMX_OUT_DEVICE, xc.RMX_IN_DEVICE | xc.RMX_OUT_DEVICE):
            in_ptr = C.c_void_p(x_d.data_ptr()) if flags & xc.RMX_IN_DEVICE else iq.ctypes.data_as(C.c_void_p)
            if flags & xc.RMX_OUT_DEVICE:
                o = [torch.full((W, P), -77, dtype=torch.int32, device=dev), torch.full((W, P), -77.0, device=dev),
                     torch.full((W, P), -77.0, device=dev)]
                ptrs = [C.c_void_p(t.data_ptr()) for t in o]
            else:
                o = [np.full((W, P), -77, np.int32), np.full((W, P), -77, np.float32), np.full((W, P), -77, np.float32)]
                ptrs = [a.ctypes.data_as(C.c_void_p) for a in o]
            rc = lib.rmx_xcorr_batch(eng._ctx, in_ptr, W, None, 0, ptrs[0], ptrs[1], ptrs[2], flags)
            assert rc == 0, (flags, lib.rmx_last_error(eng._ctx))
            assert lib.rmx_synchronize(eng._ctx) == 0
            got[flags] = [t.cpu().numpy() if flags & xc.RMX_OUT_DEVICE else t for t in o]
    _assert_parity(*got[0], ri, rf, rp)
    for flags in (1, 2, 3):
        for a, b in zip(got[0], got[flags]):
            assert np.array_equal(a, b), flags


@pytest.mark.parametrize("N", [256, 4096, 16384])
def test_abs_squared_search_near_ties(xc, N):
    """VERDICT r04 #7d.  The kernels search the maximum of |c|^2 (one fma per lag) and take the square root of the winner
    and its two neighbours only; numpy searches the maximum of |c| = hypot(re, im) rounded to float32.  Rounding the root
    can merge two different squares into one float32 |c| (numpy then takes the LOWER index) or order them either way, so
    on candidates a few ulp apart the two definitions may disagree.  Pinned expectation: whenever the GPU's lag is not the
    oracle's, the oracle's own two largest magnitudes are within 1e-5 and the GPU took the oracle's second candidate; peak
    values agree to 1e-5 either way; and the deviation never shows on anything wider than 1e-6 (k = +-4 ulp cases are
    checked to be inside that band, i.e. the construction really produces near-ties)."""
    e = near_tie_windows(N, 31 + N)
    ri, rf, rp = orc.xcorr_batch_literal(e)
    pl = orc.pair_list(2)
    margin, second = _top2(e, pl)
    assert np.all(margin <= 1e-6), margin.ravel()            # the construction: all nine windows are near-ties
    with xc.XcorrEngine(2, N, e.shape[0]) as eng:
        li, lf, pk = eng.correlate(e)
    bad = li != ri
    assert np.all(li[bad] == second[bad]), (li.ravel(), ri.ravel(), second.ravel())
    assert np.allclose(pk, rp, rtol=1e-5)
    ok = ~bad
    got, ref = li + lf.astype(np.float64), ri + rf
    assert np.all(np.abs(got[ok] - ref[ok]) <= TOL * np.maximum(np.abs(ref[ok]), 1.0))


@pytest.mark.parametrize("logn", [21, 22])
def test_longest_windows(xc, logn):
    """The longest window lengths the ABI accepts (2 Mi and 4 Mi samples: L = 2^22, 2^23 -- row passes of
    4096 / 8192 points) against the literal oracle, one pair."""
    N = 1 << logn
    iq, delays = rm.synth.make_windows(1, 2, N, 2.4e6, seed=900 + logn)
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    with xc.XcorrEngine(2, N, 1) as eng:
        li, lf, pk = eng.correlate(iq)
    _assert_parity(li, lf, pk, ri, rf, rp)
    assert abs((li + lf)[0, 0] - (delays[0, 1] - delays[0, 0])) < 0.5


def test_cfg4_full_size_properties(xc):
    """BASELINE cfg4 per GPU: 16 buoys (120 pairs) x 10 frequency channels x 512 windows = 5120 windows
    of 4096 samples, generated on the device.  No oracle at this size: ground truth (the generator's
    delays), closure lag(i,k) = lag(i,j) + lag(j,k) and a 64-window oracle subset."""
    import torch
    import bench
    B, N, W, fs = 16, 4096, 5120, 10e6
    dev = torch.device("cuda", 0)
    x, delays = bench.synth_on_device(torch, dev, W, B, N, fs, seed=1004)
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, P), dtype=torch.float32, device=dev)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        torch.cuda.synchronize()
    li, lf = lag.cpu().numpy(), frac.cpu().numpy()
    pairs = orc.pair_list(B)
    est = li + lf.astype(np.float64)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    good = np.abs(est - true) < 0.5
    assert good.mean() > 0.999, good.mean()          # 10 dB SNR: the rare miss is a noise peak
    idx = {(int(i), int(j)): q for q, (i, j) in enumerate(pairs)}
    clo = est[:, idx[(0, 1)]] + est[:, idx[(1, 2)]] - est[:, idx[(0, 2)]]
    ok3 = good[:, idx[(0, 1)]] & good[:, idx[(1, 2)]] & good[:, idx[(0, 2)]]
    assert np.abs(clo[ok3]).max() < 0.35             # three independent sub-sample estimates
    sub = x[:64].cpu().numpy().view(np.complex64).reshape(64, B, N)
    ri, rf, rp = orc.xcorr_batch_fast(sub, workers=8)
    assert np.array_equal(li[:64], ri)
    ref = ri + rf
    assert np.all(np.abs(est[:64] - ref) <= TOL * np.maximum(np.abs(ref), 1.0))


def test_cfg5_shape_ground_truth(xc):
    """BASELINE cfg5 in one window: 32 buoys (496 pairs), 262144-sample windows at 20 MS/s, a 5-bin
    Doppler grid (the oracle would need ~10 minutes here): every pair's lag within a sample of the
    generator's delay and, for pairs whose relative Doppler sits on the grid, that bin."""
    B, N, fs, D = 32, 262144, 20e6, 5
    step = 50.0 / fs
    grid = (np.arange(D) - D // 2) * step
    rng = np.random.default_rng(5)
    offs = rng.integers(-1, 2, size=B) * step              # per-buoy Doppler in {-50, 0, +50} Hz
    iq, delays = rm.synth.make_windows(1, B, N, fs, seed=1005, doppler_cps=offs)
    with xc.XcorrEngine(B, N, 1) as eng:
        dop, li, lf, pk = eng.caf(iq, grid)
    pairs = orc.pair_list(B)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    assert np.all(np.abs(li + lf - true) < 1.0)
    rel = np.rint((offs[pairs[:, 1]] - offs[pairs[:, 0]]) / step).astype(int) + D // 2     # in 0..4
    assert np.array_equal(dop[0], rel)


def test_cfg5_full_doppler_grid(xc):
    """BASELINE configs[4] as stated: 32 buoys (496 pairs), N = 262144 at 20 MS/s, the whole +-500 Hz
    grid in 50 Hz steps (21 hypotheses), one of a GPU's 8 windows.  Ground truth for every pair (lag
    within a sample, Doppler bin exact for on-grid offsets) and the oracle on four pairs."""
    B, N, fs, D = 32, 262144, 20e6, 21
    step = 50.0 / fs
    grid = (np.arange(D) - D // 2) * step
    rng = np.random.default_rng(6)
    offs = rng.integers(-5, 6, size=B) * step              # per-buoy Doppler in +-250 Hz: pairs within +-500 Hz
    iq, delays = rm.synth.make_windows(1, B, N, fs, seed=1005, doppler_cps=offs)
    with xc.XcorrEngine(B, N, 1) as eng:
        dop, li, lf, pk = eng.caf(iq, grid)
    pairs = orc.pair_list(B)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    assert np.all(np.abs(li + lf - true) < 1.0)
    rel = np.rint((offs[pairs[:, 1]] - offs[pairs[:, 0]]) / step).astype(int) + D // 2
    assert rel.min() >= 0 and rel.max() <= D - 1
    assert np.array_equal(dop[0], rel)
    sel = np.array([0, 100, 300, 495])
    rd, ri, rf, rp = orc.caf_batch(iq, grid, pairs[sel])
    assert np.array_equal(dop[:, sel], rd)
    _assert_parity(li[:, sel], lf[:, sel], pk[:, sel], ri, rf, rp)


def test_cfg5_shape_of_one_gpu_all_eight_windows(xc):
    """BASELINE configs[4] at the shape ONE GPU owns (64 windows sharded over 8 GPUs = 8 windows): 32 buoys (496 pairs),
    N = 262144 at 20 MS/s, 21 hypotheses, W = 8 in one rmx_caf_batch call (the suite only ran one window).
    Ground truth on every window -- lag within a sample for all 8 x 496 pair-windows, Doppler bin exact for the on-grid
    offsets -- and the oracle on the same four pairs of window 0 as test_cfg5_full_doppler_grid."""
    B, N, fs, D, W = 32, 262144, 20e6, 21, 8
    step = 50.0 / fs
    grid = (np.arange(D) - D // 2) * step
    rng = np.random.default_rng(6)
    offs = rng.integers(-5, 6, size=B) * step              # per-buoy Doppler in +-250 Hz: pairs within +-500 Hz
    iq, delays = rm.synth.make_windows(W, B, N, fs, seed=1005, doppler_cps=offs)
    with xc.XcorrEngine(B, N, W) as eng:
        dop, li, lf, pk = eng.caf(iq, grid)
    pairs = orc.pair_list(B)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    assert li.shape == (W, 496) and np.all(np.abs(li + lf - true) < 1.0)
    rel = np.rint((offs[pairs[:, 1]] - offs[pairs[:, 0]]) / step).astype(int) + D // 2
    assert np.array_equal(dop, np.broadcast_to(rel, (W, 496)))
    sel = np.array([0, 100, 300, 495])
    rd, ri, rf, rp = orc.caf_batch(iq[:1], grid, pairs[sel])
    assert np.array_equal(dop[:1, sel], rd)
    _assert_parity(li[:1, sel], lf[:1, sel], pk[:1, sel], ri, rf, rp)


def test_flat_peak_rule_on_short_noisy_windows(xc):
    """The two documented parity exceptions, pinned where the suite sees them (include/rmx.h next to lag_int / lag_frac;
    VERDICT r03 item 4).  Short windows at 0 - 3 dB are where they occur: >= 1e5 seeded pair-windows of N = 16, 32, 64.
      * integer lag: bit-exact, or the oracle's own two largest magnitudes are within 1e-5 AND the GPU's lag is the
        oracle's second candidate;
      * fractional lag: within 1e-5 * max(|lag|, 1), or within FOUR times what one float32 ulp on each of the oracle's
        own three taps moves the parabola's vertex (oracle.parabola_ulp_bound: a flat peak, a - 2b + c small against b);
      * nothing else; and the exceptions stay rare (a handful per 1e5), so a broken kernel cannot hide behind them."""
    total = n_int_excused = n_flat = 0
    worst = 0.0
    for N, W, snr, seed in ((16, 1300, 0.0, 41), (32, 1300, 0.0, 42), (64, 1000, 3.0, 43)):
        B = 8
        iq, _ = rm.synth.make_windows(W, B, N, 10e6, seed=seed, snr_db=snr)
        ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
        with xc.XcorrEngine(B, N, W) as eng:
            li, lf, pk = eng.correlate(iq)
        plist = orc.pair_list(B)
        total += li.size
        bad = li != ri
        for w, q in zip(*np.nonzero(bad)):
            margin, _, second = orc.peak_top2(iq[w, plist[q, 0]], iq[w, plist[q, 1]])
            assert margin <= TOL and li[w, q] == second, f"N={N} window {w} pair {q}: integer lag {li[w, q]} vs {ri[w, q]}, margin {margin:.2e}"
            n_int_excused += 1
        ok = ~bad
        ref = ri + rf
        rel = np.where(ok, np.abs(li + lf.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0), 0.0)
        worst = max(worst, float(rel.max()))
        for w, q in zip(*np.nonzero(rel > TOL)):
            bound = orc.parabola_ulp_bound(iq[w, plist[q, 0]], iq[w, plist[q, 1]])
            assert rel[w, q] <= 4.0 * bound, f"N={N} window {w} pair {q}: lag off by {rel[w, q]:.2e}, 4 ulp bounds = {4 * bound:.2e}"
            n_flat += 1
        assert np.allclose(pk[ok], rp[ok], rtol=1e-5, atol=0)
    assert total >= 100_000
    assert n_int_excused <= 20 and n_flat <= 20, (n_int_excused, n_flat)
    print(f"flat-peak rule: {total} pair-windows, {n_int_excused} near-tie integer lags, {n_flat} flat-peak lags, worst rel {worst:.2e}")


def test_cfg2_full_size_properties(xc):
    """BASELINE cfg2: 3 buoys, 64 windows of 2^20 samples at 2.4 MS/s (four-step path), generated on the
    device: ground truth, closure, and the literal oracle on two of the windows."""
    import torch
    import bench
    B, N, W, fs = 3, 1 << 20, 64, 2.4e6
    dev = torch.device("cuda", 0)
    x, delays = bench.synth_on_device(torch, dev, W, B, N, fs, seed=1002)
    lag = torch.zeros((W, 3), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, 3), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, 3), dtype=torch.float32, device=dev)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        torch.cuda.synchronize()
    li, lf, pk = lag.cpu().numpy(), frac.cpu().numpy(), peak.cpu().numpy()
    est = li + lf.astype(np.float64)
    pairs = orc.pair_list(B)
    true = delays[:, pairs[:, 1]] - delays[:, pairs[:, 0]]
    assert np.abs(est - true).max() < 0.25            # 2^20 samples at 10 dB: far below a sample
    # (0,1) + (1,2) = (0,2) up to the bias of three parabolic interpolations of a band-limited peak
    assert np.abs(est[:, 0] + est[:, 2] - est[:, 1]).max() < 0.5
    sub = x[:2].cpu().numpy().view(np.complex64).reshape(2, B, N)
    ri, rf, rp = orc.xcorr_batch_literal(sub)
    _assert_parity(li[:2], lf[:2], pk[:2], ri, rf, rp)


def test_repeated_calls_are_bit_identical(xc):
    """The fused kernel has no atomics on its data path; its LDS protocol (alternating exchange images,
    record ring, half-transform staggering, persistent workgroups) must give the same bits on every
    run: 12 repeats over 1024 windows, any race would show up as a differing lag, fraction or peak."""
    iq, _ = rm.synth.make_windows(1024, 8, 4096, 10e6, seed=4242)
    with xc.XcorrEngine(8, 4096, 1024) as eng:
        li0, lf0, pk0 = eng.correlate(iq)
        for _ in range(11):
            li, lf, pk = eng.correlate(iq)
            assert np.array_equal(li, li0) and np.array_equal(lf, lf0) and np.array_equal(pk, pk0)
    with xc.XcorrEngine(8, 4096, 1024) as eng2:      # and across engines
        li, lf, pk = eng2.correlate(iq)
        assert np.array_equal(li, li0) and np.array_equal(lf, lf0) and np.array_equal(pk, pk0)


@pytest.mark.parametrize("N", [16, 64, 512, 2048, 8192, 32768, 131072])
def test_every_path_by_window_length(xc, N):
    """One seeded case per kernel family boundary: LDS-resident transforms up to L = 16384 (N = 8192 is
    the reference's iq_stream_client capture length), four-step above; against the oracle."""
    W = 3 if N <= 8192 else 1
    iq, delays = rm.synth.make_windows(W, 3, N, 2.4e6, seed=700 + N % 997)
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    margin, second = _top2(iq[:W], orc.pair_list(3))
    with xc.XcorrEngine(3, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
    _assert_parity(li, lf, pk, ri, rf, rp, margin, second)


def test_mixed_call_patterns_on_one_engine(xc):
    """Fused and unfused calls interleaved on one ctx (the unfused path allocates its per-window scratch on
    first use and the fused path keeps working after it), shrinking and growing window counts, and the
    IQ -> lags -> positions chain entirely through device pointers on one stream."""
    import ctypes as C
    import torch
    B, N = 5, 4096
    iq, delays = rm.synth.make_windows(40, B, N, 10e6, seed=808)
    sel = np.array([[0, 4], [2, 3], [1, 4]], np.int32)
    allp = orc.pair_list(B)
    cols = [int(np.where((allp == p).all(axis=1))[0][0]) for p in sel]
    with xc.XcorrEngine(B, N, 40) as eng:
        s0 = eng.scratch_bytes()
        a = eng.correlate(iq)                       # fused
        b = eng.correlate(iq, sel)                  # unfused: custom pairs
        assert eng.scratch_bytes() >= s0            # (40 windows < 256 CUs: both paths need 40 window slots)
        c = eng.correlate(iq[:7])                   # fused again, fewer windows
        d = eng.correlate(iq)                       # and all of them
        assert np.array_equal(a[0][:, cols], b[0]) and np.allclose(a[1][:, cols], b[1], atol=TOL)
        assert all(np.array_equal(x[:7], y) for x, y in zip(a, c))
        assert all(np.array_equal(x, y) for x, y in zip(a, d))
        # device chain
        dev = torch.device("cuda", 0)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        x = torch.from_numpy(iq.view(np.float32).reshape(40, B, N, 2)).to(dev)
        P = len(allp)
        lag = torch.zeros((40, P), dtype=torch.int32, device=dev)
        frac = torch.zeros((40, P), dtype=torch.float32, device=dev)
        peak = torch.zeros((40, P), dtype=torch.float32, device=dev)
        pos = torch.zeros((40, 3), dtype=torch.float64, device=dev)
        cost = torch.zeros(40, dtype=torch.float64, device=dev)
        its = torch.zeros(40, dtype=torch.int32, device=dev)
        eng.correlate_device(x.data_ptr(), 40, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        rng = np.random.default_rng(1)
        buoys = np.ascontiguousarray(rng.normal(0, 2e4, (B, 3)) + np.array([3.9e6, -7e3, 4.9e6]))
        lib = xc.load_library()
        vp = lambda tns: C.c_void_p(tns.data_ptr())   # noqa: E731
        rc = lib.rmx_solve_batch(eng._ctx, buoys.ctypes.data_as(C.c_void_p), B, None, P, vp(lag), vp(frac), None,
                                 10e6, 40, 60, vp(pos), vp(cost), vp(its), 3)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(lag.cpu().numpy(), a[0]) and np.array_equal(frac.cpu().numpy(), a[1])
        hp, hc, hi = eng.solve(buoys, a[0], a[1], 10e6)
        assert np.array_equal(pos.cpu().numpy(), hp) and np.array_equal(its.cpu().numpy(), hi)


@pytest.mark.parametrize("N,B", [(16384, 2), (16384, 4), (32768, 3), (65536, 4), (131072, 2), (262144, 4), (524288, 3)])
def test_fused_row_kernel_by_row_length(xc, N, B, opts):
    """Four-step engines with up to 4 buoys run both row passes in one kernel (g_rows_fused, compiled for rows of
    2^9 .. 2^12 points): every row length and buoy count against the oracle, a custom pair list with a reversed
    pair through the same kernel, and the two-kernel row passes (RMX_FUSED=0) on the same input."""
    W = 2
    iq, _ = rm.synth.make_windows(W, B, N, 2.4e6, seed=900 + B + N % 991)
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    margin, second = _top2(iq[:W], orc.pair_list(B))
    custom = np.array([(B - 1, 0), (0, 1), (1, 1)], np.int32)      # reversed, plain, autocorrelation
    opts("fused", 2)       # (two windows would not fill the chip: the engine would pick the two-kernel passes)
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
        ci, cf, cp = eng.correlate(iq, custom)
    _assert_parity(li, lf, pk, ri, rf, rp, margin, second)
    oi, of_, op = orc.xcorr_batch_literal(iq, custom)
    _assert_parity(ci, cf, cp, oi, of_, op)
    opts("fused", 0)
    with xc.XcorrEngine(B, N, W) as eng:
        ui, uf, up = eng.correlate(iq)
    assert np.array_equal(ui, li) and np.all(np.abs(uf - lf) <= TOL) and np.allclose(up, pk, rtol=1e-5)


@pytest.mark.parametrize("N,B", [(256, 2), (256, 4), (512, 3), (1024, 4), (1024, 2), (2048, 3), (2048, 4)])
def test_whole_window_kernel_by_length(xc, N, B, opts):
    """LDS-resident lengths with up to 4 buoys run whole windows in one kernel (g_win_fused, compiled for L = 2^9 ..
    2^12; the spectra exist in registers only): every length and buoy count against the oracle on complex64 and raw
    uint8 input, a window count that leaves the last workgroup partly empty, a custom pair list with a reversed, a
    repeated and an autocorrelation pair, and the two-kernel path (RMX_WFUSED=0) on the same input."""
    W = 19
    iq, _, raw = rm.synth.make_windows(W, B, N, 2.4e6, seed=700 + B + N % 977, return_u8=True)
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    margin, second = _top2(iq[:W], orc.pair_list(B))
    custom = np.array([(B - 1, 0), (0, 1), (1, 1), (0, 1)], np.int32)
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
        l8, f8, p8 = eng.correlate(raw)
        ci, cf, cp = eng.correlate(iq, custom)
    _assert_parity(li, lf, pk, ri, rf, rp, margin, second)
    assert np.array_equal(li, l8) and np.array_equal(lf, f8) and np.array_equal(pk, p8)
    oi, of_, op = orc.xcorr_batch_literal(iq, custom)
    _assert_parity(ci, cf, cp, oi, of_, op)
    opts("wfused", 0)
    with xc.XcorrEngine(B, N, W) as eng:
        ui, uf, up = eng.correlate(iq)
    assert np.array_equal(ui, li) and np.all(np.abs(uf - lf) <= TOL) and np.allclose(up, pk, rtol=1e-5)


@pytest.mark.parametrize("N,B", [(256, 5), (512, 8), (1024, 6), (2048, 8), (2048, 16), (8192, 3), (8192, 8), (8192, 2), (8192, 4),
                                 (16384, 2), (16384, 3), (16384, 8), (16384, 5)])
def test_whole_window_scratch_kernel_by_length(xc, N, B, opts):
    """Every other shape with 512 <= L <= 16384 -- more than four buoys, or N = 8192 (the capture length of
    iq_stream_client.py:459) -- runs whole windows in one persistent kernel with the spectra in a per-workgroup
    scratch (g_win_scr, compiled per length; at L = 16384 g_win_scr14: one 136 KiB transform per CU, 512 threads x two
    butterflies per pass; option wscr14 = 0 selects the 1024-thread build): against the
    oracle on complex64 and raw uint8 input, more windows than one pass of the grid on the small lengths, a custom
    pair list (reversed, repeated, autocorrelation), and the previous path (option wscr = 0: two-kernel LDS path or
    four-step) on the same input.
    N = 16384 (the capture length of buoy_node.py:364; L = 32768 does not fit the LDS): g_win_eo15 (win_eo.hpp) -- the
    even-bin and odd-bin halves as two LDS-resident 16384-point transforms, the even half's result kept in registers
    -- with 2, 3, 5 and 8 buoys; 5 buoys with more windows than workgroups (persistent loop, scratch reuse)."""
    W = 700 if N <= 512 else ((530 if B == 4 else 5) if N == 8192 else (300 if (N, B) == (16384, 5) else (4 if N == 16384 else 9)))    # (530 / 300: every persistent workgroup takes several windows)
    iq, _, raw = rm.synth.make_windows(W, B, N, 2.4e6, seed=800 + B + N % 977, return_u8=True)
    sub = slice(0, min(W, 12))                                   # literal oracle on the first windows, the rest by consistency
    ri, rf, rp = orc.xcorr_batch_literal(iq[sub])
    margin, second = _top2(iq[:ri.shape[0]], orc.pair_list(B))
    custom = np.array([(B - 1, 0), (0, 1), (1, 1), (0, 1)], np.int32)
    opts("wscr", 2)        # (a few windows would not fill the chip: the engine would pick the per-transform kernels)
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
        l8, f8, p8 = eng.correlate(raw)
        ci, cf, cp = eng.correlate(iq[sub], custom)
    _assert_parity(li[sub], lf[sub], pk[sub], ri, rf, rp, margin, second)
    assert np.array_equal(li, l8) and np.array_equal(lf, f8) and np.array_equal(pk, p8)
    oi, of_, op = orc.xcorr_batch_literal(iq[sub], custom)
    _assert_parity(ci, cf, cp, oi, of_, op)
    if N == 8192:
        # three builds of this length: k_win8kl (the default above: the bin-parity halves on the fused N = 4096 kernel's
        # network, one anchor half resident in LDS -- kwin8k.hpp), g_win_scr14 (option kwin8k = 0: 512 threads x two
        # butterflies per pass) and g_win_scr<14> (+ wscr14 = 0: 1024 threads x one)
        for more in ((("kwin8k", 0),), (("kwin8k", 0), ("wscr14", 0))) if B in (3, 8) else ((("kwin8k", 0),),):
            xc.clear_default_options()
            opts("wscr", 2)
            for k, v in more:
                opts(k, v)
            with xc.XcorrEngine(B, N, W) as eng:
                ai, af, ap = eng.correlate(iq)
                a8 = eng.correlate(raw)
                qi, qf, qp = eng.correlate(iq[sub], custom)
            assert np.array_equal(ai, li) and np.all(np.abs(af - lf) <= TOL) and np.allclose(ap, pk, rtol=1e-5), more
            assert all(np.array_equal(x, y) for x, y in zip((ai, af, ap), a8))
            _assert_parity(qi, qf, qp, oi, of_, op)
        xc.clear_default_options()
    if N == 16384:
        # the product path of this length since round 5: k16_fwd + k16_pairs (kwin16k.hpp: four quarter transforms on the
        # fused N = 4096 kernel's network, spectra of a chunk of windows in HBM, the pairs of a window on one XCD); the run
        # above (wscr = 2) was g_win_eo15
        xc.clear_default_options()
        opts("kwin16k", 2)
        with xc.XcorrEngine(B, N, W) as eng:
            eng.set_option("timing", 1)
            ki, kf, kp = eng.correlate(iq)
            fam = eng.last_timing_by_kernel()
            k8_ = eng.correlate(raw)
            qi, qf, qp = eng.correlate(iq[sub], custom)
        assert set(fam) == {"k16_fwd", "k16_pairs"}, fam
        _assert_parity(ki[sub], kf[sub], kp[sub], ri, rf, rp, margin, second)
        assert np.array_equal(ki, li) and np.all(np.abs(kf - lf) <= TOL) and np.allclose(kp, pk, rtol=1e-5)
        assert all(np.array_equal(x, y) for x, y in zip((ki, kf, kp), k8_))
        _assert_parity(qi, qf, qp, oi, of_, op)
        xc.clear_default_options()
    opts("wscr", 0)
    opts("wfused", 0)
    with xc.XcorrEngine(B, N, W) as eng:
        ui, uf, up = eng.correlate(iq)
    assert np.array_equal(ui, li) and np.all(np.abs(uf - lf) <= TOL) and np.allclose(up, pk, rtol=1e-5)


@pytest.mark.parametrize("kwin8k", [1, 0])
def test_edge_cases_n8192_whole_window_kernels(xc, opts, kwin8k):
    """The edge windows of test_edge_cases_n4096 at the reference's streaming capture length (iq_stream_client.py:459) through
    the whole-window kernels (option wscr = 2; kwin8k = 1: k_win8kl, 0: g_win_scr14): zeros (every lag ties -> the lowest
    index, the kernels' exact-tie path), impulses, peaks on both edges, two peaks of equal height, constant inputs; raw
    uint8 identical.  Random data never reaches the tie path of the wave-level search."""
    N = 8192
    e = np.zeros((6, 2, N), np.complex64)
    e[1, 0, 5] = 1.0; e[1, 1, 25] = 2.0 - 1.0j
    e[2, 0, 0] = 3.0; e[2, 1, N - 1] = 1.0j
    e[3, 0, N - 1] = 1.0; e[3, 1, 0] = -2.0
    e[4, 0, 100] = 1.0; e[4, 1, 93] = 1.0; e[4, 1, 109] = 1.0
    e[5, 0, :] = 4.5 - 2.5j; e[5, 1, :] = -1.5 + 0.5j
    ri, rf, rp = orc.xcorr_batch_literal(e)
    opts("wscr", 2)
    opts("kwin8k", kwin8k)
    with xc.XcorrEngine(2, N, 6) as eng:
        li, lf, pk = eng.correlate(e)
        both = eng.correlate(e, pairs=np.array([[0, 1], [1, 0], [0, 1]], np.int32))
    assert li[0, 0] == -(N - 1) and lf[0, 0] == 0.0 and pk[0, 0] == 0.0      # all ties -> lowest index
    assert li[1, 0] == ri[1, 0] == 20
    assert li[2, 0] == ri[2, 0] == N - 1 and lf[2, 0] == 0.0
    assert li[3, 0] == ri[3, 0] == -(N - 1) and lf[3, 0] == 0.0
    margin4, first4, second4 = orc.peak_top2(e[4, 0], e[4, 1])
    assert {first4, second4} == {-7, 9} and margin4 <= TOL and li[4, 0] in (first4, second4)
    cond = float(rp[5, 0]) / abs(2.0 * float(rp[5, 0]) / N)
    assert li[5, 0] == ri[5, 0] == 0 and abs(lf[5, 0] - rf[5, 0]) <= max(TOL, 4 * 6e-8 * cond)
    assert np.allclose(pk[1:], rp[1:], rtol=1e-5)
    # a custom list (forward-first order in k_win8kl): (0,1), its mirror image, (0,1) again
    assert np.array_equal(both[0][:, 0], li[:, 0]) and np.array_equal(both[0][:, 2], li[:, 0])
    assert np.array_equal(both[0][[1, 2, 3, 5], 1], -li[[1, 2, 3, 5], 0]) and both[0][0, 1] == -(N - 1)


@pytest.mark.parametrize("B,pairs", [(6, None), (8, None), (6, "custom")])
def test_n8192_partial_last_round_through_the_four_step_kernels(xc, opts, B, pairs):
    """N = 8192, W = CUs + 40 windows: k_win8kl takes the full round, the 40 windows of the partial round -- which would cost it a
    whole round -- go through the four-step kernels on the same stream where the cost model of generic_batch says so (from
    about five buoys on with the default pair list; a five-pair custom list of six buoys does not pay and stays whole).
    Every window against the oracle; integer lags equal to the all-k_win8kl run (option wscr = 2) and to g_win_scr14's
    (kwin8k = 0); timing shows which families ran."""
    N = 8192
    W = _device_cus() + 40
    iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=77 + B)[0]
    pl = None if pairs is None else np.array([(B - 1, 0), (0, 2), (0, 1), (3, 1), (3, 4)], np.int32)
    chk = np.r_[0:6, W - 46:W]                                   # oracle on windows of the full round and all of the tail
    ri, rf, rp = orc.xcorr_batch_fast(iq[chk], pl, workers=8)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        li, lf, pk = eng.correlate(iq, pl)
        fam = eng.last_timing_by_kernel()
    assert fam["g_win_*"]["launches"] == 1 and ("g_cols_inv" in fam) == (pairs is None), fam
    _assert_parity(li[chk], lf[chk], pk[chk], ri, rf, rp)
    for k, v in (("wscr", 2), ("kwin8k", 0)):
        xc.clear_default_options()
        opts(k, v)
        with xc.XcorrEngine(B, N, W) as eng:
            eng.set_option("timing", 1)
            ai, af, ap = eng.correlate(iq, pl)
            fam2 = eng.last_timing_by_kernel()
        assert np.array_equal(ai, li) and np.all(np.abs(af - lf) <= TOL) and np.allclose(ap, pk, rtol=1e-5), k
        if k == "wscr":
            assert set(fam2) == {"g_win_*"}                      # forced: no split


def test_n16384_more_windows_than_one_chunk(xc, opts):
    """N = 16384 (buoy_node.py:364), W = CUs + 24 windows of 5 buoys.  Default: k16_fwd / k16_pairs in two chunks (the spectrum
    scratch holds one window per CU), every window of the second chunk and a few of the first against the oracle.  Option
    kwin16k = 0: g_win_eo15 takes the full round and the 24 windows of the partial round go through the four-step kernels
    (round 5's cost model); integer lags equal, and equal to the all-g_win_eo15 run (wscr = 2)."""
    N, B = 16384, 5
    W = _device_cus() + 24
    iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=515)[0]
    chk = np.r_[0:4, W - 26:W]
    ri, rf, rp = orc.xcorr_batch_fast(iq[chk], workers=8)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        li, lf, pk = eng.correlate(iq)
        fam = eng.last_timing_by_kernel()
    assert set(fam) == {"k16_fwd", "k16_pairs"} and fam["k16_fwd"]["launches"] == 2 and fam["k16_pairs"]["launches"] == 2, fam
    _assert_parity(li[chk], lf[chk], pk[chk], ri, rf, rp)
    opts("kwin16k", 0)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        ei, ef, ep = eng.correlate(iq)
        fam = eng.last_timing_by_kernel()
    assert fam["g_win_*"]["launches"] == 1 and "g_cols_inv" in fam, fam
    assert np.array_equal(ei, li) and np.all(np.abs(ef - lf) <= TOL) and np.allclose(ep, pk, rtol=1e-5)
    opts("wscr", 2)
    with xc.XcorrEngine(B, N, W) as eng:
        ai, af, ap = eng.correlate(iq)
    assert np.array_equal(ai, li) and np.all(np.abs(af - lf) <= TOL) and np.allclose(ap, pk, rtol=1e-5)


@pytest.mark.parametrize("B,W", [(3, 1), (8, 1), (2, 7), (16, 2), (5, 9), (3, 40), (4, 33)])
def test_n16384_small_batches_through_the_quarter_kernels(xc, opts, B, W):
    """k16_fwd / k16_pairs on batches far below a chip-full (option kwin16k = 2): (window, buoy) and (window, pair) items
    spread over the chip whatever the number of windows -- one window of 3 buoys: 3 forward and 3 pair workgroups; fewer than
    32 windows that are no multiple of 8: the flat item order instead of the XCD-aware one; 33 windows: XCD 0 with five
    windows, the others with four.  complex64 and raw uint8 against the oracle.  With default options the same batch takes
    these kernels from about 100 transforms (windows x (buoys + pairs)) on and the four-step kernels below."""
    N = 16384
    iq, _, raw = rm.synth.make_windows(W, B, N, 2.4e6, seed=1600 + 10 * B + W, return_u8=True)
    ri, rf, rp = orc.xcorr_batch_fast(iq[:12], workers=8)
    opts("kwin16k", 2)
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        li, lf, pk = eng.correlate(iq)
        fam = eng.last_timing_by_kernel()
        l8 = eng.correlate(raw)
    assert set(fam) == {"k16_fwd", "k16_pairs"}, fam
    _assert_parity(li[:12], lf[:12], pk[:12], ri, rf, rp)
    assert all(np.array_equal(x, y) for x, y in zip((li, lf, pk), l8))
    xc.clear_default_options()
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        di, df, dp = eng.correlate(iq)
        fam = eng.last_timing_by_kernel()
    assert ("k16_pairs" in fam) == (W * (B + B * (B - 1) // 2) >= 100), fam
    assert np.array_equal(di, li) and np.all(np.abs(df - lf) <= TOL) and np.allclose(dp, pk, rtol=1e-5)


@pytest.mark.parametrize("kwin16k", [2, 0])
def test_edge_cases_n16384(xc, opts, kwin16k):
    """The edge windows of test_edge_cases_n4096 at the reference's capture length (buoy_node.py:364) through k16_pairs
    (kwin16k = 2) and g_win_eo15 (kwin16k = 0, wscr = 2): zeros (every lag ties -> the lowest index, the kernels' exact-tie
    path through both search phases of k16_pairs), impulses, peaks on both edges (the two ends of the quarter-butterfly's
    output order), two peaks of equal height, constant inputs; raw-order custom list with the mirror pair."""
    N = 16384
    e = np.zeros((7, 2, N), np.complex64)
    e[1, 0, 5] = 1.0; e[1, 1, 25] = 2.0 - 1.0j
    e[2, 0, 0] = 3.0; e[2, 1, N - 1] = 1.0j
    e[3, 0, N - 1] = 1.0; e[3, 1, 0] = -2.0
    e[4, 0, 100] = 1.0; e[4, 1, 93] = 1.0; e[4, 1, 109] = 1.0
    e[5, 0, :] = 4.5 - 2.5j; e[5, 1, :] = -1.5 + 0.5j
    e[6, 0, 9000] = 1.0; e[6, 1, 808] = 1.0 + 1.0j            # lag -8192: the seam between two quarters of the output
    ri, rf, rp = orc.xcorr_batch_literal(e)
    opts("kwin16k", kwin16k)
    if kwin16k == 0:
        opts("wscr", 2)
    with xc.XcorrEngine(2, N, 7) as eng:
        li, lf, pk = eng.correlate(e)
        both = eng.correlate(e, pairs=np.array([[0, 1], [1, 0], [0, 1]], np.int32))
    assert li[0, 0] == -(N - 1) and lf[0, 0] == 0.0 and pk[0, 0] == 0.0      # all ties -> lowest index
    assert li[1, 0] == ri[1, 0] == 20
    assert li[2, 0] == ri[2, 0] == N - 1 and lf[2, 0] == 0.0
    assert li[3, 0] == ri[3, 0] == -(N - 1) and lf[3, 0] == 0.0
    margin4, first4, second4 = orc.peak_top2(e[4, 0], e[4, 1])
    assert {first4, second4} == {-7, 9} and margin4 <= TOL and li[4, 0] in (first4, second4)
    cond = float(rp[5, 0]) / abs(2.0 * float(rp[5, 0]) / N)
    assert li[5, 0] == ri[5, 0] == 0 and abs(lf[5, 0] - rf[5, 0]) <= max(TOL, 4 * 6e-8 * cond)
    assert li[6, 0] == ri[6, 0] == -8192 and abs(lf[6, 0] - rf[6, 0]) <= TOL
    assert np.allclose(pk[1:], rp[1:], rtol=1e-5)
    assert np.array_equal(both[0][:, 0], li[:, 0]) and np.array_equal(both[0][:, 2], li[:, 0])
    assert np.array_equal(both[0][[1, 2, 3, 5, 6], 1], -li[[1, 2, 3, 5, 6], 0]) and both[0][0, 1] == -(N - 1)


def test_n16384_pair_lists_beyond_the_lds_copy(xc):
    """k16_pairs keeps the pair list in LDS (640 entries); a longer one takes the earlier kernels.  700 pairs (repeats, both
    orders) on 4 buoys x 20 windows against the oracle's answer for the 12 distinct ordered pairs; the first 600 alone
    through k16_pairs."""
    N, B, W = 16384, 4, 20
    iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=62)[0]
    rng = np.random.default_rng(10)
    a = rng.integers(0, B, size=700)
    b = (a + rng.integers(1, B, size=700)) % B
    pairs = np.stack([a, b], axis=1).astype(np.int32)
    distinct = np.array([(i, j) for i in range(B) for j in range(B) if i != j], np.int32)
    ri, rf, rp = orc.xcorr_batch_fast(iq[:6], distinct, workers=8)
    col = {(int(i), int(j)): k for k, (i, j) in enumerate(distinct)}
    idx = np.array([col[(int(i), int(j))] for i, j in pairs])
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        li, lf, pk = eng.correlate(iq, pairs)
        fam_long = eng.last_timing_by_kernel()
        si, sf, sp = eng.correlate(iq, pairs[:600])
        fam_short = eng.last_timing_by_kernel()
    assert "k16_pairs" not in fam_long and set(fam_short) == {"k16_fwd", "k16_pairs"}, (fam_long, fam_short)
    _assert_parity(li[:6], lf[:6], pk[:6], ri[:, idx], rf[:, idx], rp[:, idx])
    _assert_parity(si[:6], sf[:6], sp[:6], ri[:, idx[:600]], rf[:, idx[:600]], rp[:, idx[:600]])
    assert np.array_equal(li[:, :600], si) and np.all(np.abs(lf[:, :600] - sf) <= TOL)


def test_n8192_pair_lists_beyond_the_lds_copy(xc, opts):
    """k_win8kl keeps a custom pair list in LDS (640 entries); a longer one takes g_win_scr14.  700 pairs (repeats, both orders)
    on 6 buoys x 260 windows against the oracle's answer for the 30 distinct ordered pairs."""
    N, B, W = 8192, 6, 260
    iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=61)[0]
    rng = np.random.default_rng(9)
    a = rng.integers(0, B, size=700)
    b = (a + rng.integers(1, B, size=700)) % B
    pairs = np.stack([a, b], axis=1).astype(np.int32)
    distinct = np.array([(i, j) for i in range(B) for j in range(B) if i != j], np.int32)
    ri, rf, rp = orc.xcorr_batch_fast(iq[:8], distinct, workers=8)
    col = {(int(i), int(j)): k for k, (i, j) in enumerate(distinct)}
    idx = np.array([col[(int(i), int(j))] for i, j in pairs])
    opts("wscr", 2)
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq, pairs)
        si, sf, sp = eng.correlate(iq, pairs[:600])              # 600 <= 640: k_win8kl
    _assert_parity(li[:8], lf[:8], pk[:8], ri[:, idx], rf[:, idx], rp[:, idx])
    _assert_parity(si[:8], sf[:8], sp[:8], ri[:, idx[:600]], rf[:, idx[:600]], rp[:, idx[:600]])
    assert np.array_equal(li[:, :600], si) and np.all(np.abs(lf[:, :600] - sf) <= TOL)


@pytest.mark.parametrize("case", range(12))
def test_seeded_random_shapes_and_pair_lists(xc, case):
    """Seeded sweep over (buoys, window length, windows, pair list): every kernel family gets shapes nobody
    hand-picked -- odd buoy counts, custom lists with reversed and repeated pairs, single windows -- against the
    literal oracle."""
    rng = np.random.default_rng(4200 + case)
    logn = int(rng.integers(4, 17))                       # N = 16 .. 65536
    N = 1 << logn
    B = int(rng.integers(2, 10))
    W = int(rng.integers(1, 4)) if logn >= 13 else int(rng.integers(1, 9))
    iq, _ = rm.synth.make_windows(W, B, N, 2.4e6, seed=4300 + case)
    pairs = None
    if case % 2:                                          # custom list: random ordered pairs, i != j, repeats allowed
        P = int(rng.integers(1, 8))
        a = rng.integers(0, B, size=P)
        b = (a + rng.integers(1, B, size=P)) % B
        pairs = np.stack([a, b], axis=1).astype(np.int32)
    ri, rf, rp = orc.xcorr_batch_literal(iq, pairs)
    plist = orc.pair_list(B) if pairs is None else pairs
    margin, second = _top2(iq[:W], plist)
    with xc.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq, pairs) if pairs is not None else eng.correlate(iq)
    _assert_parity(li, lf, pk, ri, rf, rp, margin, second)


def test_fused_kernel_half_order_modes_and_grid_sizes(xc, opts):
    """k_win's scheduling options change scheduling only: every half-order mode (`stag` 0..5) and persistent grids
    smaller than, equal to and larger than the window count (`ncus`: workgroups that take 3 windows, 1 window, none) must
    give the results of the default build bit for bit, on 3 and 8 buoys, complex64 and raw uint8."""
    for B, W in ((8, 37), (3, 300)):
        out = rm.synth.make_windows(W, B, 4096, 10e6, seed=4242 + B, return_u8=True)
        iq, raw = out[0], out[2]
        ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
        opts("small4096", 0)      # the fused kernel also for these few windows (by default they take the per-transform kernels)
        with xc.XcorrEngine(B, 4096, W) as eng:
            base = eng.correlate(iq)
            _assert_parity(*base, ri, rf, rp)
        for key, values in (("stag", (0, 2, 3, 4, 5)), ("ncus", (1, 16, W, 2 * W + 3))):
            for v in values:
                opts("small4096", 0)
                opts(key, v)
                with xc.XcorrEngine(B, 4096, W) as eng:
                    got = eng.correlate(iq)
                    got8 = eng.correlate(raw)
                xc.clear_default_options()
                for a, b, c in zip(base, got, got8):
                    assert np.array_equal(a, b) and np.array_equal(a, c), (key, v, B)


@pytest.mark.parametrize("B,W", [(3, 1), (8, 1), (8, 5), (16, 2), (8, 60), (5, 40)])
def test_small_batches_take_the_per_transform_kernels(xc, opts, B, W):
    """Few windows of N = 4096 (the reference's seam hands over one frequency group at a time) run through k_fwd + the pair
    kernels with the blocks sized to the batch instead of one workgroup per window: same bar against the oracle, complex64
    and uint8 identical, and the same integer lags as the fused kernel (option small4096 = 0) on the same input."""
    out = rm.synth.make_windows(W, B, 4096, 10e6, seed=900 + 17 * B + W, return_u8=True)
    iq, raw = out[0], out[2]
    ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
    with xc.XcorrEngine(B, 4096, max(W, 8)) as eng:
        got = eng.correlate(iq)
        _assert_parity(*got, ri, rf, rp)
        got8 = eng.correlate(raw)
        for a, b in zip(got, got8):
            assert np.array_equal(a, b)
        again = eng.correlate(iq)                       # plan and scratch reused
        for a, b in zip(got, again):
            assert np.array_equal(a, b)
    opts("small4096", 0)
    with xc.XcorrEngine(B, 4096, max(W, 8)) as eng:
        fused = eng.correlate(iq)
    assert np.array_equal(got[0], fused[0])
    assert np.allclose(got[1], fused[1], atol=2e-5) and np.allclose(got[2], fused[2], rtol=1e-5)


def test_one_engine_alternates_between_small_and_large_batches(xc):
    """One ctx, calls of 1, 300, 2, 300 and 1 windows: the dispatch (per-transform kernels below the rule's limit, the fused
    kernel above), the pair plan's block size and the scratch size change from call to call; every call meets the bar and
    repeated batches give identical arrays."""
    B = 8
    out = rm.synth.make_windows(300, B, 4096, 10e6, seed=31337, return_u8=True)
    iq = out[0]
    ri, rf, rp = orc.xcorr_batch_fast(iq, workers=8)
    seen = {}
    with xc.XcorrEngine(B, 4096, 300) as eng:
        for W in (1, 300, 2, 300, 1):
            got = eng.correlate(iq[:W])
            _assert_parity(*got, ri[:W], rf[:W], rp[:W])
            if W in seen:
                for a, b in zip(seen[W], got):
                    assert np.array_equal(a, b)
            seen[W] = got


def _correlate_dev(eng, iq):
    """the engine through device pointers (RMX_IN_DEVICE | RMX_OUT_DEVICE): host numpy in, three host arrays out"""
    import torch
    W, B, N = iq.shape
    dev = torch.device("cuda:0")
    x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32).reshape(W, B, N, 2)).to(dev)
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, P), dtype=torch.float32, device=dev)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    return lag.cpu().numpy(), frac.cpu().numpy(), peak.cpu().numpy()


def _device_cus():
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count


def _tiled_batch(B, W, seed):
    """W windows (512 distinct ones, repeated) with the oracle's answer for each"""
    out = rm.synth.make_windows(min(W, 512), B, 4096, 10e6, seed=seed, return_u8=True)
    reps = (W + 511) // 512
    iq = np.concatenate([out[0]] * reps)[:W]
    ri, rf, rp = orc.xcorr_batch_fast(out[0], workers=8)
    ri, rf, rp = (np.concatenate([a] * reps)[:W] for a in (ri, rf, rp))
    return iq, ri, rf, rp


def test_partial_last_round_goes_through_the_per_transform_kernels(xc, opts):
    """W = k CUs + r windows: the k full rounds run in the fused kernel, the r windows of the last, partial round in the
    per-transform kernels when the cost model prefers that; bar against the oracle on every window, integer lags equal to the
    all-fused run (option small4096 = 0).  Host arrays (at most 512 windows: beyond that the pipelined copy path takes the
    fused kernel for every round) and device pointers; the (8, 4096 + 100) case crosses the chunk boundary at 4096 windows
    and is driven through device pointers, where the last chunk of 100 windows is a partial round of its own."""
    from radio_mapper_amd import xcorr as x
    n_cus = _device_cus()
    for B, W, dev_ptr in ((8, n_cus + 44, False), (8, n_cus + 44, True), (3, 2 * n_cus + 5, True), (8, 4096 + 100, True)):
        iq, ri, rf, rp = _tiled_batch(B, W, 77 + W)
        run = (lambda e: _correlate_dev(e, iq)) if dev_ptr else (lambda e: e.correlate(iq))
        with xc.XcorrEngine(B, 4096, W) as eng:
            got = run(eng)
        _assert_parity(*got, ri, rf, rp)
        opts("small4096", 0)
        with xc.XcorrEngine(B, 4096, W) as eng:
            fused = run(eng)
        x.clear_default_options()
        assert np.array_equal(got[0], fused[0])
        k = (W // n_cus) * n_cus if W < 4096 else 4096   # the full rounds are the same kernel on the same data
        for a, b in zip(got, fused):
            assert np.array_equal(a[:k], b[:k])


@pytest.mark.parametrize("chunk", [16, 100, 300])
def test_partial_round_with_small_chunks(xc, opts, chunk):
    """the partial-round split is decided per chunk of windows.  With chunk_windows below the batch's
    remainder (16 < 44), not a multiple of the CU count (100), or just above it (300) every chunk's own partial round goes
    to the per-transform kernels or stays in the fused one -- never a window of another chunk, never a spectrum slot beyond
    the scratch.  Device pointers (the path that takes the split); every window against the oracle, integer lags equal to
    the all-fused run, and a second call on the same engine identical to the first."""
    from radio_mapper_amd import xcorr as x
    n_cus = _device_cus()
    B, W = 8, n_cus + 44
    iq, ri, rf, rp = _tiled_batch(B, W, 4242)
    opts("chunk_windows", chunk)
    with xc.XcorrEngine(B, 4096, W) as eng:
        got = _correlate_dev(eng, iq)
        again = _correlate_dev(eng, iq)
    _assert_parity(*got, ri, rf, rp)
    for a, b in zip(got, again):
        assert np.array_equal(a, b)
    opts("small4096", 0)
    with xc.XcorrEngine(B, 4096, W) as eng:
        fused = _correlate_dev(eng, iq)
    x.clear_default_options()
    assert np.array_equal(got[0], fused[0])
    assert np.allclose(got[1], fused[1], atol=2e-5) and np.allclose(got[2], fused[2], rtol=1e-5)


def test_per_launch_timing_over_many_partial_rounds(xc, opts):
    """with option timing = 1 every launch is bracketed by two HIP events; the pool used to be sized
    ahead of the chunk loop for at most ONE partial round per call.  chunk_windows = 300 over 5 x 300 + 44 windows of 8
    buoys: every chunk has a partial round of its own (three launches per chunk when it takes the per-transform route).
    The event pool now grows in front of every record; results as without timing, and the per-family read-back adds up."""
    n_cus = _device_cus()
    B, chunk = 8, n_cus + 44
    W = 5 * chunk + 44
    iq, ri, rf, rp = _tiled_batch(B, W, 99)
    opts("chunk_windows", chunk)
    with xc.XcorrEngine(B, 4096, W) as eng:
        plain = _correlate_dev(eng, iq)
        eng.set_option("timing", 1)
        timed = _correlate_dev(eng, iq)
        tm = eng.last_timing()
        fam = eng.last_timing_by_kernel()
        timed2 = _correlate_dev(eng, iq)
    _assert_parity(*timed, ri, rf, rp)
    for a, b, c in zip(plain, timed, timed2):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    # six chunks (5 x (CUs + 44) + 44 windows): all fused = 6 launches; with the split route the five large chunks run one
    # fused round + forward + pairs each and the last chunk (44 windows, behind full rounds) forward + pairs only = 17
    assert (tm["fwd_launches"], tm["pair_launches"]) in ((0, 6), (6, 11)), tm
    assert set(fam) <= {"k_fwd", "k_win|k_pair"} and fam["k_win|k_pair"]["launches"] == tm["pair_launches"]
    assert abs(fam["k_win|k_pair"]["ms"] - tm["pair_ms"]) < 1e-3 and tm["pair_ms"] > 0.0


def test_timing_by_kernel_family_on_the_other_paths(xc):
    """rmx_last_timing_kind on the whole-window, four-step and CAF paths: the families that ran, each with a positive time."""
    cases = ((3, 1024, 64, {"g_win_*"}),                                   # g_win_fused
             (3, 65536, 64, {"g_cols_fwd", "g_rows_fused", "g_cols_inv", "g_final"}),
             (8, 65536, 2, {"g_cols_fwd", "g_rows_fwd", "g_rows_anchor", "g_cols_inv", "g_final"}))
    for B, N, W, want in cases:
        iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=N + B)[0]
        with xc.XcorrEngine(B, N, W) as eng:
            eng.set_option("timing", 1)
            eng.correlate(iq)
            fam = eng.last_timing_by_kernel()
        assert set(fam) == want, (B, N, W, fam)
        assert all(v["ms"] > 0.0 and v["launches"] >= 1 for v in fam.values())
    B, N, W, D = 4, 4096, 2, 5
    iq = rm.synth.make_windows(W, B, N, 2.4e6, seed=5)[0]
    with xc.XcorrEngine(B, N, W) as eng:
        eng.set_option("timing", 1)
        eng.caf(iq, (np.arange(D) - 2) * 50.0 / 2.4e6)
        fam = eng.last_timing_by_kernel()
    assert set(fam) == {"k_fwd", "k_win|k_pair", "k_caf_select"} and fam["k_fwd"]["launches"] == 2


def test_caf_all_hypotheses_in_one_launch(xc):
    """N = 4096, one chunk: rmx_caf_batch runs every hypothesis in one forward and one pair launch (virtual window =
    hypothesis x window).  Against the oracle's per-hypothesis loop on 8 buoys x 3 windows x 21 hypotheses with true offsets
    inside the grid, complex64 and uint8, default and custom pair list; a differing Doppler row is accepted only when the
    oracle's own peak in that row is within 1e-5 of its best."""
    B, W, N, D = 8, 3, 4096, 21
    fs = 2.4e6
    grid = (np.arange(D) - D // 2) * 50.0 / fs
    rng = np.random.default_rng(5)
    true = rng.uniform(-400.0, 400.0, size=(W, B)) / fs
    out = rm.synth.make_windows(W, B, N, fs, seed=2024, return_u8=True, doppler_cps=true)
    iq, raw = out[0], out[2]
    sel = np.array([[0, 7], [3, 1], [2, 6], [6, 2]], np.int32)
    with xc.XcorrEngine(B, N, W) as eng:
        for pairs in (None, sel):
            rd, ri, rf, rp = orc.caf_batch(iq, grid, pairs)
            gd, li, lf, pk = eng.caf(iq, grid, pairs)
            gd8, li8, lf8, pk8 = eng.caf(raw, grid, pairs)
            assert np.array_equal(gd, gd8) and np.array_equal(li, li8) and np.array_equal(lf, lf8) and np.array_equal(pk, pk8)
            plist = orc.pair_list(B) if pairs is None else pairs
            for w, q in zip(*np.nonzero((gd != rd) | (li != ri))):
                i, j = int(plist[q, 0]), int(plist[q, 1])
                y = (iq[w, j] * orc.doppler_phasor(grid[gd[w, q]], N)).astype(np.complex64)
                alt = orc.xcorr_pair(iq[w, i], y)[2]
                assert abs(float(alt) - float(rp[w, q])) <= 1e-5 * float(rp[w, q]), (w, q)
            same = (gd == rd) & (li == ri)
            # a COUNT, not a fraction (VERDICT r04 #7b): two hypotheses tie to 1e-5 only when the true offset sits within
            # ~0.04 Hz of the middle between two grid points -- about one pair-window in a thousand
            assert int((~same).sum()) <= 2, int((~same).sum())
            ref = (ri + rf)[same]
            assert np.all(np.abs((li + lf.astype(np.float64))[same] - ref) <= TOL * np.maximum(np.abs(ref), 1.0))
            assert np.allclose(pk[same], rp[same], rtol=1e-5)
