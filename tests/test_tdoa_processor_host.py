"""Host-side mirror of tdoa_processor.py against values recorded from the reference module
(tests/golden/tdoa_conventions.json).  CPU only: no IQ on these detections."""
import json
import os

import numpy as np
import pytest

from radio_mapper_amd import tdoa_processor as tp
from radio_mapper_amd.shard import window_shard


@pytest.fixture(scope="module")
def conv(golden_dir):
    return json.load(open(os.path.join(golden_dir, "tdoa_conventions.json")))


def _proc(conv):
    p = tp.TDoAProcessor()
    for b in conv["buoys"]:
        p.register_buoy(tp.BuoyPosition(*b))
    return p


def test_pair_order_sign_units_confidence(conv):
    p = _proc(conv)
    dets = [tp.SignalDetection(*d) for d in conv["detections"]]
    meas = p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions)
    assert len(meas) == len(conv["measurements"]) == 3
    for m, g in zip(meas, conv["measurements"]):
        assert (m.buoy1_id, m.buoy2_id) == (g["buoy1_id"], g["buoy2_id"])
        assert m.time_difference_ns == g["time_difference_ns"]
        assert m.distance_difference_m == g["distance_difference_m"]
        assert m.confidence == pytest.approx(g["confidence"], rel=1e-15)
        assert m.frequency_mhz == g["frequency_mhz"]


def test_status_grouping_window(conv):
    p = _proc(conv)
    assert p.get_buoy_network_status() == conv["network_status"]
    base = conv["detections"][0][4]
    dets = [tp.SignalDetection(*d) for d in conv["detections"]] + [
        tp.SignalDetection("BUOY_ALPHA", 121.505, -50, "t", base, 0, 0, 0.5),
        tp.SignalDetection("BUOY_BETA", 156.8, -50, "t", base, 0, 0, 0.5)]
    groups = {str(k): [d.buoy_id for d in v] for k, v in p._group_by_frequency(dets).items()}
    assert groups == conv["freq_groups"]
    w = p._filter_by_time_window([tp.SignalDetection("A", 1.0, 0, "t", base, 0, 0, 1.0),
                                  tp.SignalDetection("B", 1.0, 0, "t", base - 9_000_000_000, 0, 0, 1.0),
                                  tp.SignalDetection("C", 1.0, 0, "t", base - 11_000_000_000, 0, 0, 1.0)])
    assert [d.buoy_id for d in w] == conv["time_window"]


def test_reference_error_conventions(conv):
    p = _proc(conv)
    assert p.process_signal_detections([]) == []
    one = [tp.SignalDetection(*conv["detections"][0])]
    assert p.tdoa_calculator.calculate_tdoa_measurements(one, p.buoy_positions) == []
    assert p.hyperbolic_positioner.triangulate_position([], p.buoy_positions) is None
    unknown = [tp.SignalDetection("X1", 1.0, 0, "t", 0, 0, 0, 1.0), tp.SignalDetection("X2", 1.0, 0, "t", 5, 0, 0, 1.0)]
    assert p.tdoa_calculator.calculate_tdoa_measurements(unknown, p.buoy_positions) == []   # unregistered: skipped
    assert p.triangulate_signal(one) is None


def test_positional_construction_and_extension_fields():
    d = tp.SignalDetection("B", 121.5, -55, "t", 1, 51.5, -0.09, 0.9, "emergency")
    assert d.iq_samples is None and d.sample_rate_hz is None
    r = tp.TriangulationResult(1, 2, 3, 4.5, 0.5, 121.5, "x", "t", [], [], "hyperbolic")
    assert r.accuracy_estimate_meters == 4.5


def test_geodesy_roundtrip():
    x, y, z = tp.GeodeticCalculator.lat_lng_to_xyz(35.4676, -97.5164, 120.0)
    lat, lng, alt = tp.GeodeticCalculator.xyz_to_lat_lng(x, y, z)
    assert lat == pytest.approx(35.4676, abs=1e-9) and lng == pytest.approx(-97.5164, abs=1e-9)
    assert alt == pytest.approx(120.0, abs=1e-6)
    brg, dist = tp.GeodeticCalculator.bearing_distance(0, 0, 0, 1)
    assert brg == pytest.approx(90.0) and dist == pytest.approx(6378137.0 * np.pi / 180, rel=1e-12)


def test_window_shard_partitions():
    for n, w in [(4096, 8), (10, 4), (3, 8), (0, 2)]:
        blocks = [window_shard(n, r, w) for r in range(w)]
        assert sum(c for _, c in blocks) == n
        pos = 0
        for s, c in blocks:
            assert s == pos
            pos += c
    with pytest.raises(ValueError):
        window_shard(4, 2, 2)


def _iq_dets(conv, freq, n=64, fs=2.4e6, seed=0):
    rng = np.random.default_rng(seed)
    base = conv["detections"][0][4]
    out = []
    for k, b in enumerate(conv["buoys"]):
        iq = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        out.append(tp.SignalDetection(b[0], freq, -50, "t", base + 1000 * k, b[1], b[2], 0.9, "beacon", iq, fs))
    return out


def test_iq_groups_go_to_the_engine_as_one_batch(conv, monkeypatch):
    """Three frequency groups with IQ windows of one shape -> ONE measure_lags call with [3][B][N]
    (tdoa_processor.py:363-377 loops the groups one by one; here they are one GPU batch)."""
    p = _proc(conv)
    calls = []

    def fake(iq, pairs=None):
        iq = np.asarray(iq)
        calls.append(iq.shape)
        W, B = iq.shape[:2]
        P = B * (B - 1) // 2
        li = np.arange(W * P, dtype=np.int32).reshape(W, P)          # lag q of group w = w*P + q samples
        return li, np.zeros((W, P), np.float32), np.ones((W, P), np.float32)

    monkeypatch.setattr(p.tdoa_calculator, "measure_lags", fake)
    dets = _iq_dets(conv, 121.5, seed=1) + _iq_dets(conv, 156.8, seed=2) + _iq_dets(conv, 243.0, seed=3)
    groups = {}
    orig = p.hyperbolic_positioner.triangulate_position

    def spy(meas, pos):
        groups[meas[0].frequency_mhz] = [m.time_difference_ns for m in meas]
        return orig(meas, pos)

    monkeypatch.setattr(p.hyperbolic_positioner, "triangulate_position", spy)
    p.process_signal_detections(dets)
    assert calls == [(3, 3, 64)]
    fs = 2.4e6
    for g, f in enumerate([121.5, 156.8, 243.0]):
        # window-start difference (1000 ns per buoy index) + round(lag / fs * 1e9), pair order (0,1),(0,2),(1,2)
        want = [1000 * (j - i) + int(round((g * 3 + q) / fs * 1e9)) for q, (i, j) in enumerate([(0, 1), (0, 2), (1, 2)])]
        assert groups[f] == want


def test_engine_failure_is_logged_not_raised(conv, caplog):
    """With IQ on the detections and no usable engine the seam logs on ...TDoACalculator and yields
    nothing: it neither raises (tdoa_processor.py:151-153) nor falls back to the time tags."""
    from radio_mapper_amd import xcorr
    if xcorr.device_count() > 0:
        pytest.skip("a GPU is visible: the engine works")
    p = _proc(conv)
    dets = _iq_dets(conv, 121.5)
    with caplog.at_level("ERROR"):
        assert p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions) == []
        assert p.process_signal_detections(dets) == []
        assert p.triangulate_signal(dets) is None
    assert any("engine failed" in r.message and r.name.endswith("TDoACalculator") for r in caplog.records)
    with pytest.raises(Exception):
        p.tdoa_calculator.measure_lags(np.stack([d.iq_samples for d in dets])[None])   # the engine entry itself raises


def test_partial_or_mismatched_iq_yields_no_measurements(conv, monkeypatch, caplog):
    """Once any detection of a group carries IQ the time tags alone are never used for that group: a group in which
    only some detections carry IQ, or whose windows differ in sample rate or are shorter than 16 samples, is logged on
    ...TDoACalculator and yields no measurements (the log-and-return convention of tdoa_processor.py:151-153); the engine
    is not called, the message is logged once per group, and a group WITHOUT any IQ in the same call still gets the
    reference's timestamp arithmetic (:166)."""
    p = _proc(conv)
    called = []
    monkeypatch.setattr(p.tdoa_calculator, "measure_lags", lambda iq, pairs=None: called.append(1))
    dets = _iq_dets(conv, 121.5, seed=4)
    dets[1].iq_samples = None                                        # only some carry IQ
    other = _iq_dets(conv, 156.8, seed=5)
    other[2].iq_samples = other[2].iq_samples[:9]                    # one window shorter than the engine's minimum
    third = _iq_dets(conv, 243.0, seed=6)
    third[0].sample_rate_hz = 1.0e6                                  # sample rates differ
    plain = [tp.SignalDetection(d.buoy_id, 406.0, -50, "t", d.gps_timestamp_ns, d.lat, d.lng, 0.9) for d in dets]
    with caplog.at_level("ERROR"):
        for group in (dets, other, third):
            assert p.tdoa_calculator.calculate_tdoa_measurements(group, p.buoy_positions) == []
    assert not called
    msgs = [r.message for r in caplog.records if r.name.endswith("TDoACalculator")]
    assert len(msgs) == 3 and "Only some detections" in msgs[0] and "too short" in msgs[1] and "differ" in msgs[2]
    caplog.clear()
    seen = {}
    orig = p.hyperbolic_positioner.triangulate_position
    monkeypatch.setattr(p.hyperbolic_positioner, "triangulate_position",
                        lambda meas, pos: seen.setdefault(meas[0].frequency_mhz, [m.time_difference_ns for m in meas]) and orig(meas, pos))
    with caplog.at_level("ERROR"):
        p.process_signal_detections(dets + other + plain)
    assert not called and list(seen) == [406.0]                      # only the group without IQ is measured ...
    assert seen[406.0] == [1000, 2000, 1000]                         # ... from its time tags alone
    assert len([r for r in caplog.records if r.name.endswith("TDoACalculator")]) == 2   # one line per bad group


def test_clipped_excerpts_are_cut_to_a_common_power_of_two(conv, monkeypatch, caplog):
    """The reference clips an excerpt at the end of the capture buffer (iq_stream_client.py:306-313): a peak in the last
    128 bins gives e.g. 130 samples beside the other buoys' 256.  Every window keeps its start (what its time tag dates),
    so the group is correlated on the first 2^k samples that every window holds -- 130 / 256 / 256 -> 128 -- with a
    warning, instead of yielding nothing; equal lengths that are not a power of two (200) are cut the same
    way; the reference's wire form (a list of str(complex)) takes the same route; a NaN lag from the engine is refused."""
    p = _proc(conv)
    calls = []

    def fake(iq, pairs=None):
        iq = np.asarray(iq)
        calls.append(iq)
        W, B = iq.shape[:2]
        P = B * (B - 1) // 2
        return np.full((W, P), 3, np.int32), np.zeros((W, P), np.float32), np.ones((W, P), np.float32)

    monkeypatch.setattr(p.tdoa_calculator, "measure_lags", fake)
    rng = np.random.default_rng(3)
    win = [(rng.standard_normal(256) + 1j * rng.standard_normal(256)).astype(np.complex64) for _ in range(3)]
    dets = _iq_dets(conv, 121.5, seed=7)
    for d, w in zip(dets, win):
        d.iq_samples = w
    dets[2].iq_samples = win[2][:130]                                # clipped at the end of its buffer
    with caplog.at_level("WARNING"):
        meas = p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions)
    assert len(meas) == 3 and calls[-1].shape == (1, 3, 128)
    for b in range(3):
        assert np.array_equal(calls[-1][0, b], win[b][:128])         # the windows' starts
    assert any("first 128" in r.message for r in caplog.records)
    fs = dets[0].sample_rate_hz
    assert [m.time_difference_ns for m in meas] == [1000 * (j - i) + int(round(3 / fs * 1e9)) for i, j in ((0, 1), (0, 2), (1, 2))]
    for d, w in zip(dets, win):
        d.iq_samples = w[:200]                                       # equal, not a power of two
    assert len(p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions)) == 3
    assert calls[-1].shape == (1, 3, 128)
    for d, w in zip(dets, win):                                      # the JSON form of the reference's NumpyEncoder
        d.iq_samples = [str(complex(v)) for v in w[:130 if d is dets[0] else 256]]
    assert len(p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions)) == 3
    assert calls[-1].shape == (1, 3, 128) and np.allclose(calls[-1][0, 1], win[1][:128])
    monkeypatch.setattr(p.tdoa_calculator, "measure_lags",
                        lambda iq, pairs=None: (np.zeros((1, 3), np.int32), np.full((1, 3), np.nan, np.float32), np.ones((1, 3), np.float32)))
    with caplog.at_level("ERROR"):
        assert p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions) == []
    assert any("non-finite" in r.message for r in caplog.records)


def test_measure_lags_channel_axis_and_engine_cache(monkeypatch):
    calc = tp.TDoACalculator()
    made = []

    class FakeEngine:
        def __init__(self, nb, ns, mw):
            self.max_windows, self.closed = mw, False
            made.append(self)

        def correlate(self, iq, pairs=None):
            W, B = iq.shape[:2]
            P = B * (B - 1) // 2
            return (np.arange(W * P, dtype=np.int32).reshape(W, P), np.zeros((W, P), np.float32),
                    np.zeros((W, P), np.float32))

        def close(self):
            self.closed = True

    import radio_mapper_amd.xcorr as xc
    monkeypatch.setattr(xc, "XcorrEngine", lambda nb, ns, mw, device=0: FakeEngine(nb, ns, mw))
    li, lf, pk = calc.measure_lags(np.zeros((2, 5, 3, 32), np.complex64))       # [C][W][B][N]
    assert li.shape == lf.shape == pk.shape == (2, 5, 3)
    assert np.array_equal(li.reshape(10, 3), np.arange(30).reshape(10, 3))
    with pytest.raises(ValueError):
        calc.measure_lags(np.zeros((3, 32), np.complex64))
    for n in (64, 128, 256, 512, 1024):                                          # five more shapes: LRU of 4
        calc.measure_lags(np.zeros((1, 3, n), np.complex64))
    assert len(calc._engines) == tp.TDoACalculator.MAX_ENGINES
    assert made[0].closed and made[1].closed and not made[-1].closed
    calc.close()
    assert all(e.closed for e in made)


@pytest.fixture(scope="module")
def tri(golden_dir):
    return json.load(open(os.path.join(golden_dir, "triangulate_position.json")))["scenarios"]


@pytest.mark.parametrize("name", ["square4", "penta5", "tri3", "tri3_noisy", "hex6alt", "penta5_noisy"])
def test_triangulate_position_against_reference_results(tri, name):
    """Row a8: HyperbolicPositioning.triangulate_position against the results the imported reference
    produced on the same measurements (tests/golden/make_golden.py:make_triangulation).  The
    horizontal fix and the accuracy figure are what the reference determines; its altitude wanders by
    metres along the unobserved vertical (BFGS stops where the gradient is flat), so that field is
    only held to 100 m."""
    sc = tri[name]
    pos = {b[0]: tp.BuoyPosition(*b) for b in sc["buoys"]}
    meas = [tp.TDoAMeasurement(*m) for m in sc["measurements"]]
    got = tp.HyperbolicPositioning().triangulate_position(meas, pos)
    want = sc["result"]
    if want is None:                     # the reference's BFGS reported failure -> None, no exception
        assert got is None
        return
    assert got is not None
    assert abs(got.estimated_lat - want["estimated_lat"]) < 2e-6          # 0.2 m
    assert abs(got.estimated_lng - want["estimated_lng"]) < 2e-6
    assert abs(got.estimated_altitude - want["estimated_altitude"]) < 100.0
    assert abs(got.accuracy_meters - want["accuracy_meters"]) <= 1e-3 + 1e-4 * want["accuracy_meters"]
    assert got.confidence == pytest.approx(want["confidence"], rel=1e-12)
    assert got.frequency_mhz == want["frequency_mhz"] and got.method == want["method"]
    assert sorted(got.contributing_buoys) == want["contributing_buoys"]
    # and the fix is the transmitter the scenario was built from (noise-free cases: centimetres)
    _, dist = tp.GeodeticCalculator.bearing_distance(got.estimated_lat, got.estimated_lng, *sc["transmitter"][:2])
    assert dist < (0.05 if sc["sigma_m"] == 0 else 10 * sc["sigma_m"])


def test_a_cut_below_the_shortest_reference_excerpt_is_refused(conv, monkeypatch, caplog):
    """one 17-sample excerpt must not turn a 256-sample group into a 16-sample, noise-dominated correlation.
    The reference's clipping (iq_stream_client.py:306-313) leaves at least 129 samples, so 128 is the shortest legitimate
    cut: below it the group is logged and yields nothing; min_cut_samples is a constructor parameter; equal power-of-two
    windows shorter than that (the caller's own choice, no cut) are still correlated."""
    p = _proc(conv)
    calls = []

    def fake(iq, pairs=None):
        iq = np.asarray(iq)
        calls.append(iq.shape)
        W, B = iq.shape[:2]
        P = B * (B - 1) // 2
        return np.zeros((W, P), np.int32), np.zeros((W, P), np.float32), np.ones((W, P), np.float32)

    monkeypatch.setattr(p.tdoa_calculator, "measure_lags", fake)
    dets = _iq_dets(conv, 121.5, n=256, seed=11)
    dets[1].iq_samples = dets[1].iq_samples[:17]
    with caplog.at_level("ERROR"):
        assert p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions) == []
    assert not calls and any("common cut of 16 is below 128" in r.message for r in caplog.records)
    dets[1].iq_samples = _iq_dets(conv, 121.5, n=256, seed=11)[1].iq_samples[:129]
    assert len(p.tdoa_calculator.calculate_tdoa_measurements(dets, p.buoy_positions)) == 3 and calls[-1] == (1, 3, 128)
    short = _iq_dets(conv, 121.5, n=64, seed=12)                      # equal lengths: no cut, no minimum
    assert len(p.tdoa_calculator.calculate_tdoa_measurements(short, p.buoy_positions)) == 3 and calls[-1] == (1, 3, 64)
    assert tp.TDoACalculator(min_cut_samples=16).min_cut_samples == 16 and tp.TDoACalculator().min_cut_samples == 128
