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
