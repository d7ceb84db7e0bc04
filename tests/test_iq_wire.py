"""IQ wire formats either side of the engine (radio_mapper_amd/iq_wire.py): the reference's JSON form of an IQ excerpt
and the binary capture frame.  CPU only."""
import json

import numpy as np
import pytest

from radio_mapper_amd import iq_wire
from radio_mapper_amd import tdoa_processor as tp


class _RefEncoder(json.JSONEncoder):
    """What iq_stream_client.py:31-44 does to a detection before it goes on the socket (restated: the module itself
    needs `websockets` to import)."""

    def default(self, obj):
        if isinstance(obj, (np.float32, np.float64)):
            return float(obj)
        if isinstance(obj, (np.int32, np.int64)):
            return int(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        if isinstance(obj, complex):
            return str(obj)
        return json.JSONEncoder.default(self, obj)


def _excerpt(n=256, seed=0):
    rng = np.random.default_rng(seed)
    u8 = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    # the decode of buoy_node.py:392-398: (uint8 - 127.5) as float32, I + 1j*Q
    return u8, ((u8[0::2].astype(np.float32) - 127.5) + 1j * (u8[1::2].astype(np.float32) - 127.5)).astype(np.complex64)


def test_reference_json_form_round_trips_exactly():
    _, iq = _excerpt()
    # iq_stream_client.py:237: iq_samples = signal_samples.tolist(); json.dumps(..., cls=NumpyEncoder) then str()s them
    wire = json.dumps({"iq_samples": iq.tolist()}, cls=_RefEncoder)
    back = json.loads(wire)["iq_samples"]
    assert isinstance(back[0], str) and back[0].startswith("(") and back[0].endswith("j)")
    got = iq_wire.parse_complex_list(back)
    assert got.dtype == np.complex64 and np.array_equal(got, iq)


def test_parse_accepts_mixed_items_and_rejects_garbage():
    got = iq_wire.parse_complex_list(["(1.5-2j)", "3j", 2.0, [4, -5], (0.25, 0.5), complex(7, 8), np.complex64(1 - 1j), " (1+1j) "])
    assert np.array_equal(got, np.array([1.5 - 2j, 3j, 2, 4 - 5j, 0.25 + 0.5j, 7 + 8j, 1 - 1j, 1 + 1j], np.complex64))
    assert np.isnan(iq_wire.parse_complex_list(["(nan+0j)"])[0].real)
    for bad in (["1+"], [None], [[1, 2, 3]], ["(1,2)"]):
        with pytest.raises(iq_wire.IqWireError):
            iq_wire.parse_complex_list(bad)
    assert iq_wire.parse_complex_list([]).shape == (0,)


@pytest.mark.parametrize("kind", ["u8", "c64"])
def test_binary_frame_round_trip(kind):
    u8, iq = _excerpt(n=4096, seed=3)
    payload = u8 if kind == "u8" else iq
    buf = iq_wire.pack_iq_frame("BUOY_ÅLPHA", 2.4e6, 121.5e6, 1_700_000_000_123_456_789, payload)
    assert len(buf) == 40 + len("BUOY_ÅLPHA".encode()) + payload.nbytes
    f = iq_wire.unpack_iq_frame(buf)
    assert (f.node_id, f.sample_rate_hz, f.center_freq_hz, f.gps_timestamp_ns) == ("BUOY_ÅLPHA", 2.4e6, 121.5e6, 1_700_000_000_123_456_789)
    assert f.samples.dtype == payload.dtype and np.array_equal(f.samples, payload) and f.n_samples == 4096
    assert iq_wire.unpack_iq_frame(bytearray(buf)).n_samples == 4096 and iq_wire.unpack_iq_frame(memoryview(buf)).node_id == f.node_id


def test_binary_frame_rejects_what_does_not_add_up():
    u8, _ = _excerpt(n=64)
    buf = iq_wire.pack_iq_frame("A", 1e6, 1e8, 5, u8)
    for bad in (buf[:10], buf[:-1], buf + b"\0", b"XXXX" + buf[4:], buf[:4] + bytes([9]) + buf[5:], buf[:5] + bytes([7]) + buf[6:]):
        with pytest.raises(iq_wire.IqWireError):
            iq_wire.unpack_iq_frame(bad)
    huge = bytearray(buf)
    huge[12:16] = (0xFFFFFFFF).to_bytes(4, "little")          # n_samples far beyond the buffer
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.unpack_iq_frame(bytes(huge))
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.pack_iq_frame("A", 1e6, 1e8, 5, u8[:-1])      # odd byte count
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.pack_iq_frame("A", 1e6, 1e8, 5, np.zeros((2, 4), np.complex64))


def test_frames_to_batch_orders_rows_and_applies_the_seam_rule():
    frames = [iq_wire.unpack_iq_frame(iq_wire.pack_iq_frame(n, 2.4e6, 1e8, k, _excerpt(128, seed=k)[0])) for k, n in enumerate("CAB")]
    b = iq_wire.frames_to_batch(frames, ["A", "B", "C"])
    assert b.shape == (3, 256) and b.dtype == np.uint8
    assert np.array_equal(b[0], _excerpt(128, seed=1)[0]) and np.array_equal(b[2], _excerpt(128, seed=0)[0])
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.frames_to_batch(frames, ["A", "B", "D"])
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.frames_to_batch(frames + frames[:1], ["A", "B", "C"])
    odd = iq_wire.unpack_iq_frame(iq_wire.pack_iq_frame("B", 1.0e6, 1e8, 0, _excerpt(128, seed=9)[0]))
    with pytest.raises(iq_wire.IqWireError):
        iq_wire.frames_to_batch([frames[0], frames[1], odd], ["A", "B", "C"])


def test_calculator_takes_the_reference_string_form(monkeypatch, caplog):
    """A detection whose iq_samples is the list of str(complex) the reference puts on the socket reaches the engine as
    the same complex64 window; strings that do not parse are logged and yield no measurements (never an exception)."""
    calc = tp.TDoACalculator()
    pos = {n: tp.BuoyPosition(n, 35.0 + k * 0.01, -97.0, 100.0, 50) for k, n in enumerate("ABC")}
    wins = [_excerpt(64, seed=k)[1] for k in range(3)]
    seen = []

    def fake(iq, pairs=None):
        seen.append(np.asarray(iq))
        return np.zeros((1, 3), np.int32), np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32)

    monkeypatch.setattr(calc, "measure_lags", fake)
    dets = [tp.SignalDetection(n, 121.5, -50, "t", 1000 * k, 35.0, -97.0, 0.9, "beacon",
                               json.loads(json.dumps(wins[k].tolist(), cls=_RefEncoder)), 2.4e6) for k, n in enumerate("ABC")]
    meas = calc.calculate_tdoa_measurements(dets, pos)
    assert len(meas) == 3 and seen[0].shape == (1, 3, 64) and seen[0].dtype == np.complex64
    assert np.array_equal(seen[0][0], np.stack(wins))
    dets[1].iq_samples = ["(1+2j)", "oops"] * 32
    with caplog.at_level("ERROR"):
        assert calc.calculate_tdoa_measurements(dets, pos) == []
    assert any("cannot be decoded" in r.message for r in caplog.records)
