"""multi.MultiXcorrEngine: the window split / gather logic with a stub engine (CPU), and on the GPU the one-device
and same-device-twice configurations against the single engine (bit-identical).  Beyond one GPU it is unmeasured."""
import threading

import numpy as np
import pytest

import radio_mapper_amd as rm
from radio_mapper_amd import multi
from radio_mapper_amd.shard import window_shard


class StubEngine:
    """Returns, for window w and pair q, lag_int = 1000 * tag(w) + q where tag(w) is the first sample's real part:
    the gather must put every block back at its place whichever thread finishes first."""
    instances = []

    def __init__(self, n_buoys, n_samples, max_windows, device):
        self.n_buoys, self.n_samples, self.max_windows, self.device = n_buoys, n_samples, max_windows, device
        self.threads, self.calls, self.closed = set(), [], False
        StubEngine.instances.append(self)

    def correlate(self, iq, pairs=None):
        self.threads.add(threading.get_ident())
        W = iq.shape[0]
        assert W <= self.max_windows
        P = self.n_buoys * (self.n_buoys - 1) // 2 if pairs is None else len(pairs)
        self.calls.append(W)
        tag = np.asarray(iq)[:, 0, 0].real.astype(np.int32)
        li = 1000 * tag[:, None] + np.arange(P, dtype=np.int32)[None, :]
        return li, np.full((W, P), 0.25, np.float32) * self.device, np.ones((W, P), np.float32)

    def caf(self, iq, doppler_cps, pairs=None):
        li, lf, pk = self.correlate(iq, pairs)
        return np.full_like(li, len(doppler_cps)), li, lf, pk

    def close(self):
        self.closed = True


def _tagged(W, B, N):
    iq = np.zeros((W, B, N), np.complex64)
    iq[:, 0, 0] = np.arange(W)
    return iq


@pytest.mark.parametrize("W,devs", [(10, [0, 1, 2]), (2, [0, 1, 2, 3]), (7, [5]), (0, [0, 1]), (16, [0, 0])])
def test_split_and_gather_with_a_stub_engine(W, devs):
    StubEngine.instances.clear()
    B, N = 4, 16
    with multi.MultiXcorrEngine(B, N, max(W, 1), devices=devs, engine_factory=StubEngine) as eng:
        assert [e.device for e in StubEngine.instances] == devs
        assert all(e.max_windows == max(window_shard(max(W, 1), 0, len(devs))[1], 1) for e in StubEngine.instances)
        assert eng.blocks(W) == [window_shard(W, r, len(devs)) for r in range(len(devs))]
        li, lf, pk = eng.correlate(_tagged(W, B, N))
        assert li.shape == (W, 6) and np.array_equal(li[:, 0], 1000 * np.arange(W)) and np.array_equal(li[:, 5] - li[:, 0], np.full(W, 5))
        for r, e in enumerate(StubEngine.instances):                 # block r ran on engine r, on one thread of its own
            s, c = window_shard(W, r, len(devs))
            assert e.calls == ([c] if c else [])
            assert np.all(lf[s:s + c] == 0.25 * e.device)
        pairs = np.array([(0, 1), (2, 1)], np.int32)
        dop, li2, _, _ = eng.caf(_tagged(W, B, N), np.linspace(-1e-4, 1e-4, 5), pairs)
        assert dop.shape == (W, 2) and (W == 0 or np.all(dop == 5)) and np.array_equal(li2[:, 1] - li2[:, 0], np.ones(W))
        tids = [e.threads for e in StubEngine.instances if e.threads]
        assert all(len(t) == 1 for t in tids) and len(set().union(*tids)) == len(tids) if tids else True
        with pytest.raises(ValueError):
            eng.correlate(_tagged(max(W, 1) + 1, B, N))
    assert all(e.closed for e in StubEngine.instances)


def test_worker_error_surfaces_after_all_workers_finished():
    class Failing(StubEngine):
        def correlate(self, iq, pairs=None):
            if self.device == 1:
                raise RuntimeError("device 1 refused")
            return super().correlate(iq, pairs)

    StubEngine.instances.clear()
    with multi.MultiXcorrEngine(3, 8, 9, devices=[0, 1, 2], engine_factory=Failing) as eng:
        with pytest.raises(RuntimeError, match="device 1 refused"):
            eng.correlate(_tagged(9, 3, 8))
        assert StubEngine.instances[0].calls == [3] and StubEngine.instances[2].calls == [3]


def test_calculator_uses_the_multi_engine_for_several_devices(monkeypatch):
    from radio_mapper_amd import tdoa_processor as tp, xcorr
    made = []
    monkeypatch.setattr(multi, "MultiXcorrEngine",
                        lambda b, n, w, devices=None: made.append((b, n, w, list(devices))) or StubEngine(b, n, w, -1))
    monkeypatch.setattr(xcorr, "XcorrEngine", lambda b, n, w, device=0: made.append((b, n, w, device)) or StubEngine(b, n, w, device))
    monkeypatch.setattr(xcorr, "device_count", lambda: 4)
    iq = _tagged(6, 3, 8)
    tp.TDoACalculator(devices=[0, 1]).measure_lags(iq)
    tp.TDoACalculator(devices="all").measure_lags(iq)
    tp.TDoACalculator(devices=[2]).measure_lags(iq)
    tp.TDoACalculator(device=1).measure_lags(iq)
    assert made == [(3, 8, 8, [0, 1]), (3, 8, 8, [0, 1, 2, 3]), (3, 8, 8, 2), (3, 8, 8, 1)]   # (6 windows: capacity 8)


def _visible_devices():
    """device indices for the all-devices case: counted without initialising a device (torch.cuda.device_count() reads
    sysfs on this image); one entry on a one-GPU box, every card on an 8-GPU node"""
    try:
        import torch
        return list(range(max(torch.cuda.device_count(), 1)))
    except Exception:
        return [0]


@pytest.mark.gpu
@pytest.mark.parametrize("devs", [[0], [0, 0], "all"])
def test_gpu_multi_engine_is_bit_identical_to_the_single_engine(devs):
    """devices=[0], the same-device rehearsal devices=[0, 0] (two contexts, two host threads, one GPU) and EVERY visible
    device (one on the build's GPU box; all eight on the driver's node, where this is the first run of MultiXcorrEngine
    across real devices) against one XcorrEngine on the same windows: identical arrays, complex64 and raw uint8, default
    and custom pairs, CAF."""
    import __graft_entry__ as g
    g.build()
    from radio_mapper_amd import xcorr
    if devs == "all":
        devs = _visible_devices()
        assert len(devs) == xcorr.device_count()
    W, B, N = 37, 5, 4096
    iq, _, raw = rm.synth.make_windows(W, B, N, 10e6, seed=31, return_u8=True)
    pairs = np.array([(4, 0), (1, 2), (2, 2)], np.int32)
    grid = np.linspace(-2e-5, 2e-5, 3)
    with xcorr.XcorrEngine(B, N, W) as one, multi.MultiXcorrEngine(B, N, W, devices=devs) as many:
        for x in (iq, raw):
            for pr in (None, pairs):
                a, b = one.correlate(x, pr), many.correlate(x, pr)
                assert all(np.array_equal(u, v) for u, v in zip(a, b))
        a, b = one.caf(iq[:5], grid), many.caf(iq[:5], grid)
        assert all(np.array_equal(u, v) for u, v in zip(a, b))
