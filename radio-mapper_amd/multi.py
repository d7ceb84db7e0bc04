"""Multi-device dispatch INSIDE the package: one `rmx_ctx` + one host thread per device, capture windows
block-sharded over them, host-side gather of the 12-byte-per-pair-window results.

The path shards by window (tdoa_processor.py:363 loops the frequency groups independently,
no state crosses windows), so there is no collective: the caller -- the server's single asyncio thread,
central_processor.py:335 -> 397 -> 418 -- hands over one `[W][B][N]` batch, every device gets a contiguous block
of windows (`shard.window_shard`), and the lags come back concatenated in window order.  ctypes releases the GIL
for the duration of `rmx_xcorr_batch`, so plain threads drive the devices concurrently (no torchrun, no process
group); `bench.py --gpus N` remains the one-process-per-GPU harness the driver measures.

An engine is single-owner (include/rmx.h): each `XcorrEngine` here is only ever touched by its own worker thread.
Measured on one MI355X only (devices=[0] and the same-device rehearsal devices=[0, 0]): the scaling beyond one GPU
is unmeasured.
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .shard import window_shard


class MultiXcorrEngine:
    """`XcorrEngine.correlate` / `.caf` over several devices.

    devices=None -> every visible device (`rmx_device_count`).  A device may be listed more than once (two
    contexts on one GPU: the rehearsal mode of the tests).  `engine_factory(n_buoys, n_samples, max_windows,
    device)` exists for the CPU tests of the split / gather logic (a stub engine); the default builds the HIP
    engine and raises, as `XcorrEngine` does, when the library or a GPU is missing.
    """

    def __init__(self, n_buoys: int, n_samples: int, max_windows: int, devices: Optional[Sequence[int]] = None,
                 engine_factory: Optional[Callable] = None):
        if engine_factory is None:
            from . import xcorr

            def engine_factory(b, n, w, device):
                return xcorr.XcorrEngine(b, n, w, device=device)

            if devices is None:
                devices = list(range(xcorr.device_count()))
                if not devices:
                    raise xcorr.RmxError(-2, "no device visible")
        elif devices is None:
            raise ValueError("devices must be given with an engine_factory")
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise ValueError("devices is empty")
        self.n_buoys, self.n_samples, self.max_windows = n_buoys, n_samples, max_windows
        g = len(self.devices)
        # every worker gets the largest block a batch of max_windows can hand it (rank 0's)
        per_dev = max(window_shard(max_windows, 0, g)[1], 1)
        self._engines = []
        try:
            for d in self.devices:
                self._engines.append(engine_factory(n_buoys, n_samples, per_dev, d))
        except Exception:
            self.close()
            raise
        # one thread per engine, and work item r always runs on thread r's engine: an engine never changes hands
        self._pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"rmx-dev{d}-{r}")
                       for r, d in enumerate(self.devices)]
        self._lock = threading.Lock()            # one batch at a time (the engines are stateful)

    # -- plumbing ------------------------------------------------------------------------------------
    def close(self):
        for p in getattr(self, "_pools", []):
            p.shutdown(wait=True)
        self._pools = []
        for e in getattr(self, "_engines", []):
            try:
                e.close()
            except Exception:
                pass
        self._engines = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def blocks(self, n_windows: int) -> List[Tuple[int, int]]:
        """(start, count) of every device's block, in device order (empty blocks included)."""
        return [window_shard(n_windows, r, len(self.devices)) for r in range(len(self.devices))]

    def _run(self, n_windows: int, call: Callable, n_out: int, dtypes, n_pairs: int):
        if n_windows > self.max_windows:
            raise ValueError(f"n_windows {n_windows} > max_windows {self.max_windows}")
        outs = [np.zeros((n_windows, n_pairs), dt) for dt in dtypes]
        with self._lock:
            futs = []
            for r, (s, c) in enumerate(self.blocks(n_windows)):
                if c:
                    futs.append((s, c, self._pools[r].submit(call, self._engines[r], s, c)))
            err = None
            for s, c, f in futs:                 # wait for ALL workers before raising: no engine is left mid-call
                try:
                    res = f.result()
                    for k in range(n_out):
                        outs[k][s:s + c] = res[k]
                except Exception as e:           # noqa: BLE001  (re-raised below)
                    err = err or e
            if err is not None:
                raise err
        return tuple(outs)

    # -- the hot path --------------------------------------------------------------------------------
    def correlate(self, iq: np.ndarray, pairs: Optional[np.ndarray] = None):
        """iq: complex64 [W][B][N] or uint8 [W][B][2N] -> (lag_int [W][P], lag_frac [W][P], peak [W][P])."""
        iq = np.asarray(iq)
        if iq.ndim != 3:
            raise ValueError(f"iq must be [W][B][N], got shape {iq.shape}")
        P = self.n_buoys * (self.n_buoys - 1) // 2 if pairs is None else np.asarray(pairs).reshape(-1, 2).shape[0]
        return self._run(iq.shape[0], lambda eng, s, c: eng.correlate(iq[s:s + c], pairs), 3,
                         (np.int32, np.float32, np.float32), P)

    def caf(self, iq: np.ndarray, doppler_cps, pairs: Optional[np.ndarray] = None):
        """Cross-ambiguity search -> (doppler_idx, lag_int, lag_frac, peak), each [W][P]."""
        iq = np.asarray(iq)
        if iq.ndim != 3:
            raise ValueError(f"iq must be [W][B][N], got shape {iq.shape}")
        P = self.n_buoys * (self.n_buoys - 1) // 2 if pairs is None else np.asarray(pairs).reshape(-1, 2).shape[0]
        return self._run(iq.shape[0], lambda eng, s, c: eng.caf(iq[s:s + c], doppler_cps, pairs), 4,
                         (np.int32, np.int32, np.float32, np.float32), P)
