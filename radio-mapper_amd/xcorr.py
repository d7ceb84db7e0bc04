"""ctypes binding of the C-ABI library (include/rmx.h) -- the only compute path of this package.

There is no CPU fallback here: if ``librmx_hip.so`` is missing or no MI355X is visible the
constructor raises.  (The CPU oracle lives in ``oracle/`` and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "librmx_hip.so"
LIB_PATH = os.path.join(_HERE, "csrc", LIB_NAME)

RMX_IN_DEVICE = 1
RMX_OUT_DEVICE = 2
RMX_IN_U8 = 4

_lib = None


class RmxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rmx error {code}: {msg}")
        self.code = code


def library_path() -> str:
    return os.environ.get("RMX_LIBRARY", LIB_PATH)


def load_library():
    """dlopen the HIP library (once).  torch, when importable, is imported first so that this
    library binds to the same libamdhip64 as torch (device pointers and streams are shared)."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    vp, ci, cu = C.c_void_p, C.c_int, C.c_uint
    lib.rmx_version.restype = ci
    lib.rmx_device_count.restype = ci
    lib.rmx_create.argtypes = [C.POINTER(vp), ci, ci, ci, ci, cu]
    lib.rmx_create.restype = ci
    lib.rmx_destroy.argtypes = [vp]
    lib.rmx_destroy.restype = None
    lib.rmx_last_error.argtypes = [vp]
    lib.rmx_last_error.restype = C.c_char_p
    lib.rmx_set_stream.argtypes = [vp, vp]
    lib.rmx_set_stream.restype = ci
    lib.rmx_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    lib.rmx_set_option.restype = ci
    lib.rmx_set_default_option.argtypes = [C.c_char_p, C.c_long]
    lib.rmx_set_default_option.restype = ci
    lib.rmx_clear_default_options.argtypes = []
    lib.rmx_clear_default_options.restype = None
    lib.rmx_xcorr_batch.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, cu]
    lib.rmx_xcorr_batch.restype = ci
    lib.rmx_caf_batch.argtypes = [vp, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, cu]
    lib.rmx_caf_batch.restype = ci
    lib.rmx_solve_batch.argtypes = [vp, vp, ci, vp, ci, vp, vp, vp, C.c_double, ci, ci, vp, vp, vp, cu]
    lib.rmx_solve_batch.restype = ci
    lib.rmx_detect_batch.argtypes = [vp, vp, ci, ci, C.c_float, ci, C.c_double, C.c_float, ci, vp, vp, vp, vp, vp, vp, cu]
    lib.rmx_detect_batch.restype = ci
    lib.rmx_synchronize.argtypes = [vp]
    lib.rmx_synchronize.restype = ci
    lib.rmx_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(ci), C.POINTER(C.c_float),
                                    C.POINTER(ci)]
    lib.rmx_last_timing.restype = ci
    lib.rmx_last_timing_kind.argtypes = [vp, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(ci)]
    lib.rmx_last_timing_kind.restype = ci
    lib.rmx_build_info.argtypes = []
    lib.rmx_build_info.restype = C.c_char_p
    lib.rmx_scratch_bytes.argtypes = [vp]
    lib.rmx_scratch_bytes.restype = C.c_size_t
    _lib = lib
    return lib


EXPORTS = ["rmx_version", "rmx_device_count", "rmx_create", "rmx_destroy", "rmx_last_error",
           "rmx_set_stream", "rmx_set_option", "rmx_set_default_option", "rmx_clear_default_options", "rmx_xcorr_batch", "rmx_caf_batch", "rmx_solve_batch", "rmx_detect_batch", "rmx_synchronize",
           "rmx_last_timing", "rmx_last_timing_kind", "rmx_build_info", "rmx_scratch_bytes"]


def build_info() -> dict:
    """rmx_build_info() parsed: {"source_digest": ..., "arch": ...} of the loaded binary."""
    txt = load_library().rmx_build_info().decode()
    out = {"text": txt}
    for tok in txt.split()[1:]:
        k, _, v = tok.partition("=")
        out[k] = v
    return out


def set_default_option(key: str, value: int) -> None:
    """Kernel-selection default for engines created afterwards (rmx_set_default_option: tests and A/B runs)."""
    lib = load_library()
    if lib.rmx_set_default_option(key.encode(), int(value)) != 0:
        msg = lib.rmx_last_error(None)
        raise RmxError(-1, msg.decode() if msg else "rmx_set_default_option failed")


def clear_default_options() -> None:
    load_library().rmx_clear_default_options()


def apply_env_options(environ=None, strict: bool = True, report=None) -> dict:
    """For the tools/ scripts only: turns RMX_<KEY>=<int> environment variables into default options (the library
    itself never reads the environment).  Returns what was applied.  A variable the library refuses (unknown key, value
    out of range, not an integer) raises ValueError when `strict` -- an A/B run must not silently measure the default
    path -- and is only reported otherwise; `report` (a file object, stderr in the tools) gets one line
    with what was applied and what was refused."""
    import os as _os
    env = _os.environ if environ is None else environ
    done, refused = {}, {}
    for k, v in sorted(env.items()):
        if not k.startswith("RMX_") or k in ("RMX_LIBRARY", "RMX_CPU_THREADS") or k.startswith("RMX_BENCH"):
            continue
        try:
            set_default_option(k[4:].lower(), int(v))
            done[k[4:].lower()] = int(v)
        except (RmxError, ValueError) as e:
            refused[k] = str(e)
    if report is not None and (done or refused):
        print(f"[rmx options] applied {done}" + (f"  REFUSED {refused}" if refused else ""), file=report, flush=True)
    if refused and strict:
        raise ValueError(f"environment options the library refused: {refused}")
    return done


def device_count() -> int:
    return int(load_library().rmx_device_count())


def pair_list(n_buoys: int) -> np.ndarray:
    """(i, j), i < j, nested-loop order of tdoa_processor.py:156-157."""
    return np.array([(i, j) for i in range(n_buoys) for j in range(i + 1, n_buoys)],
                    dtype=np.int32).reshape(-1, 2)


class XcorrEngine:
    """One engine = one rmx_ctx = one GPU.  Batched pairwise cross-correlation lags."""

    def __init__(self, n_buoys: int, n_samples: int, max_windows: int, device: int = 0):
        self._lib = load_library()
        self._ctx = C.c_void_p()
        rc = self._lib.rmx_create(C.byref(self._ctx), device, n_buoys, n_samples, max_windows, 0)
        if rc != 0:
            msg = self._lib.rmx_last_error(None)
            self._ctx = C.c_void_p()
            raise RmxError(rc, msg.decode() if msg else "rmx_create failed")
        self.n_buoys, self.n_samples, self.max_windows, self.device = n_buoys, n_samples, max_windows, device

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            msg = self._lib.rmx_last_error(self._ctx)
            raise RmxError(rc, msg.decode() if msg else "?")

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.rmx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, hip_stream: int):
        self._check(self._lib.rmx_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def set_option(self, key: str, value: int):
        self._check(self._lib.rmx_set_option(self._ctx, key.encode(), int(value)))

    def synchronize(self):
        self._check(self._lib.rmx_synchronize(self._ctx))

    def scratch_bytes(self) -> int:
        return int(self._lib.rmx_scratch_bytes(self._ctx))

    def last_timing(self) -> dict:
        f, p = C.c_float(), C.c_float()
        nf, npair = C.c_int(), C.c_int()
        self._check(self._lib.rmx_last_timing(self._ctx, C.byref(f), C.byref(nf), C.byref(p), C.byref(npair)))
        return dict(fwd_ms=f.value, fwd_launches=nf.value, pair_ms=p.value, pair_launches=npair.value)

    def last_timing_by_kernel(self) -> dict:
        """{kernel family: {"ms": summed HIP-event time, "launches": n}} of the last correlate / caf call (option
        "timing" = 1), families without a launch left out (rmx_last_timing_kind)."""
        out = {}
        kind = 0
        while True:
            name, ms, n = C.c_char_p(), C.c_float(), C.c_int()
            if self._lib.rmx_last_timing_kind(self._ctx, kind, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                break
            if n.value:
                out[name.value.decode()] = {"ms": ms.value, "launches": n.value}
            kind += 1
        return out

    def _check_iq(self, iq):
        """Shape/dtype check of a host window array before its pointer goes to C (which reads
        n_windows * n_buoys * n_samples samples from it).  Returns (contiguous array, flags)."""
        iq = np.asarray(iq)
        flags = 0
        if iq.dtype == np.uint8:
            flags |= RMX_IN_U8
            if iq.ndim != 3 or iq.shape[1] != self.n_buoys or iq.shape[2] != 2 * self.n_samples:
                raise ValueError(f"uint8 iq must be [W][{self.n_buoys}][{2 * self.n_samples}], got {iq.shape}")
        else:
            iq = np.ascontiguousarray(iq, dtype=np.complex64)
            if iq.ndim != 3 or iq.shape[1] != self.n_buoys or iq.shape[2] != self.n_samples:
                raise ValueError(f"iq must be [W][{self.n_buoys}][{self.n_samples}], got {iq.shape}")
        return np.ascontiguousarray(iq), flags

    # -- the hot path ----------------------------------------------------------------------------
    def correlate(self, iq: np.ndarray, pairs: Optional[np.ndarray] = None
                  ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Host arrays in, host arrays out.  iq: complex64 [W][B][N] (or uint8 [W][B][2N] raw
        rtl_sdr I,Q).  Returns (lag_int int32 [W][P], lag_frac float32 [W][P], peak float32 [W][P]);
        lag = lag_int + lag_frac = delay(j) - delay(i) in samples."""
        iq, flags = self._check_iq(iq)
        W = iq.shape[0]
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            P = pairs.shape[0]
            pp = pairs.ctypes.data_as(C.c_void_p)
        else:
            P = self.n_buoys * (self.n_buoys - 1) // 2
            pp = None
        lag_int = np.zeros((W, P), np.int32)
        lag_frac = np.zeros((W, P), np.float32)
        peak = np.zeros((W, P), np.float32)
        if W == 0 or P == 0:
            return lag_int, lag_frac, peak
        self._check(self._lib.rmx_xcorr_batch(
            self._ctx, iq.ctypes.data_as(C.c_void_p), W, pp, P,
            lag_int.ctypes.data_as(C.c_void_p), lag_frac.ctypes.data_as(C.c_void_p),
            peak.ctypes.data_as(C.c_void_p), flags))
        return lag_int, lag_frac, peak

    def caf(self, iq: np.ndarray, doppler_cps, pairs: Optional[np.ndarray] = None):
        """Cross-ambiguity search (rmx_caf_batch): host arrays in and out.  doppler_cps: hypotheses in
        cycles/sample.  Returns (doppler_idx int32, lag_int int32, lag_frac float32, peak float32),
        each [W][P]."""
        iq, flags = self._check_iq(iq)
        W = iq.shape[0]
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            P = pairs.shape[0]
            pp = pairs.ctypes.data_as(C.c_void_p)
        else:
            P = self.n_buoys * (self.n_buoys - 1) // 2
            pp = None
        dc = np.ascontiguousarray(doppler_cps, dtype=np.float64).reshape(-1)
        dop = np.zeros((W, P), np.int32)
        lag_int = np.zeros((W, P), np.int32)
        lag_frac = np.zeros((W, P), np.float32)
        peak = np.zeros((W, P), np.float32)
        if W == 0 or P == 0:
            return dop, lag_int, lag_frac, peak
        self._check(self._lib.rmx_caf_batch(
            self._ctx, iq.ctypes.data_as(C.c_void_p), W, pp, P, dc.ctypes.data_as(C.c_void_p), dc.shape[0],
            dop.ctypes.data_as(C.c_void_p), lag_int.ctypes.data_as(C.c_void_p),
            lag_frac.ctypes.data_as(C.c_void_p), peak.ctypes.data_as(C.c_void_p), flags))
        return dop, lag_int, lag_frac, peak

    def solve(self, buoy_xyz, lag_int, lag_frac, sample_rate_hz: float, weight=None,
              pairs: Optional[np.ndarray] = None, max_iter: int = 60):
        """Batched hyperbolic position solve (rmx_solve_batch), host arrays in and out.
        buoy_xyz [B][3] ECEF metres; lag_int/lag_frac [W][P] as returned by correlate().
        Returns (pos float64 [W][3], cost float64 [W], iters int32 [W])."""
        bx = np.ascontiguousarray(buoy_xyz, dtype=np.float64).reshape(-1, 3)
        li = np.ascontiguousarray(lag_int, dtype=np.int32)
        lf = np.ascontiguousarray(lag_frac, dtype=np.float32)
        W, P = li.shape
        wp = None
        if weight is not None:
            wgt = np.ascontiguousarray(weight, dtype=np.float32)
            wp = wgt.ctypes.data_as(C.c_void_p)
        pp = None
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            pp = pairs.ctypes.data_as(C.c_void_p)
        pos = np.zeros((W, 3), np.float64)
        cost = np.zeros(W, np.float64)
        iters = np.zeros(W, np.int32)
        if W == 0:
            return pos, cost, iters
        self._check(self._lib.rmx_solve_batch(
            self._ctx, bx.ctypes.data_as(C.c_void_p), bx.shape[0], pp, P, li.ctypes.data_as(C.c_void_p),
            lf.ctypes.data_as(C.c_void_p), wp, float(sample_rate_hz), W, int(max_iter),
            pos.ctypes.data_as(C.c_void_p), cost.ctypes.data_as(C.c_void_p), iters.ctypes.data_as(C.c_void_p), 0))
        return pos, cost, iters

    def detect(self, iq: np.ndarray, threshold_db: float = -70.0, distance: int = 10, dc_exclude_bins: float = 0.0,
               min_confidence: float = 0.3, max_peaks: int = 2048):
        """Spectral detection (rmx_detect_batch), host arrays in and out.  iq: complex64 [W][N] (or uint8
        [W][2N]).  Returns a list of W tuples (bins int32, power_db, snr_db, confidence float32 arrays,
        noise_floor_db float)."""
        iq = np.asarray(iq)
        flags = 0
        if iq.dtype == np.uint8:
            flags |= RMX_IN_U8
            W, N = iq.shape[0], iq.shape[1] // 2
        else:
            iq = np.ascontiguousarray(iq, dtype=np.complex64)
            W, N = iq.shape
        iq = np.ascontiguousarray(iq)
        cnt = np.zeros(W, np.int32)
        bins = np.zeros((W, max_peaks), np.int32)
        pw = np.zeros((W, max_peaks), np.float32)
        snr = np.zeros((W, max_peaks), np.float32)
        conf = np.zeros((W, max_peaks), np.float32)
        floor = np.zeros(W, np.float32)
        if W == 0:
            return []
        vp = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        self._check(self._lib.rmx_detect_batch(self._ctx, vp(iq), W, N, float(threshold_db), int(distance),
                                               float(dc_exclude_bins), float(min_confidence), int(max_peaks), vp(cnt),
                                               vp(bins), vp(pw), vp(snr), vp(conf), vp(floor), flags))
        out = []
        for w in range(W):
            n = min(int(cnt[w]), max_peaks)
            out.append((bins[w, :n].copy(), pw[w, :n].copy(), snr[w, :n].copy(), conf[w, :n].copy(), float(floor[w])))
        return out

    def caf_device(self, iq_ptr: int, n_windows: int, doppler_cps, dop_ptr: int, lag_int_ptr: int,
                   lag_frac_ptr: int, peak_ptr: int, pairs: Optional[np.ndarray] = None, u8: bool = False):
        """rmx_caf_batch with device pointers in and out (the Doppler grid and the pair list stay host
        arrays); asynchronous on the ctx stream."""
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            P = pairs.shape[0]
            pp = pairs.ctypes.data_as(C.c_void_p)
        else:
            P = self.n_buoys * (self.n_buoys - 1) // 2
            pp = None
        dc = np.ascontiguousarray(doppler_cps, dtype=np.float64).reshape(-1)
        flags = RMX_IN_DEVICE | RMX_OUT_DEVICE | (RMX_IN_U8 if u8 else 0)
        self._check(self._lib.rmx_caf_batch(self._ctx, C.c_void_p(iq_ptr), n_windows, pp, P,
                                            dc.ctypes.data_as(C.c_void_p), dc.shape[0], C.c_void_p(dop_ptr),
                                            C.c_void_p(lag_int_ptr), C.c_void_p(lag_frac_ptr), C.c_void_p(peak_ptr),
                                            flags))

    def correlate_device(self, iq_ptr: int, n_windows: int, lag_int_ptr: int, lag_frac_ptr: int,
                         peak_ptr: int, pairs: Optional[np.ndarray] = None, u8: bool = False):
        """Device pointers in and out (inputs already resident in HBM); asynchronous on the ctx
        stream."""
        if pairs is not None:
            pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
            P = pairs.shape[0]
            pp = pairs.ctypes.data_as(C.c_void_p)
        else:
            P = self.n_buoys * (self.n_buoys - 1) // 2
            pp = None
        flags = RMX_IN_DEVICE | RMX_OUT_DEVICE | (RMX_IN_U8 if u8 else 0)
        self._check(self._lib.rmx_xcorr_batch(self._ctx, C.c_void_p(iq_ptr), n_windows, pp, P,
                                              C.c_void_p(lag_int_ptr), C.c_void_p(lag_frac_ptr),
                                              C.c_void_p(peak_ptr), flags))
