"""CPU oracle for the TDoA cross-correlation hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product path (``radio-mapper_amd/``) never does: it fails loudly when the HIP
library is missing.

What is restated here
---------------------
The reference (physiii/radio-mapper) never *calls* a cross-correlation: ``tdoa_processor.py:20``
imports ``scipy.signal.correlate`` and ``tdoa_processor.py:166`` subtracts two timestamps instead.
The hot path named by BASELINE.json is therefore specified in SURVEY.md §8a-spec from the
reference's conventions, and this oracle is the literal numpy/scipy statement of that spec:

* pair order: nested ``for i / for j in range(i+1, n)`` over the detection list
  (``tdoa_processor.py:156-157``);
* sign: lag/time difference is *buoy2 - buoy1*, positive when buoy2 (index j) received later
  (``tdoa_processor.py:51``)  =>  ``correlate(x_j, x_i)``;
* primitive: ``scipy.signal.correlate(in1, in2, mode='full', method='fft')`` -- the one xcorr
  symbol the reference module exposes (``tdoa_processor.correlate``).  scipy 1.15.3
  ``_signaltools.py``: ``correlate`` -> ``convolve(in1, conj(in2[::-1]))`` -> ``fftconvolve`` with
  FFT length ``next_fast_len(2N-1)`` (= 2N for power-of-two N), single precision kept (pocketfft);
* magnitude: ``np.abs`` on complex64 (float32 hypot); integer peak: ``np.argmax`` (ties -> lowest
  index in 'full' order, i.e. most negative lag first);
* sub-sample: 3-point parabola on the magnitude (a build *choice*),
  evaluated in float64 from the float32 taps; 0 at the two edges or on a flat top;
* units: ``time_difference_ns = round(lag / fs * 1e9)``, ``distance = lag / fs * c`` with
  ``c = 299792458.0`` (``tdoa_processor.py:141,169-170``).

Pinning: no reference test holds a golden vector for this path -- "parity
unpinned" by the reference's own tests.  The oracle is instead pinned against outputs of the
reference module itself generated in the build container (``tests/golden/make_golden.py`` imports
``/root/reference/tdoa_processor.py`` and calls ``tdoa_processor.correlate``); the fixtures are
committed under ``tests/golden/`` and ``tests/test_oracle_golden.py`` checks this file against them.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is present in the build image and on the GPU box; numpy-only fallback otherwise
    from scipy import fft as _sp_fft
    from scipy.signal import correlate as _sp_correlate
    HAVE_SCIPY = True
except Exception:  # pragma: no cover
    _sp_fft = None
    _sp_correlate = None
    HAVE_SCIPY = False

SPEED_OF_LIGHT = 299792458.0  # tdoa_processor.py:141


def pair_list(n_buoys: int) -> np.ndarray:
    """All pairs (i, j), i < j, in the reference's nested-loop order (tdoa_processor.py:156-157)."""
    return np.array([(i, j) for i in range(n_buoys) for j in range(i + 1, n_buoys)],
                    dtype=np.int32).reshape(-1, 2)


def parabolic_offset(m_prev: float, m_peak: float, m_next: float) -> float:
    """3-point parabolic vertex offset in samples, float64."""
    a, b, c = float(m_prev), float(m_peak), float(m_next)
    den = a - 2.0 * b + c
    if den == 0.0:
        return 0.0
    return 0.5 * (a - c) / den


def peak_from_magnitude(m: np.ndarray, n_samples: int):
    """S5+S6 on a 'full'-order magnitude vector m[0..2N-2]: (lag_int, lag_frac, peak)."""
    k = int(np.argmax(m))  # ties -> lowest k
    lag_int = k - (n_samples - 1)
    if 0 < k < m.shape[0] - 1:
        frac = parabolic_offset(m[k - 1], m[k], m[k + 1])
    else:
        frac = 0.0
    return lag_int, frac, float(m[k])


def xcorr_full_scipy(x_i: np.ndarray, x_j: np.ndarray) -> np.ndarray:
    """The literal primitive: correlate(x_j, x_i, 'full', 'fft') -> complex64[2N-1]."""
    if not HAVE_SCIPY:  # pragma: no cover
        return xcorr_full_numpy(x_i, x_j)
    return _sp_correlate(np.ascontiguousarray(x_j, dtype=np.complex64),
                         np.ascontiguousarray(x_i, dtype=np.complex64),
                         mode="full", method="fft")


def xcorr_full_numpy(x_i: np.ndarray, x_j: np.ndarray) -> np.ndarray:
    """numpy.fft restatement: r = IFFT_L(FFT_L(x_j) * conj(FFT_L(x_i))), L = 2N, reordered to
    'full' order (lag -(N-1) .. N-1).  Agrees with the scipy primitive to float32 rounding."""
    x_i = np.ascontiguousarray(x_i, dtype=np.complex64)
    x_j = np.ascontiguousarray(x_j, dtype=np.complex64)
    n = x_i.shape[-1]
    L = 2 * n
    fi = np.fft.fft(x_i, L)
    fj = np.fft.fft(x_j, L)
    r = np.fft.ifft(fj * np.conj(fi)).astype(np.complex64)
    return np.concatenate([r[L - (n - 1):], r[:n]])


def xcorr_pair(x_i: np.ndarray, x_j: np.ndarray, use_scipy: bool = True):
    """One pair-window: (lag_int, lag_frac, peak).  lag = lag_int + lag_frac = delay_j - delay_i."""
    n = x_i.shape[-1]
    c = xcorr_full_scipy(x_i, x_j) if use_scipy else xcorr_full_numpy(x_i, x_j)
    m = np.abs(c)  # float32 hypot on complex64
    return peak_from_magnitude(m, n)


def xcorr_batch_literal(iq: np.ndarray, pairs: np.ndarray | None = None):
    """CPU-baseline variant (i): Python loop over windows x pairs of the literal primitive.

    iq: complex64 [W][B][N].  Returns lag_int int32 [W][P], lag_frac float64 [W][P],
    peak float32 [W][P]."""
    iq = np.asarray(iq)
    W, B, N = iq.shape
    if pairs is None:
        pairs = pair_list(B)
    P = pairs.shape[0]
    lag_int = np.zeros((W, P), np.int32)
    lag_frac = np.zeros((W, P), np.float64)
    peak = np.zeros((W, P), np.float32)
    for w in range(W):
        for q in range(P):
            i, j = int(pairs[q, 0]), int(pairs[q, 1])
            lag_int[w, q], lag_frac[w, q], peak[w, q] = xcorr_pair(iq[w, i], iq[w, j])
    return lag_int, lag_frac, peak


def xcorr_batch_fast(iq: np.ndarray, pairs: np.ndarray | None = None, workers: int = 1):
    """CPU-baseline variant (ii): batched scipy.fft with forward-spectrum reuse (one FFT per
    (window, buoy), one IFFT per (window, pair)).  Same definition, same output contract."""
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    W, B, N = iq.shape
    if pairs is None:
        pairs = pair_list(B)
    P = pairs.shape[0]
    L = 2 * N
    if HAVE_SCIPY:
        spec = _sp_fft.fft(iq, n=L, axis=-1, workers=workers)
    else:  # pragma: no cover
        spec = np.fft.fft(iq, L, axis=-1)
    lag_int = np.zeros((W, P), np.int32)
    lag_frac = np.zeros((W, P), np.float64)
    peak = np.zeros((W, P), np.float32)
    ar = np.arange(W)
    for q in range(P):
        i, j = int(pairs[q, 0]), int(pairs[q, 1])
        prod = spec[:, j, :] * np.conj(spec[:, i, :])
        if HAVE_SCIPY:
            r = _sp_fft.ifft(prod, axis=-1, workers=workers)
        else:  # pragma: no cover
            r = np.fft.ifft(prod, axis=-1)
        r = r.astype(np.complex64, copy=False)
        m = np.abs(np.concatenate([r[:, L - (N - 1):], r[:, :N]], axis=1))
        k = np.argmax(m, axis=1)
        lag_int[:, q] = k - (N - 1)
        peak[:, q] = m[ar, k]
        inner = (k > 0) & (k < 2 * N - 2)
        km = np.clip(k - 1, 0, 2 * N - 2)
        kp = np.clip(k + 1, 0, 2 * N - 2)
        a = m[ar, km].astype(np.float64)
        b = m[ar, k].astype(np.float64)
        c = m[ar, kp].astype(np.float64)
        den = a - 2.0 * b + c
        ok = inner & (den != 0.0)
        frac = np.zeros(W, np.float64)
        frac[ok] = 0.5 * (a[ok] - c[ok]) / den[ok]
        lag_frac[:, q] = frac
    return lag_int, lag_frac, peak


def doppler_phasor(nu_cps: float, n_samples: int) -> np.ndarray:
    """exp(-2j*pi*nu*n), n = 0..N-1, evaluated in float64 and rounded once to complex64 (S8)."""
    return np.exp(-2j * np.pi * float(nu_cps) * np.arange(n_samples)).astype(np.complex64)


def caf_pair(x_i: np.ndarray, x_j: np.ndarray, doppler_cps):
    """S8 (cross-ambiguity over a Doppler grid; a build choice):
    r_d = correlate(x_j * exp(-2j*pi*nu_d*n), x_i); 2-D argmax over (d, lag) in d-major order (ties ->
    lowest d, then lowest lag index); lag interpolated along the lag axis of the winning row.
    Returns (doppler_idx, lag_int, lag_frac, peak)."""
    n = x_i.shape[-1]
    best = None
    for d, nu in enumerate(doppler_cps):
        y = (np.ascontiguousarray(x_j, np.complex64) * doppler_phasor(nu, n)).astype(np.complex64)
        li, lf, pk = xcorr_pair(x_i, y)
        if best is None or pk > best[3]:
            best = (d, li, lf, pk)
    return best


def caf_batch(iq: np.ndarray, doppler_cps, pairs: np.ndarray | None = None):
    iq = np.asarray(iq)
    W, B, N = iq.shape
    if pairs is None:
        pairs = pair_list(B)
    P = pairs.shape[0]
    dop = np.zeros((W, P), np.int32)
    lag_int = np.zeros((W, P), np.int32)
    lag_frac = np.zeros((W, P), np.float64)
    peak = np.zeros((W, P), np.float32)
    for w in range(W):
        for q in range(P):
            i, j = int(pairs[q, 0]), int(pairs[q, 1])
            dop[w, q], lag_int[w, q], lag_frac[w, q], peak[w, q] = caf_pair(iq[w, i], iq[w, j], doppler_cps)
    return dop, lag_int, lag_frac, peak


def peak_margin(x_i: np.ndarray, x_j: np.ndarray) -> float:
    """Relative gap between the largest and the second largest magnitude sample: the integer
    argmax is only meaningfully 'bit-exact' between two float32 FFTs when this is >> 1e-6."""
    m = np.abs(xcorr_full_scipy(x_i, x_j)).astype(np.float64)
    k = int(np.argmax(m))
    top = m[k]
    m[k] = -1.0
    second = m.max()
    return float((top - second) / top) if top > 0 else 0.0


def peak_top2(x_i: np.ndarray, x_j: np.ndarray):
    """(margin, lag of the largest magnitude sample, lag of the second largest): the two candidates between which
    float32 rounding of a different FFT may decide when the margin is below ~1e-6."""
    m = np.abs(xcorr_full_scipy(x_i, x_j)).astype(np.float64)
    n = x_i.shape[-1]
    k = int(np.argmax(m))
    top = m[k]
    m[k] = -1.0
    k2 = int(np.argmax(m))
    margin = float((top - m[k2]) / top) if top > 0 else 0.0
    return margin, k - (n - 1), k2 - (n - 1)


def lag_to_tdoa(lag_samples: float, sample_rate_hz: float):
    """S7: (time_difference_ns:int, distance_difference_m:float), tdoa_processor.py:166-170."""
    t = lag_samples / sample_rate_hz
    return int(round(t * 1e9)), t * SPEED_OF_LIGHT


def decode_u8_iq(raw: np.ndarray) -> np.ndarray:
    """rtl_sdr interleaved uint8 I,Q -> complex64, exactly as buoy_node.py:392-398 /
    iq_stream_client.py:149-157: astype(float32) - 127.5, I + 1j*Q, no scaling."""
    f = np.asarray(raw, dtype=np.uint8).astype(np.float32) - np.float32(127.5)
    return (f[..., 0::2] + 1j * f[..., 1::2]).astype(np.complex64)


def parabola_ulp_bound(x_i, x_j):
    """How far one float32 ulp on each of the three taps moves the interpolated lag, relative to max(|lag|, 1): the part
    of a lag difference that two correct float32 transforms cannot avoid (a flat peak: a - 2b + c small against b)."""
    m = np.abs(xcorr_full_scipy(x_i, x_j)).astype(np.float32)
    n = x_i.shape[-1]
    k = int(np.argmax(m))
    if k == 0 or k == 2 * n - 2:
        return 0.0
    a, b, c = (float(v) for v in m[k - 1:k + 2])
    den = a - 2.0 * b + c
    if den == 0.0:
        return 0.0
    ulp = float(np.spacing(np.float32(b)))
    pa = 0.5 * (1.0 / den - (a - c) / den ** 2)
    pb = (a - c) / den ** 2
    pc = 0.5 * (-1.0 / den - (a - c) / den ** 2)
    lag = (k - (n - 1)) + 0.5 * (a - c) / den
    return (abs(pa) + abs(pb) + abs(pc)) * ulp / max(abs(lag), 1.0)
