"""Synthetic IQ capture windows of the shapes BASELINE.md names.

Signal model (BASELINE.md §2 / SURVEY.md §8d): one common band-limited (0.8*fs) complex Gaussian
source; per-buoy true delay applied as a frequency-domain phase ramp (integer + fractional
samples), |delay| <= 50 km / c (``config.yaml:145`` maximum_baseline_km); independent AWGN at
10 dB SNR (``config.yaml:149``); amplitudes in the rtl_sdr decode range +-127.5
(``buoy_node.py:392-398``: uint8 - 127.5, no scaling), optionally quantised to that uint8 grid so
that the very same data can be fed as raw uint8 I/Q.

numpy only; deterministic for a given seed (``np.random.default_rng``).
"""
from __future__ import annotations

import numpy as np

SPEED_OF_LIGHT = 299792458.0
MAX_BASELINE_M = 50e3  # config.yaml:145


def max_delay_samples(sample_rate_hz: float) -> float:
    return MAX_BASELINE_M / SPEED_OF_LIGHT * sample_rate_hz


def _next_pow2(n: int) -> int:
    p = 1
    while p < n:
        p *= 2
    return p


def make_windows(n_windows: int, n_buoys: int, n_samples: int, sample_rate_hz: float,
                 seed: int, snr_db: float = 10.0, bandwidth: float = 0.8,
                 rms: float = 32.0, quantise: bool = True, max_delay: float | None = None,
                 return_u8: bool = False, doppler_cps: np.ndarray | None = None,
                 delays: np.ndarray | None = None):
    """Returns (iq complex64 [W][B][N], delays float64 [W][B]) and, with return_u8, also the raw
    interleaved uint8 [W][B][2N] that decodes (u8 - 127.5) to exactly ``iq``.

    True lag of pair (i, j) is delays[:, j] - delays[:, i] (buoy2 - buoy1, tdoa_processor.py:51).
    delays: optional [W][B] true delays in samples (default: uniform in +-max_delay).
    doppler_cps: optional [W][B] (or [B]) frequency offsets in cycles/sample applied per buoy as
    exp(+2j*pi*nu*n) after the delay (SURVEY.md section 8a-spec S8 / BASELINE cfg5).
    """
    rng = np.random.default_rng(seed)
    W, B, N = n_windows, n_buoys, n_samples
    D = max_delay_samples(sample_rate_hz) if max_delay is None else float(max_delay)
    D = min(D, (N - 2) / 2.0)  # keep every pair lag inside the 'full' range
    margin = int(np.ceil(D)) + 2
    Ns = _next_pow2(N + 2 * margin)
    freqs = np.fft.fftfreq(Ns)  # cycles/sample
    mask = (np.abs(freqs) <= bandwidth / 2.0)
    out = np.empty((W, B, N), np.complex64)
    if delays is None:
        delays = rng.uniform(-D, D, size=(W, B))
    else:   # caller-supplied true delays in samples, e.g. from a transmitter/buoy geometry
        delays = np.asarray(delays, np.float64).reshape(W, B)
        assert np.abs(delays).max() <= margin - 1, "delays exceed the generator's margin"
    sig_amp = 1.0
    noise_amp = 10.0 ** (-snr_db / 20.0)
    scale = rms / np.sqrt(sig_amp ** 2 + noise_amp ** 2)
    chunk = max(1, min(W, (1 << 22) // (Ns * B)))
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        c = w1 - w0
        S = (rng.standard_normal((c, Ns)) + 1j * rng.standard_normal((c, Ns))) * mask
        # unit average power in the time domain
        S *= np.sqrt(Ns / (2.0 * mask.sum())) * np.sqrt(Ns)
        ramp = np.exp(-2j * np.pi * freqs[None, None, :] * delays[w0:w1, :, None])
        s = np.fft.ifft(S[:, None, :] * ramp, axis=-1)[:, :, margin:margin + N]
        if doppler_cps is not None:
            nu = np.broadcast_to(np.asarray(doppler_cps, np.float64), (W, B))[w0:w1]
            s = s * np.exp(2j * np.pi * nu[:, :, None] * np.arange(N)[None, None, :])
        noise = (rng.standard_normal((c, B, N)) + 1j * rng.standard_normal((c, B, N))) \
            * (noise_amp / np.sqrt(2.0))
        x = (s * sig_amp + noise) * scale
        out[w0:w1] = x.astype(np.complex64)
    if quantise or return_u8:
        re = np.clip(np.floor(out.real + 128.0), 0, 255).astype(np.uint8)
        im = np.clip(np.floor(out.imag + 128.0), 0, 255).astype(np.uint8)
        out = ((re.astype(np.float32) - np.float32(127.5))
               + 1j * (im.astype(np.float32) - np.float32(127.5))).astype(np.complex64)
        if return_u8:
            raw = np.empty((W, B, 2 * N), np.uint8)
            raw[..., 0::2] = re
            raw[..., 1::2] = im
            return out, delays, raw
    else:
        np.clip(out.real, -127.5, 127.5, out=out.real)
        np.clip(out.imag, -127.5, 127.5, out=out.imag)
    return out, delays


# The BASELINE.json configs as concrete synthetic shapes.
CONFIGS = {
    "cfg1": dict(n_buoys=3, sample_rate_hz=2.4e6, n_samples=262144, n_windows=1, seed=1001),
    "cfg2": dict(n_buoys=3, sample_rate_hz=2.4e6, n_samples=1048576, n_windows=64, seed=1002),
    "cfg3": dict(n_buoys=8, sample_rate_hz=10e6, n_samples=4096, n_windows=4096, seed=1003),
    "cfg4": dict(n_buoys=16, sample_rate_hz=10e6, n_samples=4096, n_windows=4096, n_channels=10,
                 seed=1004),
    "cfg5": dict(n_buoys=32, sample_rate_hz=20e6, n_samples=262144, n_windows=64, seed=1005,
                 doppler_hz=500.0, doppler_step_hz=50.0),
}
