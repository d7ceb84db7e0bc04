"""CPU oracle of the spectral detection step (TEST INFRASTRUCTURE ONLY).

Literal restatement of the reference's detector (``buoy_node.py:401-433``, the same code in
``iq_stream_client.py:186-217``) with the very library calls it makes -- ``scipy.fft.fft``
(``buoy_node.py:28``), ``np.log10``, ``scipy.signal.find_peaks(height=..., distance=10)``
(``buoy_node.py:411-415``), ``np.median`` (``:425``):

    P[k]   = 20 log10(|FFT_N(iq)[k]| + 1e-12)                  float32, N-point, unpadded, unwindowed
    peaks  = find_peaks(P, height=threshold_db, distance=10)   local maxima (plateau midpoints), P >= height,
                                                               then highest-first removal within < distance bins
    floor  = median(P);  snr = P[peak] - floor;  confidence = min(max(snr / 20, 0), 1)
    report a peak unless |f - f_centre| < 10 kHz (``:419``) or confidence < 0.3 (``:430``)

Pinning.  ``buoy_node.py`` itself cannot be imported in the build container (ModuleNotFoundError: websockets), but
the reference's ``signal_analyzer.py`` holds the same decode (``load_iq_data``, ``:14-41``) and the same dB
spectrum (``analyze_spectrum``, ``:47-86``) and does import: ``tests/golden/make_golden.py --sa-only`` records their
outputs on synthetic captures (``tests/golden/signal_analyzer.npz``), and ``tests/test_oracle_golden.py`` checks
``power_spectrum_db`` (and ``xcorr_ref.decode_u8_iq``) against them.  The peak rule beyond the spectrum --
``find_peaks(height, distance=10)``, the median floor, the confidence cut -- is the reference's literal library
calls (checked statement by statement in ``tests/test_detect.py::test_oracle_is_the_reference_call_sequence``).
"""
from __future__ import annotations

import numpy as np
import scipy.fft
import scipy.signal


def power_spectrum_db(iq: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(iq, np.complex64)
    return 20 * np.log10(np.abs(scipy.fft.fft(x, axis=-1)) + 1e-12)


def detect_one(iq: np.ndarray, threshold_db: float = -70.0, distance: int = 10,
               dc_exclude_bins: float = 0.0, min_confidence: float = 0.3):
    """Returns (bins int32 [n], power_db float32 [n], snr_db float32 [n], confidence float32 [n],
    noise_floor_db float32)."""
    p = power_spectrum_db(iq)
    assert p.dtype == np.float32
    n = p.shape[0]
    peaks, _ = scipy.signal.find_peaks(p, height=threshold_db, distance=distance)
    floor = np.median(p)
    bins, pw, snr, conf = [], [], [], []
    for k in peaks:
        sb = k if k < n // 2 else k - n                      # signed bin of fftfreq
        if abs(sb) < dc_exclude_bins:
            continue
        s = p[k] - floor
        c = min(max(s / 20.0, 0.0), 1.0)
        if c < min_confidence:
            continue
        bins.append(k); pw.append(p[k]); snr.append(s); conf.append(c)
    return (np.array(bins, np.int32), np.array(pw, np.float32), np.array(snr, np.float32),
            np.array(conf, np.float32), np.float32(floor))


def detect_batch(iq: np.ndarray, **kw):
    return [detect_one(iq[w], **kw) for w in range(iq.shape[0])]
