#!/usr/bin/env python3
"""Generates the golden fixtures in this directory FROM THE REFERENCE MODULE ITSELF.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/tdoa_processor.py`` and records

* ``xcorr_*.npz``  -- outputs of ``tdoa_processor.correlate`` (the module's only xcorr symbol,
  ``tdoa_processor.py:20``; it *is* ``scipy.signal.correlate``) called as
  ``correlate(x_j, x_i, mode='full', method='fft')`` for every pair i<j in the reference's loop
  order (``tdoa_processor.py:156-157``), reduced by the path's spec (SURVEY.md §8a-spec S4-S6:
  ``np.abs`` -> ``np.argmax`` -> 3-point parabola).  Small cases store their inputs (as the raw
  uint8 I/Q that decodes to them); large cases store the generator seed + an input checksum.
  ``--r04-only`` writes just the fixture added in round 4 (``xcorr_b3_n8192``: the capture length of
  ``iq_stream_client.py:459``).
* ``caf_*.npz`` -- the same primitive over a Doppler grid (``run_caf_case``); ``--caf-only``
  regenerates just these.
* ``triangulate_position.json`` -- ``HyperbolicPositioning.triangulate_position`` on deterministic
  scenarios (``make_triangulation``); ``--tri-only`` regenerates just this.
* ``tdoa_conventions.json`` -- ``TDoACalculator.calculate_tdoa_measurements`` on the hand-built
  detections of the reference's own ``main()`` example (``tdoa_processor.py:475-490``): pins the
  pair order, the sign (buoy2 - buoy1) and the ns -> metres conversion.

* ``signal_analyzer.npz`` -- the reference's own uint8 decode and dB spectrum: ``load_iq_data`` (``signal_analyzer.py:14-41``,
  the same three statements as ``buoy_node.py:392-398``) and ``analyze_spectrum`` (``:47-86``: ``np.fft.fft`` ->
  ``fftshift`` -> ``20 log10(|X| + 1e-12)`` -> ``scipy.signal.find_peaks``) of the imported ``/root/reference/signal_analyzer.py``
  on synthetic rtl_sdr captures written to a temporary ``.bin`` (``make_signal_analyzer``); ``--sa-only`` regenerates just this.

Only arrays / JSON are written: no reference source text.
"""
import hashlib
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import radio_mapper_amd as rm  # noqa: E402  (synthetic inputs only)


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_tdoa_processor",
                                                  "/root/reference/tdoa_processor.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reduce_full(c, n):
    m = np.abs(c)
    k = int(np.argmax(m))
    if 0 < k < m.shape[0] - 1:
        a, b, cc = float(m[k - 1]), float(m[k]), float(m[k + 1])
        den = a - 2.0 * b + cc
        frac = 0.0 if den == 0.0 else 0.5 * (a - cc) / den
    else:
        frac = 0.0
    mm = m.astype(np.float64).copy()
    top = mm[k]
    mm[k] = -1.0
    margin = (top - mm.max()) / top if top > 0 else 0.0
    taps = [float(m[max(k - 1, 0)]), float(m[k]), float(m[min(k + 1, m.shape[0] - 1)])]
    return k - (n - 1), frac, float(m[k]), margin, taps


def run_case(ref, iq):
    W, B, N = iq.shape
    pairs = [(i, j) for i in range(B) for j in range(i + 1, B)]
    P = len(pairs)
    lag_int = np.zeros((W, P), np.int32)
    lag_frac = np.zeros((W, P), np.float64)
    peak = np.zeros((W, P), np.float32)
    margin = np.zeros((W, P), np.float64)
    taps = np.zeros((W, P, 3), np.float32)
    for w in range(W):
        for q, (i, j) in enumerate(pairs):
            c = ref.correlate(iq[w, j], iq[w, i], mode="full", method="fft")
            assert c.dtype == np.complex64 and c.shape[0] == 2 * N - 1
            lag_int[w, q], lag_frac[w, q], peak[w, q], margin[w, q], taps[w, q] = reduce_full(c, N)
    return dict(pairs=np.array(pairs, np.int32), lag_int=lag_int, lag_frac=lag_frac, peak=peak,
                margin=margin, taps=taps)


def run_caf_case(ref, iq, doppler_cps):
    """Cross-ambiguity fixture: per pair and Doppler hypothesis d the
    reference primitive is called on the de-rotated later window,
    ``correlate((x_j * exp(-2j*pi*nu_d*n)).astype(complex64), x_i, 'full', 'fft')``; the winning
    (d, lag) is the d-major first maximum of |c|."""
    W, B, N = iq.shape
    pairs = [(i, j) for i in range(B) for j in range(i + 1, B)]
    P, D = len(pairs), len(doppler_cps)
    n = np.arange(N)
    rot = [np.exp(-2j * np.pi * nu * n).astype(np.complex64) for nu in doppler_cps]
    out = dict(pairs=np.array(pairs, np.int32), doppler_cps=np.asarray(doppler_cps, np.float64),
               dop_idx=np.zeros((W, P), np.int32), lag_int=np.zeros((W, P), np.int32),
               lag_frac=np.zeros((W, P), np.float64), peak=np.zeros((W, P), np.float32),
               bin_peak=np.zeros((W, P, D), np.float32), bin_lag=np.zeros((W, P, D), np.int32))
    for w in range(W):
        for q, (i, j) in enumerate(pairs):
            best = None
            for d in range(D):
                y = (iq[w, j] * rot[d]).astype(np.complex64)
                c = ref.correlate(y, iq[w, i], mode="full", method="fft")
                li, fr, pk, _, _ = reduce_full(c, N)
                out["bin_peak"][w, q, d] = pk
                out["bin_lag"][w, q, d] = li
                if best is None or np.float32(pk) > np.float32(best[3]):
                    best = (d, li, fr, pk)
            out["dop_idx"][w, q], out["lag_int"][w, q], out["lag_frac"][w, q], out["peak"][w, q] = best
    return out


def make_caf(ref):
    cases = [
        ("caf_b3_n4096", dict(n_windows=2, n_buoys=3, n_samples=4096, sample_rate_hz=2.4e6, seed=21), 9, 0.5,
         [0, 1, -2]),
        ("caf_b3_n1024", dict(n_windows=2, n_buoys=3, n_samples=1024, sample_rate_hz=2.4e6, seed=22), 7, 0.5,
         [1, -1, 2]),
        # BASELINE configs[4]'s grid: 21 hypotheses (+-500 Hz step 50 Hz at 20 MS/s = +-10 steps), small window;
        # the step is 0.25 bin of this window, so neighbouring hypotheses overlap as they do at N = 2^18
        # (50 Hz = 0.66 bin of 76 Hz): the winner is decided by the peak heights, not by orthogonality
        ("caf_b4_n2048_d21", dict(n_windows=1, n_buoys=4, n_samples=2048, sample_rate_hz=20e6, seed=23), 21, 0.25,
         [0, 5, -4, 2]),
    ]
    for name, kw, nd, step_bins, offs in cases:
        step = step_bins / kw["n_samples"]                     # cycles/sample
        grid = (np.arange(nd) - nd // 2) * step
        iq, delays, raw = rm.synth.make_windows(return_u8=True, doppler_cps=np.array(offs) * step, **kw)
        res = run_caf_case(ref, iq, grid)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), raw_u8=raw, delays=delays,
                            buoy_doppler_cps=np.array(offs) * step, sample_rate_hz=kw["sample_rate_hz"], **res)
        print(name, "dop", res["dop_idx"].tolist(), "lag", res["lag_int"].tolist())


TRI_SCENARIOS = {
    # name: (buoys (id, lat, lng, altitude, timing_accuracy_ns), transmitter (lat, lng, alt), range noise sigma [m])
    "square4": ([("A", 35.40, -97.60, 0.0, 50000), ("B", 35.40, -97.40, 0.0, 50000), ("C", 35.55, -97.40, 0.0, 50000),
                 ("D", 35.55, -97.60, 0.0, 50000)], (35.47, -97.52, 0.0), 0.0),
    "penta5": ([("A", 51.50, -0.20, 0.0, 50000), ("B", 51.42, -0.05, 10.0, 50000), ("C", 51.50, 0.10, 0.0, 50000),
                ("D", 51.60, 0.02, 5.0, 50000), ("E", 51.60, -0.15, 0.0, 50000)], (51.52, -0.06, 0.0), 0.0),
    "tri3": ([("A", 51.505, -0.09, 0.0, 50000), ("B", 51.51, -0.1, 0.0, 75000), ("C", 51.5, -0.12, 0.0, 60000)],
             (51.507, -0.1, 0.0), 0.0),
    "tri3_noisy": ([("A", 51.505, -0.09, 0.0, 50000), ("B", 51.51, -0.1, 0.0, 75000), ("C", 51.5, -0.12, 0.0, 60000)],
                   (51.507, -0.1, 0.0), 5.0),
    "hex6alt": ([("A", 10.0, 20.0, 0.0, 1000), ("B", 10.05, 20.08, 50.0, 1000), ("C", 10.12, 20.08, 0.0, 1000),
                 ("D", 10.17, 20.0, 30.0, 1000), ("E", 10.12, 19.92, 0.0, 1000), ("F", 10.05, 19.92, 80.0, 1000)],
                (10.09, 20.01, 0.0), 0.0),
    "penta5_noisy": ([("A", 51.50, -0.20, 0.0, 50000), ("B", 51.42, -0.05, 10.0, 50000), ("C", 51.50, 0.10, 0.0, 50000),
                      ("D", 51.60, 0.02, 5.0, 50000), ("E", 51.60, -0.15, 0.0, 50000)], (51.52, -0.06, 0.0), 5.0),
}


def make_triangulation(ref):
    """``HyperbolicPositioning.triangulate_position`` (tdoa_processor.py:218-328) of the imported reference
    on deterministic scenarios: exact (or seeded-noisy) distance differences of a known transmitter for
    every pair i<j.  Records the reference's result (or null where its BFGS reports failure) -- the
    consumer row a8 and the cost bar of the batched GPU solve are pinned to these numbers."""
    G = ref.GeodeticCalculator
    out = {}
    for name, (buoys, tx, sigma) in TRI_SCENARIOS.items():
        rng = np.random.default_rng(3)
        pos = {b[0]: ref.BuoyPosition(*b) for b in buoys}
        xyz = {b[0]: np.array(G.lat_lng_to_xyz(b[1], b[2], b[3])) for b in buoys}
        t = np.array(G.lat_lng_to_xyz(*tx))
        meas = []
        ids = [b[0] for b in buoys]
        for i in range(len(ids)):
            for j in range(i + 1, len(ids)):
                dd = float(np.linalg.norm(t - xyz[ids[j]]) - np.linalg.norm(t - xyz[ids[i]]))
                if sigma:
                    dd += float(rng.normal(0.0, sigma))
                meas.append([ids[i], ids[j], int(round(dd / 299792458.0 * 1e9)), dd, 0.8, 121.5])
        res = ref.HyperbolicPositioning().triangulate_position([ref.TDoAMeasurement(*m) for m in meas], pos)
        out[name] = dict(buoys=[list(b) for b in buoys], transmitter=list(tx), sigma_m=sigma, measurements=meas,
                         result=None if res is None else dict(
                             estimated_lat=res.estimated_lat, estimated_lng=res.estimated_lng,
                             estimated_altitude=res.estimated_altitude, accuracy_meters=res.accuracy_meters,
                             confidence=res.confidence, frequency_mhz=res.frequency_mhz, method=res.method,
                             contributing_buoys=sorted(res.contributing_buoys),
                             cost=res.accuracy_meters ** 2 * len(meas)))
        print(name, None if res is None else (res.estimated_lat, res.estimated_lng, res.accuracy_meters))
    import scipy
    with open(os.path.join(HERE, "triangulate_position.json"), "w") as f:
        json.dump(dict(meta=dict(numpy=np.__version__, scipy=scipy.__version__,
                                 source="HyperbolicPositioning.triangulate_position of /root/reference/tdoa_processor.py"),
                       scenarios=out), f, indent=1)


def make_signal_analyzer():
    """Decode + dB spectrum fixture from the reference's signal_analyzer.py (a5 / a6 of SURVEY.md section 8a)."""
    import contextlib
    import io
    import tempfile
    os.environ.setdefault("MPLBACKEND", "Agg")
    spec = importlib.util.spec_from_file_location("ref_signal_analyzer", "/root/reference/signal_analyzer.py")
    sa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sa)
    rng = np.random.default_rng(4242)
    fs, fc_mhz = 2.048e6, 121.5                      # buoy_node.py:363 sample rate; an aviation-band centre
    out = {}
    raws, decs, specs, freqs, peaks = [], [], [], [], []
    for n in (4096, 16384):                          # 16384 = the capture length of buoy_node.py:364
        t = np.arange(n)
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 5.0
        for f, a in ((0.11, 55.0), (-0.23, 30.0), (0.37, 18.0)):
            x += a * np.exp(2j * np.pi * (f * t + rng.uniform()))
        raw = np.empty(2 * n, np.uint8)
        raw[0::2] = np.clip(np.floor(x.real + 128.0), 0, 255)
        raw[1::2] = np.clip(np.floor(x.imag + 128.0), 0, 255)
        # every byte value once more at the end of the larger capture: the decode is pinned on all 256 of them
        if n == 16384:
            raw[-512:-256] = np.arange(256, dtype=np.uint8)
            raw[-256:] = np.arange(255, -1, -1, dtype=np.uint8)
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "iq_capture_121.5MHz_test.bin")   # the naming of sdr_capture.py:24-45
            raw.tofile(path)
            with contextlib.redirect_stdout(io.StringIO()):
                dec, fs_out = sa.load_iq_data(path, sample_rate=int(fs))
                fr, ps, pk = sa.analyze_spectrum(dec, fs_out, fc_mhz)
        assert fs_out == int(fs)
        key = "n%d" % n
        out[key + "_raw_u8"] = raw
        out[key + "_decoded"] = dec
        out[key + "_power_spectrum_db"] = ps          # fftshift order, as the reference returns it
        bins = np.searchsorted(fr, pk)                 # the reference returns peak FREQUENCIES; keep their shifted bins
        assert np.array_equal(fr[bins], pk)
        out[key + "_freq_first_last_mhz"] = np.array([fr[0], fr[-1]])
        out[key + "_peak_bins_shifted"] = bins.astype(np.int32)
        print("signal_analyzer n=%d: decoded %s, spectrum %s, %d peaks" % (n, dec.dtype, ps.dtype, len(pk)))
    out["sample_rate_hz"] = np.float64(fs)
    out["center_freq_mhz"] = np.float64(fc_mhz)
    np.savez_compressed(os.path.join(HERE, "signal_analyzer.npz"), **out)


def checksum(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if "--sa-only" in sys.argv:
        make_signal_analyzer()
        return
    ref = load_reference()
    if "--caf-only" in sys.argv:
        make_caf(ref)
        return
    if "--tri-only" in sys.argv:
        make_triangulation(ref)
        return
    import scipy
    meta = dict(numpy=np.__version__, scipy=scipy.__version__,
                primitive="tdoa_processor.correlate(x_j, x_i, mode='full', method='fft')")
    if "--r04-only" in sys.argv:
        # round 4: the streaming client's capture length (iq_stream_client.py:459: 8192 samples), which had no fixture of
        # its own; written without touching the files of the earlier rounds
        for name, kw in [("xcorr_b3_n8192", dict(n_windows=2, n_buoys=3, n_samples=8192, sample_rate_hz=2.4e6, seed=15))]:
            iq, delays, raw = rm.synth.make_windows(return_u8=True, **kw)
            res = run_case(ref, iq)
            np.savez_compressed(os.path.join(HERE, name + ".npz"), raw_u8=raw, delays=delays,
                                sample_rate_hz=kw["sample_rate_hz"], **res)
            print(name, "min margin %.3e" % res["margin"].min())
        return

    small = [
        ("xcorr_b3_n1024", dict(n_windows=2, n_buoys=3, n_samples=1024, sample_rate_hz=2.4e6, seed=11)),
        ("xcorr_b3_n4096", dict(n_windows=1, n_buoys=3, n_samples=4096, sample_rate_hz=2.4e6, seed=12)),
        ("xcorr_b8_n4096", dict(n_windows=4, n_buoys=8, n_samples=4096, sample_rate_hz=10e6, seed=1003)),
        ("xcorr_b4_n256", dict(n_windows=3, n_buoys=4, n_samples=256, sample_rate_hz=2.048e6, seed=13)),
        ("xcorr_b3_n16384", dict(n_windows=1, n_buoys=3, n_samples=16384, sample_rate_hz=2.048e6, seed=14)),
        ("xcorr_b3_n8192", dict(n_windows=2, n_buoys=3, n_samples=8192, sample_rate_hz=2.4e6, seed=15)),   # (round 4)
    ]
    for name, kw in small:
        iq, delays, raw = rm.synth.make_windows(return_u8=True, **kw)
        res = run_case(ref, iq)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), raw_u8=raw, delays=delays,
                            sample_rate_hz=kw["sample_rate_hz"], **res)
        print(name, "min margin %.3e" % res["margin"].min())

    # edge cases with stored complex inputs (ties, zeros, impulses, edge peaks)
    N = 256
    e = np.zeros((6, 2, N), np.complex64)
    # w0: all zeros -> every magnitude ties at 0 -> argmax = index 0 -> lag -(N-1), frac 0
    # w1: impulses: x_i = delta[5], x_j = delta[25] -> lag +20
    e[1, 0, 5] = 1.0; e[1, 1, 25] = 2.0 - 1.0j
    # w2: peak at the most positive lag N-1 (edge, frac 0): x_i = delta[0], x_j = delta[N-1]
    e[2, 0, 0] = 3.0; e[2, 1, N - 1] = 1.0j
    # w3: peak at the most negative lag -(N-1)
    e[3, 0, N - 1] = 1.0; e[3, 1, 0] = -2.0
    # w4: two equal peaks (lags -7 and +9): ties -> most negative lag
    e[4, 0, 100] = 1.0; e[4, 1, 93] = 1.0; e[4, 1, 109] = 1.0
    # w5: constant inputs -> triangular magnitude, peak at lag 0
    e[5, 0, :] = 4.5 - 2.5j; e[5, 1, :] = -1.5 + 0.5j
    res = run_case(ref, e)
    np.savez_compressed(os.path.join(HERE, "xcorr_edge_n256.npz"), iq=e, **res)
    print("edge lags", res["lag_int"].ravel(), res["lag_frac"].ravel())

    make_caf(ref)
    make_triangulation(ref)

    # large, seed-regenerated cases (inputs not stored)
    large = [
        ("xcorr_cfg1_n262144", dict(n_windows=1, n_buoys=3, n_samples=262144, sample_rate_hz=2.4e6, seed=1001)),
        ("xcorr_b3_n1048576", dict(n_windows=1, n_buoys=3, n_samples=1048576, sample_rate_hz=2.4e6, seed=1002)),
    ]
    for name, kw in large:
        iq, delays = rm.synth.make_windows(**kw)
        res = run_case(ref, iq)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), delays=delays,
                            gen=json.dumps(kw), input_sha256=checksum(iq), **res)
        print(name, "min margin %.3e" % res["margin"].min(), res["lag_int"].ravel())

    # conventions of the pair loop (tdoa_processor.py:146-198) on main()'s example
    proc = ref.TDoAProcessor()
    buoys = [("BUOY_ALPHA", 51.505, -0.09, 0.0, 50000), ("BUOY_BETA", 51.51, -0.1, 0.0, 75000),
             ("BUOY_GAMMA", 51.5, -0.12, 0.0, 60000)]
    for b in buoys:
        proc.register_buoy(ref.BuoyPosition(*b))
    base = 1_700_000_000_000_000_000
    dets = [("BUOY_ALPHA", 121.5, -55, "2025-01-18T16:30:00Z", base, 51.505, -0.09, 0.9, "emergency"),
            ("BUOY_BETA", 121.5, -60, "2025-01-18T16:30:00Z", base + 150000, 51.51, -0.1, 0.85, "emergency"),
            ("BUOY_GAMMA", 121.5, -58, "2025-01-18T16:30:00Z", base + 300000, 51.5, -0.12, 0.88, "emergency")]
    meas = proc.tdoa_calculator.calculate_tdoa_measurements(
        [ref.SignalDetection(*d) for d in dets], proc.buoy_positions)
    conv = dict(meta=meta, buoys=buoys, detections=dets,
                measurements=[dict(buoy1_id=m.buoy1_id, buoy2_id=m.buoy2_id,
                                   time_difference_ns=m.time_difference_ns,
                                   distance_difference_m=m.distance_difference_m,
                                   confidence=m.confidence, frequency_mhz=m.frequency_mhz)
                              for m in meas],
                network_status=proc.get_buoy_network_status(),
                freq_groups={str(k): [d.buoy_id for d in v] for k, v in proc._group_by_frequency(
                    [ref.SignalDetection(*d) for d in dets] +
                    [ref.SignalDetection("BUOY_ALPHA", 121.505, -50, "t", base, 0, 0, 0.5),
                     ref.SignalDetection("BUOY_BETA", 156.8, -50, "t", base, 0, 0, 0.5)]).items()},
                time_window=[d.buoy_id for d in proc._filter_by_time_window(
                    [ref.SignalDetection("A", 1.0, 0, "t", base, 0, 0, 1.0),
                     ref.SignalDetection("B", 1.0, 0, "t", base - 9_000_000_000, 0, 0, 1.0),
                     ref.SignalDetection("C", 1.0, 0, "t", base - 11_000_000_000, 0, 0, 1.0)])])
    with open(os.path.join(HERE, "tdoa_conventions.json"), "w") as f:
        json.dump(conv, f, indent=1)
    print("conventions:", [(m["buoy1_id"], m["buoy2_id"], m["time_difference_ns"]) for m in conv["measurements"]])


if __name__ == "__main__":
    main()
