"""GPU box: host-side latency of the seam for ONE frequency group (the reference's call pattern): numpy in, measurements out.
usage: python tools/exp_seam_latency.py"""
import sys, time
sys.path.insert(0, '.')
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr, tdoa_processor as tp


def t(fn, n=300):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n * 1e6


for B, N in ((3, 4096), (8, 4096), (8, 16384), (3, 262144)):
    out = rm.synth.make_windows(1, B, N, 2.4e6, seed=5, return_u8=True)
    iq, raw = out[0], out[2]
    with xcorr.XcorrEngine(B, N, 1) as eng:
        a = t(lambda: eng.correlate(iq))
        b = t(lambda: eng.correlate(raw))
    calc = tp.TDoACalculator()
    pos = {f"B{k}": tp.BuoyPosition(f"B{k}", 35.0 + 0.01 * k, -97.0 + 0.01 * (k % 3), 100.0) for k in range(B)}
    dets = [tp.SignalDetection(f"B{k}", 121.5, -50.0, "t", 1000 * k, 35.0, -97.0, 0.9, "beacon", iq[0, k], 2.4e6) for k in range(B)]
    c = t(lambda: calc.calculate_tdoa_measurements(dets, pos), 200)
    dets8 = [tp.SignalDetection(f"B{k}", 121.5, -50.0, "t", 1000 * k, 35.0, -97.0, 0.9, "beacon", raw[0, k], 2.4e6) for k in range(B)]
    d = t(lambda: calc.calculate_tdoa_measurements(dets8, pos), 200)
    print(f"B={B} N={N:6d}: XcorrEngine.correlate host arrays {a:7.1f} us (uint8 {b:7.1f})   calculate_tdoa_measurements {c:7.1f} us (uint8 {d:7.1f})", flush=True)
    calc.close() if hasattr(calc, "close") else None
