#!/usr/bin/env python3
"""GPU box: rmx_caf_batch at a cfg5-like shape (host arrays in/out; the Doppler loop dominates).
usage: bench_caf.py B N W D"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr
xcorr.apply_env_options(report=sys.stderr)   # (a refused knob raises) RMX_<KEY>=<int> of the calling shell -> default options (the library reads no environment)
B, N, W, D = [int(a) for a in sys.argv[1:5]]
fs = 20e6
step = 50.0 / fs
grid = (np.arange(D) - D // 2) * step
rng = np.random.default_rng(5)
offs = rng.integers(-(D // 2) + 1, D // 2, size=B) * step * 0.5
iq, delays = rm.synth.make_windows(W, B, N, fs, seed=1005, doppler_cps=offs)
eng = xcorr.XcorrEngine(B, N, W)
t0 = time.perf_counter(); dop, li, lf, pk = eng.caf(iq, grid); t1 = time.perf_counter()
t2 = time.perf_counter(); dop, li, lf, pk = eng.caf(iq, grid); t3 = time.perf_counter()
P = B * (B - 1) // 2
true = delays[:, xcorr.pair_list(B)[:, 1]] - delays[:, xcorr.pair_list(B)[:, 0]]
ok = np.mean(np.abs(li + lf - true) < 1.0)
print(f"caf B={B} N={N} W={W} D={D}: first {t1-t0:.3f} s, second {t3-t2:.3f} s = {W*P*D*N/(t3-t2)/1e9:.2f} Gsample-bins/s; "
      f"lags within 1 sample of the truth: {ok*100:.1f}%  scratch {eng.scratch_bytes()/2**30:.2f} GiB")
