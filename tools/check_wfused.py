#!/usr/bin/env python3
"""GPU box: the whole-window kernel (g_win_fused: <= 4 buoys, N = 256 .. 2048) against the two-kernel LDS path
(RMX_WFUSED=0) on the same input: complex64 and uint8 input, default plan and a custom pair list (reversed, repeated
and auto pairs), a window count that leaves the last workgroup partly empty.  Integer lags must agree exactly,
lag_frac / peak to 1e-5.  Then timings of both on a large batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr, synth

def run(iq, pairs, fused, N):
    xcorr.set_default_option("wfused", int("1" if fused else "0"))
    W, B = iq.shape[:2]
    with xcorr.XcorrEngine(B, N, W) as eng:
        return eng.correlate(iq, pairs)

def timeit(B, N, W, fused):
    xcorr.set_default_option("wfused", int("1" if fused else "0"))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    eng.close()
    return sorted(ts)[len(ts) // 2]

bad = 0
for logN in (8, 9, 10, 11):
    for B in (2, 3, 4):
        N, W = 1 << logN, 37
        iq, delays, raw = synth.make_windows(W, B, N, 10e6, seed=100 + logN + B, return_u8=True)
        custom = np.array([[B - 1, 0], [0, B - 1], [0, 0], [1, 0], [0, 1], [B - 1, B - 1], [1, 0]], dtype=np.int32)
        for name, data, pairs in (("c64", iq, None), ("u8", raw, None), ("custom", iq, custom)):
            a = run(data, pairs, True, N); b = run(data, pairs, False, N)
            dl = int((a[0] != b[0]).sum())
            df = float(np.abs(a[1] - b[1]).max()); dp = float((np.abs(a[2] - b[2]) / np.abs(b[2])).max())
            ok = dl == 0 and df < 1e-5 and dp < 1e-5
            print(f"N=2^{logN} B={B} {name}: lag mismatches {dl} dfrac {df:.2e} dpeak {dp:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
for B, N, W in ((3, 2048, 4096), (3, 1024, 8192), (3, 512, 16384), (3, 256, 16384), (4, 2048, 4096), (2, 2048, 4096)):
    tf, tu = timeit(B, N, W, True), timeit(B, N, W, False)
    alg = W * (B * (B - 1) // 2) * (16 * N + 12)
    print(f"B={B} N={N} W={W}: fused {tf:.3f} ms ({alg / tf / 1e6 / 8000 * 100:.1f} % of 8 TB/s)   two kernels {tu:.3f} ms ({alg / tu / 1e6 / 8000 * 100:.1f} %)", flush=True)
sys.exit(1 if bad else 0)
