#!/usr/bin/env python3
"""GPU box: the whole-window kernel with scratch spectra (g_win_scr: any buoy count, N = 256 .. 8192) against the
two-kernel LDS path / the four-step path (RMX_WSCR=0) on the same input: complex64 and uint8 input, default plan and
a custom pair list, a window count that leaves the last workgroup partly empty.  Then timings of both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr, synth

def run(iq, pairs, on, N):
    xcorr.set_default_option("wfused", int("0"))
    xcorr.set_default_option("wscr", int("2" if on else "0"))
    W, B = iq.shape[:2]
    with xcorr.XcorrEngine(B, N, W) as eng:
        return eng.correlate(iq, pairs)

def timeit(B, N, W, on, wf="1"):
    xcorr.set_default_option("wfused", int(wf))
    xcorr.set_default_option("wscr", int("2" if on else "0"))
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
logNs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 9, 10, 11, 13]
if "time" not in sys.argv:
  for logN in logNs:
    for B in (2, 3, 5, 8):
        N, W = 1 << logN, 11
        iq, delays, raw = synth.make_windows(W, B, N, 10e6, seed=200 + logN + B, return_u8=True)
        custom = np.array([[B - 1, 0], [0, B - 1], [0, 0], [1, 0], [0, 1], [B - 1, B - 1], [1, 0]], dtype=np.int32)
        for name, data, pairs in (("c64", iq, None), ("u8", raw, None), ("custom", iq, custom)):
            a = run(data, pairs, True, N); b = run(data, pairs, False, N)
            dl = int((a[0] != b[0]).sum())
            df = float(np.abs(a[1] - b[1]).max()); dp = float((np.abs(a[2] - b[2]) / np.abs(b[2])).max())
            ok = dl == 0 and df < 1e-5 and dp < 1e-5
            print(f"N=2^{logN} B={B} {name}: lag mismatches {dl} dfrac {df:.2e} dpeak {dp:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
            bad += not ok
shapes = ((8, 2048, 2048), (8, 1024, 4096), (8, 256, 8192), (3, 8192, 1024), (8, 8192, 512), (16, 2048, 1024), (3, 2048, 4096), (5, 8192, 512))
for B, N, W in shapes:
    if (N.bit_length() - 1) not in logNs and "all" not in sys.argv: continue
    tf, tu = timeit(B, N, W, True, "0"), timeit(B, N, W, False)
    alg = W * (B * (B - 1) // 2) * (16 * N + 12)
    print(f"B={B} N={N} W={W}: g_win_scr {tf:.3f} ms ({alg / tf / 1e6 / 8000 * 100:.1f} % of 8 TB/s)   before {tu:.3f} ms ({alg / tu / 1e6 / 8000 * 100:.1f} %)", flush=True)
sys.exit(1 if bad else 0)
