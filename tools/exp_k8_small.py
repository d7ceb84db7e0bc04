#!/usr/bin/env python3
"""GPU box: N = 8192, few windows -- k_win8kl forced (wscr = 2) against the per-transform / four-step kernels (wscr = 0) and
the default dispatch: where is the crossover now?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from radio_mapper_amd import xcorr

def run(B, W, opt, N=8192, reps=15):
    xcorr.clear_default_options()
    for k, v in opt.items():
        xcorr.set_default_option(k, v)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    t0 = time.time()
    while time.time() - t0 < 0.2:
        call(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    eng.close(); xcorr.clear_default_options()
    return sorted(ts)[len(ts) // 2]

for B in (3, 8):
    for W in (8, 16, 32, 48, 64, 96, 128, 192):
        a = run(B, W, {"wscr": 2}); b = run(B, W, {"wscr": 0}); c = run(B, W, {})
        print(f"B={B} W={W:4d}: k_win8kl {a*1e3:7.1f} us   per-transform/four-step {b*1e3:7.1f} us   default {c*1e3:7.1f} us", flush=True)
