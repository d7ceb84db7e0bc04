#!/usr/bin/env python3
"""GPU box: N = 16384 -- k16_fwd / k16_pairs (kwin16k = 2) against g_win_eo15 (kwin16k = 0, wscr = 2), the four-step kernels
(kwin16k = 0, wscr = 0) and round 5's earlier dispatch (kwin16k = 0) over buoy counts and batch sizes, complex64 and raw
uint8 input.  Medians of 15 calls behind a 0.2 s warm-up, inputs resident."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from radio_mapper_amd import xcorr

N = 16384

def run(B, W, opt, u8=False, reps=15):
    xcorr.clear_default_options()
    for k, v in opt.items():
        xcorr.set_default_option(k, v)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    if u8:
        x = torch.randint(0, 256, (W, B, N, 2), device=dev, generator=g, dtype=torch.uint8)
    else:
        x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr(), u8=u8)
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

if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128, 192, 256, 300, 512, 1024]
    for B in (2, 3, 5, 8, 16):
        P = B * (B - 1) // 2
        for W in sizes:
            if W * B * N * 8 > 6e9:
                continue
            k = run(B, W, {"kwin16k": 2}); k8 = run(B, W, {"kwin16k": 2}, u8=True)
            e = run(B, W, {"kwin16k": 0, "wscr": 2}); f = run(B, W, {"kwin16k": 0, "wscr": 0}); d = run(B, W, {"kwin16k": 0})
            frac = W * P * 16.0 * N / (k * 1e-3) / 8e12
            print(f"B={B:2d} W={W:5d}: k16 {k*1e3:8.1f} us ({100*frac:4.1f} %)  k16 uint8 {k8*1e3:8.1f}   g_win_eo15 {e*1e3:8.1f}   four-step {f*1e3:8.1f}   "
                  f"earlier dispatch {d*1e3:8.1f}   best other / k16 {min(e, f, d) / k:5.2f}", flush=True)
