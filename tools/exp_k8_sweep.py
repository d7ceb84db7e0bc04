#!/usr/bin/env python3
"""GPU box, round 5: N = 8192 through k_win8kl (option kwin8k = 1, the default) against g_win_scr14 (kwin8k = 0), buoys x
windows sweep: ms per call (median of 9), fraction of the 8 TB/s algorithmic roofline, integer lags equal?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr


def run(B, N, W, opt, reps=9):
    xcorr.clear_default_options()
    xcorr.set_default_option("wscr", 2)
    for k, v in opt.items():
        xcorr.set_default_option(k, v)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    src = torch.randn((W, 1, N + 256, 2), device=dev, generator=g) * 30.0
    sh = [int(v) for v in torch.randint(0, 200, (B,), generator=torch.Generator().manual_seed(B))]
    x = torch.stack([src[:, 0, s:s + N] for s in sh], dim=1).contiguous() + torch.randn((W, B, N, 2), device=dev, generator=g) * 10.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev)
    peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    t0 = time.time()
    while time.time() - t0 < 0.3:
        call()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    out = (lag.cpu().numpy().copy(), frac.cpu().numpy().copy(), peak.cpu().numpy().copy())
    eng.close()
    xcorr.clear_default_options()
    return sorted(ts)[len(ts) // 2], out


def main():
    N = 8192
    shapes = [(2, 1024), (3, 1024), (3, 256), (4, 512), (5, 512), (6, 512), (8, 512), (8, 256), (8, 1024), (12, 256), (16, 256), (32, 256)]
    if len(sys.argv) > 1:
        shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
    for B, W in shapes:
        t1, o1 = run(B, N, W, {"kwin8k": 1})
        t0, o0 = run(B, N, W, {"kwin8k": 0})
        alg = W * (B * (B - 1) // 2) * (16 * N + 12)
        same = int(np.sum(o1[0] != o0[0]))
        dl = float(np.max(np.abs(o1[1] - o0[1])))
        print(f"B={B:2d} W={W:4d}: k_win8kl {t1:7.3f} ms ({alg / t1 / 8e9 * 100:4.1f} %)   g_win_scr14 {t0:7.3f} ms ({alg / t0 / 8e9 * 100:4.1f} %)   "
              f"ratio {t1 / t0:.3f} | lag_int differ {same}, max |dfrac| {dl:.1e}", flush=True)


if __name__ == "__main__":
    main()
