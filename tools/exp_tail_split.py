#!/usr/bin/env python3
"""GPU box: the partial last round of the whole-window kernels (k_win8kl at N = 8192; g_win_eo15 at N = 16384, i.e. option
kwin16k = 0 since k16_fwd / k16_pairs became that length's default) through the four-step kernels against the whole-window
kernel for every window (option wscr = 2): us per call, results equal?  At N = 16384 a third column: the default dispatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr

dev = torch.device("cuda", 0)
shapes = [(8192, 8, 300), (8192, 5, 300), (8192, 3, 300), (8192, 16, 290), (8192, 8, 560),
          (16384, 8, 296), (16384, 3, 300), (16384, 5, 330), (16384, 8, 400), (16384, 16, 280)]
for N, B, W in shapes:
    g = torch.Generator(device=dev); g.manual_seed(1)
    src = torch.randn((W, 1, N + 300, 2), device=dev, generator=g) * 30.0
    sh = [int(v) for v in torch.randint(0, 250, (B,), generator=torch.Generator().manual_seed(B))]
    x = torch.stack([src[:, 0, s:s + N] for s in sh], dim=1).contiguous() + torch.randn((W, B, N, 2), device=dev, generator=g) * 10.0
    P = B * (B - 1) // 2
    res, outs = [], []
    for opt in (({"kwin16k": 0}, {"wscr": 2, "kwin16k": 0}, {}) if N == 16384 else ({}, {"wscr": 2})):
        xcorr.clear_default_options()
        for k, v in opt.items():
            xcorr.set_default_option(k, v)
        lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
        fr = torch.zeros((W, P), device=dev); pk = torch.zeros((W, P), device=dev)
        with xcorr.XcorrEngine(B, N, W) as eng:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pk.data_ptr())
            for _ in range(15):
                call()
            torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            eng.set_option("timing", 1)
            call()
            fam = sorted(eng.last_timing_by_kernel())
        res.append((sorted(ts)[4], fam))
        outs.append((lag.cpu().numpy().copy(), fr.cpu().numpy().copy()))
    xcorr.clear_default_options()
    same = int(np.sum(outs[0][0] != outs[1][0]))
    print(f"N={N} B={B:2d} W={W}: split dispatch {res[0][0] * 1e3:8.1f} us ({'split' if len(res[0][1]) > 1 else 'whole'})   whole-window kernel only {res[1][0] * 1e3:8.1f} us   "
          f"lag_int differ {same}, max |dfrac| {np.abs(outs[0][1] - outs[1][1]).max():.1e}"
          + (f"   k16_fwd + k16_pairs (the default) {res[2][0] * 1e3:8.1f} us, lag_int differ {int(np.sum(outs[2][0] != outs[1][0]))}" if len(res) > 2 else ""), flush=True)
