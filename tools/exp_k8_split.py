#!/usr/bin/env python3
"""GPU box: where k_win8kl's time goes -- the same B buoys x W windows with pair lists of 1, B-1, P/2, P pairs (custom lists
in the default order): time = forward part + pairs x per-pair cost.   usage: exp_k8_split.py [B W [N]]   (N = 16384: g_win_eo15)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr

B, N, W = 8, 8192, 512
if len(sys.argv) > 2:
    B, W = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    N = int(sys.argv[3])
xcorr.set_default_option("wscr", 2)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
allp = np.array([(i, j) for i in range(B) for j in range(i + 1, B)], np.int32)
eng = xcorr.XcorrEngine(B, N, W)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
res = []
for n in sorted({1, B - 1, max(len(allp) // 2, 1), len(allp)}):
    P = n
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    pr = allp[:n].copy()
    call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr(), pairs=pr)
    t0 = time.time()
    while time.time() - t0 < 0.3:
        call(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[4]
    res.append((n, t))
    print(f"B={B} W={W} pairs {n:3d}: {t:.4f} ms", flush=True)
(n0, t0_), (n1, t1_) = res[0], res[-1]
per_pair = (t1_ - t0_) / max(n1 - n0, 1)
fwd = t0_ - n0 * per_pair
rounds = (W + 255) // 256
print(f"per pair {per_pair * 1e3 / rounds:.2f} us per window (two half transforms), forward part {fwd:.4f} ms = {fwd * 1e3 / rounds / (2 * B):.2f} us per forward half transform")
eng.close()
