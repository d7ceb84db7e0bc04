#!/usr/bin/env python3
"""GPU box: N = 8192 (and 4096 / 16384) with raw uint8 I/Q resident in HBM (RMX_IN_U8 | RMX_IN_DEVICE) against complex64:
ms per call, results identical?   usage: exp_k8_u8.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr

def run(B, N, W, reps=9):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(5)
    raw = torch.randint(0, 256, (W, B, N, 2), device=dev, generator=g, dtype=torch.uint8)
    x = raw.to(torch.float32) - 127.5
    P = B * (B - 1) // 2
    outs = []
    res = []
    with xcorr.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for u8, buf in ((False, x), (True, raw)):
            lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
            frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
            call = lambda: eng.correlate_device(buf.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr(), u8=u8)
            t0 = time.time()
            while time.time() - t0 < 0.25:
                call(); torch.cuda.synchronize()
            ts = []
            for _ in range(reps):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); call(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            res.append(sorted(ts)[len(ts) // 2])
            outs.append((lag.cpu().numpy(), frac.cpu().numpy(), peak.cpu().numpy()))
    same = all(np.array_equal(a, b) for a, b in zip(*outs))
    print(f"N={N} B={B} W={W}: complex64 {res[0]:.4f} ms   uint8 {res[1]:.4f} ms   ratio {res[1] / res[0]:.3f}   identical {same}", flush=True)

for B, N, W in ((8, 8192, 512), (3, 8192, 1024), (16, 8192, 256), (8, 4096, 4096), (8, 16384, 256), (3, 16384, 256)):
    run(B, N, W)
