#!/usr/bin/env python3
"""GPU box: time k_win at the cfg3 shape under each ablation mask (needs tools/ablate.sh's build).
Results are wrong by construction and, because degenerate data lets the chip hold a higher clock,
every mask overstates what removing its component would save."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RMX_LIBRARY"] = os.path.join(ROOT, "radio-mapper_amd/csrc/librmx_ablate.so")
import torch
from radio_mapper_amd import xcorr


def timing(W=4096, B=8, N=4096, chunk=None, ppb=None, reps=5, resident=1, dbg=0, fused=1):
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    x = torch.randn((W, B, N, 2), device=dev, generator=gen, dtype=torch.float32) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), device=dev, dtype=torch.int32)
    frac = torch.zeros((W, P), device=dev, dtype=torch.float32)
    peak = torch.zeros((W, P), device=dev, dtype=torch.float32)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if chunk: eng.set_option("chunk_windows", chunk)
    if ppb: eng.set_option("pairs_per_block", ppb)
    eng.set_option("timing", 1)
    eng.set_option("resident", resident)
    eng.set_option("dbg", dbg)
    eng.set_option("fused", fused)
    for _ in range(2):
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    tm = eng.last_timing()
    t = min(ts)
    alg = W * P * (16 * N + 12)
    print(f"fused={fused} dbg={dbg} W={W} B={B} res={resident} chunk={chunk} ppb={ppb}: best {t:.3f} ms  med {sorted(ts)[len(ts)//2]:.3f}  "
          f"fwd {tm['fwd_ms']:.3f} ms/{tm['fwd_launches']}  pair {tm['pair_ms']:.3f} ms/{tm['pair_launches']}  "
          f"=> {W*P*N/t/1e6:.1f} Gsamp/s  roofline {alg/t/1e-3/8e12*100:.1f}% of 8 TB/s")
    eng.close()


masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 12, 16, 3, 15, 31]
for m in masks:
    timing(chunk=4096, dbg=m, reps=7)
