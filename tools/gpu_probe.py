#!/usr/bin/env python3
"""Scratch probe (GPU box): parity of the HIP path on a fixture + raw timing at the cfg3 shape."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr
from oracle import xcorr_ref as orc

def parity():
    g = np.load(os.path.join(ROOT, "tests/golden/xcorr_b8_n4096.npz"))
    iq = orc.decode_u8_iq(g["raw_u8"])
    eng = xcorr.XcorrEngine(8, 4096, 64)
    li, lf, pk = eng.correlate(iq)
    ok = np.array_equal(li, g["lag_int"])
    ref = g["lag_int"] + g["lag_frac"]
    err = np.abs((li + lf.astype(np.float64)) - ref) / np.maximum(np.abs(ref), 1.0)
    print("parity b8_n4096: lag_int exact:", ok, " max frac rel err %.3e" % err.max(),
          " peak rel err %.3e" % (np.abs(pk - g["peak"]) / g["peak"]).max())
    if not ok:
        bad = np.argwhere(li != g["lag_int"])
        print("mismatches", len(bad), bad[:10], li[li != g["lag_int"]][:10], g["lag_int"][li != g["lag_int"]][:10])
    # u8 path
    li2, lf2, pk2 = eng.correlate(g["raw_u8"])
    print("u8 path identical:", np.array_equal(li, li2), np.array_equal(lf, lf2), np.array_equal(pk, pk2))
    eng.close()
    return ok

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

if __name__ == "__main__":
    print("lib:", xcorr.library_path(), "devices:", xcorr.device_count())
    parity()
    timing(chunk=4096)
    timing(chunk=512)
