#!/usr/bin/env python3
"""GPU box, round 5 (VERDICT r04 #3a): 32-column tiles (256-byte row segments, option col_logt = 5) in the four-step
column kernels against the default 16-column tiles, at the column lengths of cfg2 (L = 2^21) and cfg5 (L = 2^19):
correctness of the new instances against the default path on the same input (integer lags equal, lag / peak to 1e-5),
then per-kernel-family HIP-event times (option timing) and the call time.
usage: exp_cols32.py            -> one table on stdout"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from radio_mapper_amd import xcorr


def run(B, N, W, opts, reps=7, grid=None):
    xcorr.clear_default_options()
    for k, v in opts.items():
        xcorr.set_default_option(k, v)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev)
    peak = torch.zeros((W, P), device=dev)
    dop = torch.zeros((W, P), dtype=torch.int32, device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    if grid is None:
        call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    else:
        call = lambda: eng.caf_device(x.data_ptr(), W, grid, dop.data_ptr(), lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    t0 = time.time()
    while time.time() - t0 < 0.4:
        call()
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        call()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    eng.set_option("timing", 1)
    call()
    fam = eng.last_timing_by_kernel()
    out = (lag.cpu().numpy().copy(), frac.cpu().numpy().copy(), peak.cpu().numpy().copy())
    eng.close()
    xcorr.clear_default_options()
    return sorted(ts)[len(ts) // 2], fam, out


def main():
    shapes = [("cfg2-like 3 x 2^20 x 16", 3, 1 << 20, 16, None),
              ("cfg5 kernels: 8 x 2^18 x 8", 8, 1 << 18, 8, None),
              ("3 x 2^18 x 64", 3, 1 << 18, 64, None)]
    variants = [("default", {}), ("col_logt=5", {"col_logt": 5}), ("logl1=9", {"logl1": 9}),
                ("col_logt=5 logl1=9", {"col_logt": 5, "logl1": 9}), ("col_logt=5 logl1=8", {"col_logt": 5, "logl1": 8})]
    for name, B, N, W, grid in shapes:
        print(f"== {name}", flush=True)
        ref = None
        for vn, o in variants:
            try:
                ms, fam, out = run(B, N, W, o, grid=grid)
            except Exception as e:
                print(f"   {vn:22s} failed: {e}", flush=True)
                continue
            if ref is None:
                ref = out
            same = int(np.sum(out[0] != ref[0]))
            dl = float(np.max(np.abs((out[0] + out[1].astype(np.float64)) - (ref[0] + ref[1].astype(np.float64)))))
            dp = float(np.max(np.abs(out[2] - ref[2]) / np.maximum(ref[2], 1e-30)))
            ks = "  ".join(f"{k} {v['ms'] / max(v['launches'], 1) * 1e3:.0f}us x{v['launches']}" for k, v in fam.items())
            print(f"   {vn:22s} {ms:8.3f} ms | lag_int differ {same}, max |dlag| {dl:.1e}, max rel dpeak {dp:.1e} | {ks}", flush=True)


if __name__ == "__main__":
    main()
