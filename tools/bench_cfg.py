#!/usr/bin/env python3
"""GPU box: time the hot path on other BASELINE shapes (parity-test cases, not the bench line).
usage: bench_cfg.py B N W [reps]   -> ms per call, samples/s, fraction of the 8 TB/s algorithmic roofline"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from radio_mapper_amd import xcorr
xcorr.apply_env_options(report=sys.stderr)   # (a refused knob raises) RMX_<KEY>=<int> of the calling shell -> default options (the library reads no environment)

def run(B, N, W, reps=9):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    x = torch.randn((W, B, N, 2), device=dev, generator=g) * 30.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    t0 = time.time()                       # warm-up: the device needs ~0.3 s of load to settle on its sustained clock
    while time.time() - t0 < 0.4:
        for _ in range(4):
            eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2]
    alg = W * P * (16 * N + 12)
    print(f"B={B} N={N} W={W}: {t:.3f} ms  {W*P*N/t/1e6:.2f} Gsamples/s  alg {alg/t/1e6:.1f} GB/s = "
          f"{alg/t/1e6/8000*100:.1f}% of 8 TB/s  scratch {eng.scratch_bytes()/2**30:.2f} GiB", flush=True)
    eng.close()

if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    run(a[0], a[1], a[2], a[3] if len(a) > 3 else 9)
