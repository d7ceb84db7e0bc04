#!/usr/bin/env python3
"""GPU box: rmx_solve_batch throughput (device arrays in and out) beside the CPU oracle (kept under
tests/ because it imports the oracle; not collected by pytest).
usage: python tests/perf_solve.py [B] [W]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # noqa: E702
import ctypes as C
import numpy as np
import torch
from radio_mapper_amd import xcorr
from oracle import solve_ref as sr
from test_solve import scenario

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = int(sys.argv[2]) if len(sys.argv) > 2 else 40960
buoys, tx, pairs, li, lf, fs = scenario(B, W, seed=5, noise_m=5.0)
P = len(pairs)
dev = torch.device("cuda", 0)
dli = torch.from_numpy(li).to(dev); dlf = torch.from_numpy(lf).to(dev)
pos = torch.zeros((W, 3), dtype=torch.float64, device=dev); cost = torch.zeros(W, dtype=torch.float64, device=dev)
it = torch.zeros(W, dtype=torch.int32, device=dev)
eng = xcorr.XcorrEngine(B, 4096, 1)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
lib = xcorr.load_library()
bx = np.ascontiguousarray(buoys)
def call():
    rc = lib.rmx_solve_batch(eng._ctx, bx.ctypes.data_as(C.c_void_p), B, None, P, C.c_void_p(dli.data_ptr()),
                             C.c_void_p(dlf.data_ptr()), None, float(fs), W, 60, C.c_void_p(pos.data_ptr()),
                             C.c_void_p(cost.data_ptr()), C.c_void_p(it.data_ptr()), 3)
    assert rc == 0
call(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
t = sorted(ts)[2]
n_cpu = 512
t0 = time.perf_counter(); sr.solve_batch(buoys, pairs, sr.lags_to_dist(li[:n_cpu], lf[:n_cpu], fs)); tc = time.perf_counter() - t0
print(f"solve B={B} P={P} W={W}: GPU {t:.3f} ms = {W/t*1e3:.3e} solves/s (mean {it.float().mean().item():.1f} iterations); "
      f"CPU oracle (1 core, {n_cpu} windows) {n_cpu/tc:.3e} solves/s")
