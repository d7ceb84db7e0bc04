#!/usr/bin/env python3
"""GPU box: rmx_detect_batch throughput (device arrays in and out) beside the CPU oracle (kept under
tests/ because it imports the oracle; not collected by pytest).
usage: python tests/perf_detect.py [N] [W]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # noqa: E702
import numpy as np
import torch
from radio_mapper_amd import xcorr
from oracle import detect_ref as dr
from test_detect import make_windows

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
W = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
x, _ = make_windows(64, N, seed=9)
dev = torch.device("cuda", 0)
xd = torch.from_numpy(np.tile(x.view(np.float32).reshape(64, N, 2), (W // 64, 1, 1))).to(dev)
MP = 2048
cnt = torch.zeros(W, dtype=torch.int32, device=dev); bins = torch.zeros((W, MP), dtype=torch.int32, device=dev)
pw = torch.zeros((W, MP), device=dev); snr = torch.zeros((W, MP), device=dev); cf = torch.zeros((W, MP), device=dev)
fl = torch.zeros(W, device=dev)
eng = xcorr.XcorrEngine(2, 4096, 1)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
lib = xcorr.load_library()
p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
def call():
    assert lib.rmx_detect_batch(eng._ctx, p(xd), W, N, -70.0, int(os.environ.get("DET_DIST", "10")), 10e3 * N / 2.4e6, 0.3, MP, p(cnt), p(bins), p(pw), p(snr),
                                p(cf), p(fl), 3) == 0
call(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
t = sorted(ts)[2]
t0 = time.perf_counter(); dr.detect_batch(x[:32], dc_exclude_bins=10e3 * N / 2.4e6); tc = (time.perf_counter() - t0) / 32
print(f"detect N={N} W={W}: GPU {t:.3f} ms = {W/t*1e3:.3e} windows/s = {W*N/t/1e6:.2f} Gsamples/s "
      f"({W*N*8/t/1e6:.0f} GB/s of input); mean peaks/window {cnt.float().mean().item():.0f}; "
      f"CPU oracle (1 core) {1/tc:.3e} windows/s")
