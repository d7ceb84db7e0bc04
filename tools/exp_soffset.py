#!/usr/bin/env python3
"""GPU box experiment: buffer stores of the spectrum scratch with an SGPR soffset.
Runs the fused kernel of a library variant (RMX_LIBRARY) on cfg3-shaped random windows many times and
counts pair-windows whose integer lag differs from the first call of the reference library's result
file (written by the first invocation with --write)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from radio_mapper_amd import xcorr
W, B, N = 4096, 8, 4096
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.randint(0, 256, (W, B, N, 2), device=dev, generator=g, dtype=torch.int32).float() - 127.5
P = B * (B - 1) // 2
lag = torch.zeros((W, P), device=dev, dtype=torch.int32); frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
eng = xcorr.XcorrEngine(B, N, W); eng.set_stream(torch.cuda.current_stream().cuda_stream)
ref_path = "gpurun_out/soffset_ref.npy"
bad_calls = 0; bad_pw = 0; calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ref = None if "--write" in sys.argv else np.load(ref_path)
for k in range(calls):
    eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    li = lag.cpu().numpy()
    if ref is None:
        ref = li.copy(); np.save(ref_path, ref)
    d = int((li != ref).sum())
    bad_calls += d > 0; bad_pw += d
print(f"{os.environ.get('RMX_LIBRARY', 'default')[-14:]}: {calls} calls, {bad_calls} calls with differences, {bad_pw} differing pair-windows of {calls * W * P}", flush=True)
