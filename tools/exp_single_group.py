"""GPU box: device time of ONE frequency group (one window) per window length and buoy count -- the reference's call pattern.
usage: python tools/exp_single_group.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr

print("us per call (device pointers), one window:  B=3     B=8    B=16")
for logn in range(8, 19):
    N = 1 << logn
    row = []
    for B in (3, 8, 16):
        iq, _ = rm.synth.make_windows(1, B, N, 2.4e6, seed=logn)
        x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
        P = B * (B - 1) // 2
        o = [torch.zeros((1, P), dtype=torch.int32, device='cuda')] + [torch.zeros((1, P), device='cuda') for _ in range(2)]
        with xcorr.XcorrEngine(B, N, 1) as eng:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            call = lambda: eng.correlate_device(x.data_ptr(), 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
            for _ in range(20): call()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): call()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"N = {N:7d}                              {row[0]:7.1f} {row[1]:7.1f} {row[2]:7.1f}", flush=True)
