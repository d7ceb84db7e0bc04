"""GPU box: latency of the Doppler search on small N = 4096 batches (host arrays, and device pointers with the per-kernel
HIP-event breakdown left to rocprofv3).   usage: python tools/exp_caf_latency.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr

for B, N, W, D in ((8, 4096, 1, 21), (3, 4096, 1, 21), (8, 4096, 4, 21), (8, 4096, 16, 21), (8, 16384, 1, 21)):
    iq, _ = rm.synth.make_windows(W, B, N, 2.4e6, seed=3)
    dop = (np.arange(D) - D // 2) * 50.0 / 2.4e6
    P = B * (B - 1) // 2
    x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    o = [torch.zeros((W, P), dtype=torch.int32, device='cuda') for _ in range(2)] + [torch.zeros((W, P), device='cuda') for _ in range(2)]
    with xcorr.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(10): eng.caf(iq, dop)
        t0 = time.perf_counter()
        for _ in range(50): eng.caf(iq, dop)
        host = (time.perf_counter() - t0) / 50 * 1e6
        call = lambda: eng.caf_device(x.data_ptr(), W, dop, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr())
        for _ in range(10): call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): call()
        torch.cuda.synchronize()
        dev = (time.perf_counter() - t0) / 50 * 1e6
    print(f"caf B={B} N={N} W={W} D={D}: {host:7.0f} us per call from host arrays, {dev:7.0f} us with device pointers", flush=True)
