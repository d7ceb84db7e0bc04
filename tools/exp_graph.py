"""GPU box: does a captured graph of one engine call beat the plain launches on launch-bound shapes?
usage: python tools/exp_graph.py   (cfg1: 3 buoys x one 2^18 window; single windows of 4096 / 8192 / 16384 samples)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr


def run(B, N, W):
    iq, _ = rm.synth.make_windows(W, B, N, 2.4e6, seed=7)
    x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device='cuda'); fr = torch.zeros((W, P), device='cuda'); pk = torch.zeros((W, P), device='cuda')
    with xcorr.XcorrEngine(B, N, W) as eng:
        s = torch.cuda.current_stream()
        eng.set_stream(s.cuda_stream)
        call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pk.data_ptr())
        for _ in range(20): call()
        torch.cuda.synchronize()
        ref = (lag.clone(), fr.clone(), pk.clone())

        def timed(fn, n=300):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        plain = min(timed(call) for _ in range(3))
        # capture
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        with torch.cuda.stream(cs):
            eng.set_stream(cs.cuda_stream)
            call(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cs):
                call()
        torch.cuda.synchronize()
        lag.zero_(); fr.zero_(); pk.zero_()
        g.replay(); torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(ref, (lag, fr, pk)))
        graph = min(timed(g.replay) for _ in range(3))
        print(f"B={B} N={N:7d} W={W:3d}  plain {plain:7.1f} us/call   graph replay {graph:7.1f} us/call   results identical: {same}", flush=True)


for B, N, W in ((3, 1 << 18, 1), (3, 4096, 1), (8, 4096, 1), (3, 8192, 1), (3, 16384, 1), (8, 16384, 1), (3, 1 << 20, 1), (8, 4096, 16)):
    run(B, N, W)
