"""GPU box: N = 4096, few windows -- the fused one-workgroup-per-window kernel against the per-transform kernels
(k_fwd + k_pair_res / k_pair_str with 7 / 4 / 2 / 1 pairs per workgroup).  usage: python tools/exp_small4096.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr

N = 4096


def run(B, W, opts):
    iq, _ = rm.synth.make_windows(W, B, N, 10e6, seed=11)
    x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device='cuda'); fr = torch.zeros((W, P), device='cuda'); pk = torch.zeros((W, P), device='cuda')
    with xcorr.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for k, v in opts.items():
            eng.set_option(k, v)
        call = lambda: eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pk.data_ptr())
        for _ in range(30): call()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(200): call()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e6)
        return best, lag.cpu().numpy().copy()


for B in (3, 8, 16):
    for W in (1, 4, 16, 32, 64, 128, 256, 300, 600, 1100):
        auto, ref = run(B, W, {})
        base, ref0 = run(B, W, {"small4096": 0})
        row = [f"default {auto:7.1f}{'' if np.array_equal(ref, ref0) else ' !!'}", f"fused {base:7.1f}"]
        for name, o in (("res7", {"fused": 0}), ("res2", {"fused": 0, "pairs_per_block": 2}), ("res1", {"fused": 0, "pairs_per_block": 1}),
                        ("str4", {"fused": 0, "resident": 0, "pairs_per_block": 4}), ("str1", {"fused": 0, "resident": 0, "pairs_per_block": 1})):
            t, got = run(B, W, o)
            row.append(f"{name} {t:7.1f}{'' if np.array_equal(got, ref) else ' !!'}")
        print(f"B={B:2d} W={W:3d} us/call: " + "  ".join(row), flush=True)
