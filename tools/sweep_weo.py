"""GPU box: N = 16384 -- g_win_eo15 (option wscr = 2: forced) against the four-step path (wscr = 0) over buoy counts and
batch sizes; the dispatch rule of rmx_hip.hip (generic_batch) is read off this table.   usage: python tools/sweep_weo.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr

N = 16384


def run(B, W, opt):
    iq, _ = rm.synth.make_windows(min(W, 16), B, N, 2.048e6, seed=1)
    iq = np.concatenate([iq] * ((W + iq.shape[0] - 1) // iq.shape[0]))[:W]
    x = torch.from_numpy(np.ascontiguousarray(iq).view(np.float32)).cuda()
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device='cuda'); fr = torch.zeros((W, P), device='cuda'); pk = torch.zeros((W, P), device='cuda')
    xcorr.set_default_option("wscr", opt)
    with xcorr.XcorrEngine(B, N, W) as eng:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        for _ in range(6):
            eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pk.data_ptr())
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pk.data_ptr())
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    xcorr.clear_default_options()
    return sorted(ts)[len(ts) // 2] * 1e3


print("B    W   | eo15 ms  frac | four-step ms  frac | eo/four-step")
for B in (2, 3, 4, 6, 8, 12, 16):
    for W in (64, 128, 192, 256, 512):
        if B * W > 6144:
            continue
        a, b = run(B, W, 2), run(B, W, 0)
        alg = W * (B * (B - 1) // 2) * (16 * N + 12)
        print("%-3d %4d | %7.3f  %.3f | %9.3f  %.3f | %.2f" % (B, W, a, alg / (a * 1e-3) / 8e12, b, alg / (b * 1e-3) / 8e12, a / b), flush=True)
