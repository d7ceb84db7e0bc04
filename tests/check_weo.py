"""GPU box: the N = 16384 whole-window kernel (g_win_eo15, win_eo.hpp) against the oracle and against the four-step path\non the same input (complex64, raw uint8, a custom pair list), then timings of both.   usage: python tests/check_weo.py"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr
from oracle import xcorr_ref as orc
N = 16384
for B, W in ((3, 4), (8, 3), (2, 5)):
    iq, d, raw = rm.synth.make_windows(W, B, N, 2.048e6, seed=500 + B, return_u8=True)
    ri, rf, rp = orc.xcorr_batch_literal(iq)
    xcorr.set_default_option("wscr", 2)
    with xcorr.XcorrEngine(B, N, W) as eng:
        li, lf, pk = eng.correlate(iq)
        l8, f8, p8 = eng.correlate(raw)
        custom = np.array([(B - 1, 0), (0, 1), (1, 1)], np.int32)
        ci, cf, cp = eng.correlate(iq, custom)
    xcorr.set_default_option("wscr", 0)
    with xcorr.XcorrEngine(B, N, W) as eng:
        oi, of_, op = eng.correlate(iq)
        oc = eng.correlate(iq, custom)
    xcorr.clear_default_options()
    print(B, W, "lag_int mismatches vs oracle:", int(np.sum(li != ri)), "vs four-step:", int(np.sum(li != oi)),
          "max frac err", float(np.max(np.abs((li + lf) - (ri + rf)) / np.maximum(np.abs(ri + rf), 1))),
          "peak rel", float(np.max(np.abs(pk - rp) / rp)), "u8 same:", bool(np.array_equal(li, l8) and np.array_equal(lf, f8)),
          "custom ok:", bool(np.array_equal(ci, oc[0])))
import torch
for B, W in ((8, 256), (3, 512), (16, 128)):
    iq, _ = rm.synth.make_windows(W, B, N, 2.048e6, seed=1)
    x = torch.from_numpy(iq.view(np.float32)).cuda()
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device='cuda'); fr = torch.zeros((W, P), device='cuda'); pkk = torch.zeros((W, P), device='cuda')
    for opt in (1, 0):
        xcorr.set_default_option("wscr", opt)
        with xcorr.XcorrEngine(B, N, W) as eng:
            eng.set_stream(torch.cuda.current_stream().cuda_stream)
            for _ in range(10):
                eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pkk.data_ptr())
            torch.cuda.synchronize()
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), fr.data_ptr(), pkk.data_ptr())
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ms = sorted(ts)[len(ts) // 2] * 1e3
        print("B=%d W=%d wscr=%d: %.3f ms = %.1f %% of the 8 TB/s algorithmic roofline" % (B, W, opt, ms, 100 * W * P * (16 * N + 12) / (ms * 1e-3) / 8e12))
    xcorr.clear_default_options()
