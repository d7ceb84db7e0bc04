"""GPU box: the N = 8192 kernels on the fused kernel's network -- determinism, complex64 against raw uint8, the oracle, and
g_win_scr14 (option kwin8k = 0).  Default: the product's k_win8kl (kwin8k = 1).  `python tests/check_k8.py 2` checks the shelved
k_win8k instead and needs a -DRMX_EXPERIMENTS library (RMX_LIBRARY=/path/librmx_exp.so).  (The suite's own coverage of
k_win8kl: tests/test_gpu_parity.py, test_whole_window_scratch_kernel_by_length and the two N = 8192 tests behind it.)"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr
from oracle import xcorr_ref as orc
N = 8192
KIND = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for B, W in ((3, 300), (8, 300), (5, 2), (2, 600)):
    iq, d, raw = rm.synth.make_windows(W, B, N, 2.4e6, seed=800 + B, return_u8=True)
    xcorr.set_default_option("wscr", 2)
    xcorr.set_default_option("kwin8k", KIND)
    with xcorr.XcorrEngine(B, N, W) as eng:
        a = eng.correlate(iq)
        b = eng.correlate(iq)
        c = eng.correlate(raw)
        custom = np.array([(B - 1, 0), (0, 1), (1, 1)], np.int32)
        cu = eng.correlate(iq[:8], custom)
    xcorr.set_default_option("kwin8k", 0)
    with xcorr.XcorrEngine(B, N, W) as eng:
        o = eng.correlate(iq)
    xcorr.clear_default_options()
    nsub = min(W, 16)
    ri, rf, rp = orc.xcorr_batch_fast(iq[:nsub], workers=8)
    oi, of_, op = orc.xcorr_batch_literal(iq[:8], custom)
    print(f"B={B} W={W}: rerun identical {all(np.array_equal(x, y) for x, y in zip(a, b))}; u8 identical "
          f"{[bool(np.array_equal(x, y)) for x, y in zip(a, c)]} (max dfrac {np.abs(a[1] - c[1]).max():.2e}, dpeak rel {np.abs(a[2] / c[2] - 1).max():.2e}); "
          f"vs oracle: int mismatches {int(np.sum(a[0][:nsub] != ri))}, max lag err {np.abs(a[0][:nsub] + a[1][:nsub] - ri - rf).max():.2e}, "
          f"peak rel {np.abs(a[2][:nsub] / rp - 1).max():.2e}; vs g_win_scr14: int mismatches {int(np.sum(a[0] != o[0]))}, "
          f"max dfrac {np.abs(a[1] - o[1]).max():.2e}; custom pairs: int mismatches {int(np.sum(cu[0] != oi))}, lag err {np.abs(cu[0] + cu[1] - oi - of_).max():.2e}", flush=True)
