#!/usr/bin/env python3
"""GPU box: the fused row kernel (g_rows_fused) against the two-kernel row passes on the same input, every row
length it is compiled for (2^9 .. 2^12) and 2..4 buoys.  Integer lags must agree exactly; lag_frac / peak to 1e-5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from radio_mapper_amd import xcorr

def run(B, N, W, fused):
    xcorr.set_default_option("fused", int("2" if fused else "0"))
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(11 + B + N)
    base = torch.randn((W, 1, N + 4096, 2), device=dev, generator=g) * 20.0
    x = torch.empty((W, B, N, 2), device=dev)
    for b in range(B):                      # shifted copies + noise: a real peak at a known lag
        x[:, b] = base[:, 0, 37 * b * b + 5 * b: 37 * b * b + 5 * b + N] + torch.randn((W, N, 2), device=dev, generator=g) * 5.0
    P = B * (B - 1) // 2
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), device=dev); peak = torch.zeros((W, P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    eng.close()
    return lag.cpu(), frac.cpu(), peak.cpu()

bad = 0
for logN in (14, 15, 16, 17, 18, 19, 20) if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]:
    for B in (2, 3, 4):
        N, W = 1 << logN, 3
        a = run(B, N, W, True); b = run(B, N, W, False)
        ok = bool((a[0] == b[0]).all()) and float((a[1] - b[1]).abs().max()) < 1e-5 and \
             float(((a[2] - b[2]).abs() / b[2].abs()).max()) < 1e-5
        print(f"N=2^{logN} B={B}: lags {a[0][0].tolist()} dfrac {float((a[1] - b[1]).abs().max()):.2e} "
              f"dpeak {float(((a[2] - b[2]).abs() / b[2].abs()).max()):.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
