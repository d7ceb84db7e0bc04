import os, sys
sys.path.insert(0, os.getcwd())
import torch
from radio_mapper_amd import xcorr
xcorr.apply_env_options(report=sys.stderr)   # (a refused knob raises) RMX_<KEY>=<int> of the calling shell -> default options (the library reads no environment)
def run(W, B=8, N=4096, reps=7, **opts):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = (torch.randint(0, 256, (W, B, N, 2), device=dev, generator=g, dtype=torch.int32).float() - 127.5) * 0.25
    P = B*(B-1)//2
    lag = torch.zeros((W,P), device=dev, dtype=torch.int32); frac = torch.zeros((W,P), device=dev); peak = torch.zeros((W,P), device=dev)
    eng = xcorr.XcorrEngine(B, N, W)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    for k, v in opts.items(): eng.set_option(k, v)
    for _ in range(30): eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1)/reps
    eng.close()
    return t
if __name__ == "__main__":
    W = int(sys.argv[1]); opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
    t = run(W, reps=100, **opts)
    ncu = int(os.environ.get("RMX_NCUS", "256"))
    print(f"W={W} ncus={ncu} opts={opts} lib={os.environ.get('RMX_LIBRARY','default')[-20:]}: {t:.4f} ms  per-window-per-WG {t*1000/(W/min(W,ncu)):.1f} us", flush=True)
