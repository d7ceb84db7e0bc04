#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE configs[2] (cfg3): 8 buoys (28 pairs), 10 MS/s,
4096-sample windows, 4096 windows batched per GPU, synthetic IQ already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (forward spectra -> pair kernels -> lags) over one batch of
4096 windows on every rank.  Windows shard across ranks with no data-path collective (weak scaling:
every rank owns 4096 windows); rank 0 gathers the per-pair lag scalars once after the timed region.
One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_on_device(torch, dev, W, B, N, fs, seed):
    """Same signal model as radio_mapper_amd.synth.make_windows, evaluated with torch on the GPU
    (input generation only; torch.fft is not part of the measured path)."""
    from radio_mapper_amd import synth
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    D = min(synth.max_delay_samples(fs), (N - 2) / 2.0)
    margin = int(np.ceil(D)) + 2
    Ns = 1
    while Ns < N + 2 * margin:
        Ns *= 2
    freqs = torch.fft.fftfreq(Ns, device=dev, dtype=torch.float64)
    mask = (freqs.abs() <= 0.4).to(torch.float32)
    delays = (torch.rand((W, B), device=dev, generator=g, dtype=torch.float64) * 2 - 1) * D
    noise_amp = 10.0 ** (-10.0 / 20.0)
    scale = 32.0 / np.sqrt(1.0 + noise_amp ** 2)
    out = torch.empty((W, B, N, 2), device=dev, dtype=torch.float32)
    chunk = 128
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        c = w1 - w0
        S = torch.complex(torch.randn((c, Ns), device=dev, generator=g), torch.randn((c, Ns), device=dev, generator=g))
        S = S * mask * float(np.sqrt(Ns / (2.0 * float(mask.sum()))) * np.sqrt(Ns))
        ph = (-2.0 * np.pi) * freqs[None, None, :] * delays[w0:w1, :, None]
        ramp = torch.complex(torch.cos(ph).float(), torch.sin(ph).float())
        s = torch.fft.ifft(S[:, None, :] * ramp, dim=-1)[:, :, margin:margin + N]
        nz = torch.complex(torch.randn((c, B, N), device=dev, generator=g), torch.randn((c, B, N), device=dev, generator=g))
        x = (s + nz * float(noise_amp / np.sqrt(2.0))) * float(scale)
        xr = torch.view_as_real(x)
        # rtl_sdr grid: uint8 - 127.5 (buoy_node.py:392-398)
        out[w0:w1] = torch.clamp(torch.floor(xr + 128.0), 0, 255) - 127.5
    return out, delays.cpu().numpy()


def cpu_baseline(iq_sample, budget_s=20.0):
    """The oracle (numpy/scipy restatement) timed on this box's host cores on a bounded sample."""
    from oracle import xcorr_ref as orc
    # threads actually used: the box's CPU share for one GPU is 16 (never the 256 logical CPUs)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("RMX_CPU_THREADS", "16"))))
    W, B, N = iq_sample.shape
    P = B * (B - 1) // 2
    # variant (i): literal per-pair scipy.signal.correlate loop, one core
    w_lit = max(1, min(W, 8))
    t0 = time.perf_counter()
    orc.xcorr_batch_literal(iq_sample[:w_lit])
    t_lit = time.perf_counter() - t0
    lit_rate = w_lit * P * N / t_lit
    # variant (ii): batched scipy.fft with spectrum reuse, all cores
    w_fast, t_fast, done = 16, 0.0, 0
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < budget_s * 0.6 and done < W:
        w1 = min(W, done + w_fast)
        t0 = time.perf_counter()
        orc.xcorr_batch_fast(iq_sample[done:w1], workers=cores)
        t_fast += time.perf_counter() - t0
        done = w1
    fast_rate = done * P * N / t_fast
    return {"value": fast_rate, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"{done} of the cfg3 windows (B={B}, N={N}, {P} pairs) with oracle.xcorr_batch_fast "
                      f"(scipy.fft workers={cores}, spectrum reuse); literal per-pair scipy.signal.correlate loop "
                      f"on 1 core: {lit_rate:.3e} samples/s over {w_lit} windows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--windows", type=int, default=4096, help="windows per GPU (cfg3: 4096)")
    ap.add_argument("--buoys", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as ge
    from radio_mapper_amd import xcorr
    from radio_mapper_amd.shard import gather_lags, window_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal knobs for a one-GPU box (never set by the driver): RMX_BENCH_BACKEND=gloo and
    # RMX_BENCH_SAME_DEVICE=1 run all ranks on cuda:0 so that the N>1 code path can be exercised
    backend = os.environ.get("RMX_BENCH_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("RMX_BENCH_SAME_DEVICE") == "1" else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    n_gpus = world if world > 1 else 1
    if rank == 0:
        ge.build()
    if world > 1:
        barrier()
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    B, N, fs = args.buoys, 4096, 10e6
    W_total = args.windows * n_gpus
    w_start, W = window_shard(W_total, rank, n_gpus)
    P = B * (B - 1) // 2
    # every rank generates its own block of the job's windows (seed 1003 = cfg3, offset by rank)
    x, delays = synth_on_device(torch, dev, W, B, N, fs, seed=1003 + 7919 * rank)
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, P), dtype=torch.float32, device=dev)

    eng = xcorr.XcorrEngine(B, N, W, device=dev_index)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    eng.set_option("timing", 1)   # HIP events around every launch, on the launch stream

    def step():
        eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            barrier()
            torch.cuda.synchronize()

    # untimed pre-warm beyond the W warm-up steps: the device needs some hundred milliseconds of load to
    # settle on its sustained clock (measured: 20 timed steps right after 5 warm-up steps run 8 % slower
    # per step than 3000); reported as "prewarm_steps" in the JSON line
    prewarm = max(0, int(os.environ.get("RMX_BENCH_PREWARM", "300")) - args.warmup)
    for _ in range(prewarm):
        step()
    for _ in range(args.warmup):
        step()
    sync_all()
    fwd_ms = pair_ms = 0.0
    fwd_n = pair_n = 0
    # HIP events on the launch stream (the ctx uses torch's current stream) around the timed region:
    # the sustained, back-to-back launch duration (isolated launches run at a higher clock)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
        # (rmx_last_timing would synchronise; it is read once per step only after the loop below)
    ev1.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    # kernel durations from the HIP events bracketing every launch on the launch stream: the last
    # step of the timed region, then the same step repeated with a read-back after each
    n_meas = max(min(args.steps, 10), 1)
    for k in range(n_meas):
        if k > 0:
            step()
        tm = eng.last_timing()
        fwd_ms += tm["fwd_ms"]; fwd_n += tm["fwd_launches"]
        pair_ms += tm["pair_ms"]; pair_n += tm["pair_launches"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps

    # PCIe-inclusive rate of the same step through the host-pointer entry of the C ABI (reported in
    # DESIGN.md, never `value`): pageable numpy in, numpy out
    host_ms = None
    if rank == 0 and n_gpus == 1:
        xh = x.cpu().numpy().view(np.complex64).reshape(W, B, N)
        eng.correlate(xh[:64])
        th = time.perf_counter()
        eng.correlate(xh)
        host_ms = (time.perf_counter() - th) * 1e3
        del xh

    # host-side gather of the lag scalars (the only exchange of the multi-GPU path)
    li, lf, pk = lag.cpu().numpy(), frac.cpu().numpy(), peak.cpu().numpy()
    if world > 1 and backend == "nccl":
        # every rank holds the same number of windows: plain all_gather of the three result tensors
        # (12 bytes per pair-window; outside the timed region)
        outs = []
        for tns in (lag, frac, peak):
            bucket = [torch.empty_like(tns) for _ in range(world)]
            dist.all_gather(bucket, tns)
            outs.append(torch.cat(bucket, dim=0).cpu().numpy() if rank == 0 else None)
        gathered = tuple(outs) if rank == 0 else None
    else:
        gathered = gather_lags(li, lf, pk) if world > 1 else (li, lf, pk)
    if rank == 0:
        assert gathered[0].shape == (W_total, P)

    # parity in the same run (rank 0, 32 windows) + cpu baseline
    parity = None
    cpu = None
    if rank == 0:
        from oracle import xcorr_ref as orc
        nchk = min(32, W)
        iq_chk = x[:nchk].cpu().numpy().view(np.complex64).reshape(nchk, B, N)
        ri, rf, rp = orc.xcorr_batch_fast(iq_chk, workers=16)
        ref = ri + rf
        got = li[:nchk] + lf[:nchk].astype(np.float64)
        parity = {"windows": nchk, "lag_int_mismatches": int(np.sum(li[:nchk] != ri)),
                  "max_lag_err_rel": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)))}
        if n_gpus == 1 and not args.no_cpu_baseline:
            ns = min(256, W)
            cpu = cpu_baseline(x[:ns].cpu().numpy().view(np.complex64).reshape(ns, B, N))

    if rank == 0:
        units_per_step = W_total * P * N                      # IQ samples cross-correlated per step
        value = units_per_step / (ms_per_step * 1e-3)
        alg_bytes_per_pw = 16 * N + 12
        isolated_launch_ms = pair_ms / max(pair_n, 1)
        launches_per_step = max(pair_n / n_meas, 1.0)
        # average duration of the dominant kernel's launches inside the timed region
        pair_launch_ms = region_ms / (args.steps * launches_per_step) if fwd_n == 0 else isolated_launch_ms
        windows_per_launch = W / launches_per_step
        alg_bytes_per_launch = windows_per_launch * P * alg_bytes_per_pw
        achieved = alg_bytes_per_launch / (pair_launch_ms * 1e-3) / 1e9 if pair_launch_ms > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "IQ samples cross-correlated per second (8-buoy, 10 MS/s, 4096 windows/GPU)",
            "value": value, "unit": "samples/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": prewarm, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg3: 8 buoys (28 pairs), 10 MS/s complex64, N=4096-sample windows "
                                   "(L=8192), 4096 windows per GPU resident in HBM",
                       "n_buoys": B, "n_pairs": P, "n_samples": N, "windows_per_gpu": W,
                       "windows_total": W_total, "parallelism": f"windows sharded x{n_gpus}, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_win (fused forward + pair kernel)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "launch_ms": pair_launch_ms, "isolated_launch_ms": isolated_launch_ms,
                         "fwd_kernel_ms_per_step": fwd_ms / n_meas,
                         "pair_kernel_ms_per_step": pair_ms / n_meas,
                         "whole_path_frac": (W * P * alg_bytes_per_pw) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "host_path_ms_per_step": host_ms,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
