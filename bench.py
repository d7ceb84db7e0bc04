#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: IQ samples cross-correlated per second + fraction of the HBM
roofline, on synthetic IQ already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg4|cfg5|cfg1]

Default workload = BASELINE configs[2] ("cfg3", the configuration the metric is quoted on): 8 buoys
(28 pairs), 10 MS/s, 4096-sample windows, 4096 windows batched per GPU.  The other BASELINE shapes
are selectable (they are parity-test cases; each prints the same kind of line for its own shape).

A step = one pass of the hot path (IQ windows -> per-pair lags) over one batch on every rank.
Windows shard across ranks with no data-path collective (weak scaling: every rank owns a full batch);
rank 0 gathers the per-pair lag scalars once after the timed region.  One JSON line is printed by
rank 0.

Multi-GPU: `python bench.py --gpus N` with WORLD_SIZE unset starts its own N ranks
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) as a CHILD process
before this process has imported torch or touched the GPU, relays the child's output and exits with
its code.  Launched under torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# BASELINE.json configs as bench shapes (per GPU)
CONFIGS = {
    # name: buoys, samples, windows per GPU, channels per GPU-window, fs, seed, doppler grid (Hz half-width, step)
    "cfg1": dict(B=3, N=262144, W=1, C=1, fs=2.4e6, seed=1001,
                 what="cfg1: 3 buoys, 2.4 MS/s, one 262144-sample window (the reference's CPU-runnable case)"),
    "cfg2": dict(B=3, N=1048576, W=64, C=1, fs=2.4e6, seed=1002,
                 what="cfg2: 3 buoys (3 pairs), 2.4 MS/s complex64, 64 windows of 1048576 samples (L=2^21)"),
    "cfg3": dict(B=8, N=4096, W=4096, C=1, fs=10e6, seed=1003,
                 what="cfg3: 8 buoys (28 pairs), 10 MS/s complex64, N=4096-sample windows (L=8192), "
                      "4096 windows per GPU resident in HBM"),
    "cfg4": dict(B=16, N=4096, W=512, C=10, fs=10e6, seed=1004,
                 what="cfg4: 16 buoys (120 pairs) x 10 frequency channels, 10 MS/s, N=4096; 4096 windows sharded "
                      "over 8 GPUs = 512 windows x 10 channels per GPU"),
    "cfg5": dict(B=32, N=262144, W=8, C=1, fs=20e6, seed=1005, doppler_hz=500.0, doppler_step_hz=50.0,
                 what="cfg5: 32 buoys (496 pairs), 20 MS/s, N=262144, Doppler grid +-500 Hz step 50 Hz (21 bins); "
                      "64 windows sharded over 8 GPUs = 8 windows per GPU"),
}
DEFAULT_STEPS = {"cfg1": (20, 3), "cfg2": (30, 5), "cfg3": (200, 30), "cfg4": (30, 5), "cfg5": (3, 1)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Parent of an N-rank run: nothing here imports torch or touches HIP.  The ranks are a child
    process tree (never an exec of this process); stdout/stderr are inherited, so rank 0's JSON line
    is this command's JSON line."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def synth_on_device(torch, dev, W, B, N, fs, seed, doppler_cps=None):
    """Same signal model as radio_mapper_amd.synth.make_windows, evaluated with torch on the GPU
    (input generation only; torch.fft is not part of the measured path).  doppler_cps: [B] per-buoy
    frequency offsets in cycles/sample (cfg5)."""
    import numpy as np
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
    chunk = max(1, min(128, (1 << 24) // (Ns * B)))
    rot = None
    if doppler_cps is not None:
        nu = torch.as_tensor(np.asarray(doppler_cps, np.float64), device=dev)
        ph = (2.0 * np.pi) * nu[:, None] * torch.arange(N, device=dev, dtype=torch.float64)[None, :]
        rot = torch.complex(torch.cos(ph).float(), torch.sin(ph).float())
    for w0 in range(0, W, chunk):
        w1 = min(W, w0 + chunk)
        c = w1 - w0
        S = torch.complex(torch.randn((c, Ns), device=dev, generator=g), torch.randn((c, Ns), device=dev, generator=g))
        S = S * mask * float(np.sqrt(Ns / (2.0 * float(mask.sum()))) * np.sqrt(Ns))
        ph = (-2.0 * np.pi) * freqs[None, None, :] * delays[w0:w1, :, None]
        ramp = torch.complex(torch.cos(ph).float(), torch.sin(ph).float())
        s = torch.fft.ifft(S[:, None, :] * ramp, dim=-1)[:, :, margin:margin + N]
        if rot is not None:
            s = s * rot[None]
        nz = torch.complex(torch.randn((c, B, N), device=dev, generator=g), torch.randn((c, B, N), device=dev, generator=g))
        x = (s + nz * float(noise_amp / np.sqrt(2.0))) * float(scale)
        xr = torch.view_as_real(x)
        # rtl_sdr grid: uint8 - 127.5 (buoy_node.py:392-398)
        out[w0:w1] = torch.clamp(torch.floor(xr + 128.0), 0, 255) - 127.5
    return out, delays.cpu().numpy()


def cpu_info():
    import numpy as np
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    try:
        import scipy
        sv = scipy.__version__
    except Exception:
        sv = None
    return {"cpu_model": model, "numpy": np.__version__, "scipy": sv, "logical_cpus": os.cpu_count()}


def cpu_threads():
    # threads actually used: the box's CPU share for one GPU is 16 (never the 256 logical CPUs)
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    return max(1, min(avail, int(os.environ.get("RMX_CPU_THREADS", "16"))))


def cpu_baseline(get_windows, W, B, N, budget_s, doppler=None):
    """The oracle (numpy/scipy restatement) timed on this box's host cores on a BOUNDED sample of the
    same workload: the sample grows until it has cost >= 2 s and stops at
    budget_s.  get_windows(a, b) returns complex64 [b-a][B][N] host windows of the workload."""
    from oracle import xcorr_ref as orc
    cores = cpu_threads()
    P = B * (B - 1) // 2
    info = cpu_info()
    if doppler is not None:
        # CAF: literal per-pair, per-bin loop on 1 core over the first pairs of window 0
        iq = get_windows(0, 1)
        pairs = orc.pair_list(B)
        D = len(doppler)
        n_done, t0 = 0, time.perf_counter()
        while n_done < P and (time.perf_counter() - t0) < max(2.0, budget_s * 0.5):
            i, j = pairs[n_done]
            orc.caf_pair(iq[0, i], iq[0, j], doppler)
            n_done += 1
        dt = time.perf_counter() - t0
        return {"value": n_done * D * N / dt, "unit": "samples/s (pair-window-Doppler-bin samples)", "cores": 1,
                "kind": "port", "seconds": dt,
                "sample": f"oracle.caf_pair (scipy.signal.correlate per pair and bin) on the first {n_done} of {P} "
                          f"pairs of window 0, {D} Doppler bins, N={N}", **info}
    # variant (i): literal per-pair scipy.signal.correlate loop, one core
    w_lit = 1 if N >= 65536 else max(1, min(W, 8))
    t0 = time.perf_counter()
    orc.xcorr_batch_literal(get_windows(0, w_lit))
    t_lit = time.perf_counter() - t0
    lit_rate = w_lit * P * N / t_lit
    # variant (ii): batched scipy.fft with spectrum reuse, all cores; >= 2 s of work, <= budget
    step_w = 1 if N >= 65536 else 64
    done, t_fast = 0, 0.0
    while done < W and (t_fast < 2.0 or (t_fast < 0.5 * budget_s and done < 1024)):
        w1 = min(W, done + step_w)
        blk = get_windows(done, w1)
        t0 = time.perf_counter()
        orc.xcorr_batch_fast(blk, workers=cores)
        t_fast += time.perf_counter() - t0
        done = w1
        if t_fast + t_lit > budget_s:
            break
    if done == W and t_fast < 2.0:       # small workloads (cfg1): repeat the whole batch
        reps = 0
        blk = get_windows(0, W)
        t_fast = 0.0
        while t_fast < 2.0:
            t0 = time.perf_counter()
            orc.xcorr_batch_fast(blk, workers=cores)
            t_fast += time.perf_counter() - t0
            reps += 1
        done = W * reps
    return {"value": done * P * N / t_fast, "unit": "samples/s", "cores": cores, "kind": "port", "seconds": t_fast,
            "sample": f"{done} windows of this workload (B={B}, N={N}, {P} pairs) with oracle.xcorr_batch_fast "
                      f"(scipy.fft workers={cores}, spectrum reuse); literal per-pair scipy.signal.correlate loop "
                      f"on 1 core: {lit_rate:.3e} samples/s over {w_lit} window(s) ({t_lit:.2f} s)", **info}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--windows", type=int, default=None, help="windows per GPU (default: the config's)")
    ap.add_argument("--buoys", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for cpu_baseline")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + shard + gather only (no compute, no metric): tests the N-rank launcher")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with --nproc-per-node {args.gpus}",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    # rehearsal knobs for a one-GPU box (never set by the driver): RMX_BENCH_SAME_DEVICE=1 runs all
    # ranks on cuda:0 (with the gloo backend: RCCL refuses two ranks on one device)
    same_dev = os.environ.get("RMX_BENCH_SAME_DEVICE") == "1"
    # --launch-check never initialises a device (it is documented, and tested, as needing no GPU: on a box with fewer
    # GPUs than ranks the nccl branch would set_device() a card that does not exist)
    backend = os.environ.get("RMX_BENCH_BACKEND", "gloo" if (same_dev or args.launch_check) else "nccl")
    dev_index = 0 if same_dev else local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    def barrier():
        if backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    def ranks_seen():
        if world == 1:
            return 1
        t = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", dev_index) if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        return int(t.item())

    from radio_mapper_amd.shard import gather_lags, window_shard

    if args.launch_check:
        # the launcher's contract without a GPU: every rank takes its window block, rank 0 gathers
        s, c = window_shard(64 * world, rank, world)
        li = np.full((c, 3), rank, np.int32)
        got = gather_lags(li, li.astype(np.float32), li.astype(np.float32)) if world > 1 else (li,) * 3
        seen = ranks_seen()
        if rank == 0:
            ok = got[0].shape == (64 * world, 3) and all(int(got[0][64 * r, 0]) == r for r in range(world))
            print(json.dumps({"launch_check": bool(ok), "n_gpus": world, "ranks_seen": seen, "backend": backend if world > 1 else None}),
                  flush=True)
        if world > 1:
            barrier()
            dist.destroy_process_group()
        return

    import __graft_entry__ as ge
    from radio_mapper_amd import xcorr

    n_gpus = world
    if rank == 0:
        ge.build()
    if world > 1:
        barrier()
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    cfg = dict(CONFIGS[args.config])
    B = args.buoys or cfg["B"]
    N, fs, C = cfg["N"], cfg["fs"], cfg["C"]
    W = (args.windows or cfg["W"]) * C            # (window, channel) units per GPU: channels are a batch axis
    steps, warm = DEFAULT_STEPS[args.config]
    steps = args.steps if args.steps is not None else steps
    warm = args.warmup if args.warmup is not None else warm
    W_total = W * n_gpus
    w_start, W_rank = window_shard(W_total, rank, n_gpus)
    assert W_rank == W
    P = B * (B - 1) // 2
    caf = "doppler_hz" in cfg
    grid = None
    buoy_dop = None
    if caf:
        nb = int(round(cfg["doppler_hz"] / cfg["doppler_step_hz"]))
        grid = np.arange(-nb, nb + 1) * (cfg["doppler_step_hz"] / fs)          # cycles/sample, 21 bins
        rng = np.random.default_rng(cfg["seed"])
        buoy_dop = rng.integers(-8, 9, size=B) * (cfg["doppler_step_hz"] / fs) * 0.5   # true offsets within +-200 Hz
    D = len(grid) if caf else 1
    # every rank generates its own block of the job's windows (seed per config, offset by rank)
    x, delays = synth_on_device(torch, dev, W, B, N, fs, seed=cfg["seed"] + 7919 * rank, doppler_cps=buoy_dop)
    lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
    frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
    peak = torch.zeros((W, P), dtype=torch.float32, device=dev)
    dop = torch.zeros((W, P), dtype=torch.int32, device=dev) if caf else None

    eng = xcorr.XcorrEngine(B, N, W, device=dev_index)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)
    eng.set_option("timing", 1)   # HIP events around every launch, on the launch stream

    calls = [0]      # full-size engine calls made by this process (tools/summarize_prof.py divides counter totals by it)
    if caf:
        def step():
            calls[0] += 1
            eng.caf_device(x.data_ptr(), W, grid, dop.data_ptr(), lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
    else:
        def step():
            calls[0] += 1
            eng.correlate_device(x.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            barrier()
            torch.cuda.synchronize()

    # untimed pre-warm beyond the W warm-up steps: the device needs some hundred milliseconds of load to
    # settle on its sustained clock (measured: 20 timed steps right after 5 warm-up steps run 8 % slower
    # per step than 3000); reported as "prewarm_steps" in the JSON line
    prewarm = 0
    if args.config == "cfg3":
        prewarm = max(0, int(os.environ.get("RMX_BENCH_PREWARM", "300")) - warm)
    for _ in range(prewarm):
        step()
    for _ in range(warm):
        step()
    sync_all()
    # HIP events on the launch stream (the ctx uses torch's current stream) around the timed region:
    # the sustained, back-to-back launch duration (isolated launches run at a higher clock)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        step()
        # (rmx_last_timing would synchronise; it is read once per step only after the loop below)
    ev1.record(stream)
    sync_all()
    elapsed = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    # kernel durations from the HIP events bracketing every launch on the launch stream: the last
    # step of the timed region, then the same step repeated with a read-back after each
    fwd_ms = pair_ms = 0.0
    fwd_n = pair_n = 0
    n_meas = max(min(steps, 10 if not caf else 1), 1)
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
    ms_per_step = elapsed * 1e3 / steps
    seen = ranks_seen()

    # PCIe-inclusive rate of the same step through the host-pointer entry of the C ABI (reported in
    # DESIGN.md, never `value`): pageable numpy in, numpy out
    host_ms = None
    if rank == 0 and n_gpus == 1 and args.config == "cfg3":
        xh = x.cpu().numpy().view(np.complex64).reshape(W, B, N)
        eng.correlate(xh[:64])
        th = time.perf_counter()
        eng.correlate(xh)
        calls[0] += 1
        host_ms = (time.perf_counter() - th) * 1e3
        del xh
        step()                    # leave the device-pointer results of the timed path in lag/frac/peak
        torch.cuda.synchronize()

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

    def host_windows(a, b):
        return x[a:b].cpu().numpy().view(np.complex64).reshape(b - a, B, N)

    # parity in the same run (rank 0) + cpu baseline
    parity = None
    cpu = None
    if rank == 0:
        from oracle import xcorr_ref as orc
        true_lag = delays[:, orc.pair_list(B)[:, 1]] - delays[:, orc.pair_list(B)[:, 0]]
        truth_ok = float(np.mean(np.abs(li + lf - true_lag) < 1.0))
        if caf:
            # oracle on a bounded subset: window 0, the first 32 pairs (all pairs of buoy 0 and (1, 2)), the whole
            # Doppler grid -- 672 calls of the reference primitive at cfg5 (about 30 s on the box's host cores)
            npr = min(32, P)
            prs = orc.pair_list(B)[:npr]
            rd, ri, rf, rp = orc.caf_batch(host_windows(0, 1), grid, prs)
            dgot = dop.cpu().numpy()
            ref, got = ri + rf, li[:1, :npr] + lf[:1, :npr].astype(np.float64)
            parity = {"windows": 1, "pairs": npr, "doppler_bins": D,
                      "doppler_idx_mismatches": int(np.sum(dgot[:1, :npr] != rd)),
                      "lag_int_mismatches": int(np.sum(li[:1, :npr] != ri)),
                      "max_lag_err_rel": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))),
                      "lags_within_1_sample_of_truth": truth_ok}
        else:
            nchk = min(256 if N <= 4096 else 8, W)           # >= 256 windows at cfg3; 8 of the long ones
            ri, rf, rp = orc.xcorr_batch_fast(host_windows(0, nchk), workers=cpu_threads())
            ref = ri + rf
            got = li[:nchk] + lf[:nchk].astype(np.float64)
            parity = {"windows": nchk, "pair_windows": int(nchk * P),
                      "lag_int_mismatches": int(np.sum(li[:nchk] != ri)),
                      "max_lag_err_rel": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))),
                      "lags_within_1_sample_of_truth": truth_ok}
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(host_windows, W, B, N, args.cpu_budget, doppler=grid)
        # latency of ONE window of the same shape (the reference's call pattern: one frequency group per call), device
        # pointers, outside the timed region; extra information, not part of the metric
        single = None
        if not caf:
            eng.set_option("timing", 0)
            one = lambda: eng.correlate_device(x.data_ptr(), 1, lag.data_ptr(), frac.data_ptr(), peak.data_ptr())
            for _ in range(20):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                one()
            torch.cuda.synchronize()
            single = {"windows": 1, "us_per_call": (time.perf_counter() - t0) / 200 * 1e6}

    if rank == 0:
        units_per_step = W_total * P * N * D                  # IQ samples cross-correlated per step
        value = units_per_step / (ms_per_step * 1e-3)
        # algorithmic bytes: 16 N + 12 per pair-window; CAF: 16 N per pair-window-bin
        alg_bytes_per_step_gpu = W * P * ((16 * N) * D + (16 if caf else 12))
        fused = (N == 4096 and not caf)
        if fused:
            launches_per_step = max(pair_n / n_meas, 1.0)
            kernel = "k_win (fused forward + pair kernel), one launch per step"
            launch_ms = region_ms / (steps * launches_per_step)
            isolated_launch_ms = pair_ms / max(pair_n, 1)
        else:
            # multi-kernel paths: the figure is for the whole kernel sequence of one step (HIP events on the
            # launch stream around the timed region); per-kernel durations are in profiles/r02_<cfg>_*
            launches_per_step = 1.0
            kernel = ("rmx_caf_batch kernel sequence (un-rotated spectra once; per Doppler bin: de-rotated forward "
                      "g_cols_fwd + g_rows, g_rows(product, inverse), g_cols_inv, g_final, k_caf_select)" if caf else
                      ("four-step sequence g_cols_fwd + g_rows_fused (forward rows, products, inverse rows) + g_cols_inv "
                       "+ g_final" if B <= 4 and W >= 4 else
                       "four-step sequence g_cols_fwd + g_rows + g_rows(product, inverse) + g_cols_inv + g_final"))
            launch_ms = region_ms / steps
            isolated_launch_ms = None
        alg_bytes_per_launch = alg_bytes_per_step_gpu / launches_per_step
        achieved = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic = traffic_src = None
        tj = os.path.join(ROOT, "profiles", "traffic_latest.json" if args.config == "cfg3" else f"traffic_{args.config}.json")
        if os.path.exists(tj):
            try:
                tjd = json.load(open(tj))
                traffic = tjd.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{os.path.basename(tj)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tag " \
                              f"{tjd.get('tag')}; recorded by tools/profile.sh, not measured in this run)"
            except Exception:
                traffic = None
        unit = "samples/s"
        line = {
            "metric": "IQ samples cross-correlated per second" + (" (pair-window-Doppler-bin samples)" if caf else ""),
            "value": value, "unit": unit, "n_gpus": n_gpus, "ranks_seen": seen, "steps": steps,
            "warmup": warm, "prewarm_steps": prewarm, "engine_calls": calls[0], "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["what"], "name": args.config,
                       "n_buoys": B, "n_pairs": P, "n_samples": N, "windows_per_gpu": W, "channels": C,
                       "doppler_bins": D if caf else None,
                       "windows_total": W_total, "parallelism": f"windows sharded x{n_gpus}, no collective"},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "launch_ms": launch_ms, "isolated_launch_ms": isolated_launch_ms,
                         "fwd_kernel_ms_per_step": fwd_ms / n_meas,
                         "pair_kernel_ms_per_step": pair_ms / n_meas,
                         "whole_path_frac": alg_bytes_per_step_gpu / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "host_path_ms_per_step": host_ms,
            "single_group": single,
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
