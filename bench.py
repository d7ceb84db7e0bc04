#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: IQ samples cross-correlated per second + fraction of the HBM
roofline, on synthetic IQ already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3|cfg2|cfg4|cfg5|cfg1] [--scaling weak|strong]

Default workload = BASELINE configs[2] ("cfg3", the configuration the metric is quoted on): 8 buoys
(28 pairs), 10 MS/s, 4096-sample windows, 4096 windows batched per GPU.  The other BASELINE shapes
are selectable (they are parity-test cases; each prints the same kind of line for its own shape).

A step = one pass of the hot path (IQ windows -> per-pair lags) over one batch on every rank.
Windows shard across ranks with no data-path collective -- `--scaling weak` (default): every rank owns a full
batch; `--scaling strong`: the config's windows are block-sharded over the ranks (radio_mapper_amd.shard.window_shard,
as BASELINE.json words cfg3/cfg4/cfg5) -- and rank 0 gathers the per-pair lag scalars once after the timed region.
One JSON line is printed by rank 0.  Headline figures (`ms_per_step`, `value`, `roofline`) come from the region that
follows EXACTLY `--warmup` steps of the headline shape; the same K steps after an untimed pre-warm are reported beside
them as `ms_per_step_sustained` (+ `prewarm_steps`).  At N = 1 the default run also times the other four BASELINE
shapes in the same process (`other_configs`, a bounded number of steps each, with per-kernel-family times;
`--no-other-configs` skips them) and a one-GPU projection of the 1 -> 8 GPU strong-scaling curve
(`strong_scaling_projection`) -- both run BEFORE the headline region (`region_order` says so) -- and reports the ingest
rate, the real-time factor, and the host-pointer path for complex64 and raw uint8 input.  The line names the binary
that ran (`build.binary_digest`) beside the digest of the sources it sees and the run refuses to start on a mismatch.

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
# The reference's own capture lengths (iq_stream_client.py:459: 8192 samples; buoy_node.py:364: 16384) at its fleet size
# (3 buoys) and at 8 buoys -- not BASELINE configs: reported beside `other_configs` as `reference_capture_lengths`, same
# bracket, same parity check
CAPTURE_SHAPES = {
    "cap8192_b3": dict(B=3, N=8192, W=1024, C=1, fs=2.4e6, seed=1011, what="3 buoys, 2.4 MS/s, 1024 windows of 8192 samples (iq_stream_client.py:459)"),
    "cap8192_b8": dict(B=8, N=8192, W=512, C=1, fs=2.4e6, seed=1012, what="8 buoys, 2.4 MS/s, 512 windows of 8192 samples"),
    "cap16384_b3": dict(B=3, N=16384, W=512, C=1, fs=2.4e6, seed=1013, what="3 buoys, 2.4 MS/s, 512 windows of 16384 samples (buoy_node.py:364)"),
    "cap16384_b8": dict(B=8, N=16384, W=256, C=1, fs=2.4e6, seed=1014, what="8 buoys, 2.4 MS/s, 256 windows of 16384 samples"),
}
CONFIGS.update(CAPTURE_SHAPES)
DEFAULT_STEPS.update({k: (30, 5) for k in CAPTURE_SHAPES})


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


def source_digest():
    """sha256 over everything the library is built from (__graft_entry__.source_digest: kernel sources, include/rmx.h,
    compiler flags).  The same value is compiled into the binary (rmx_build_info) and written beside every recorded
    profiles/traffic_*.json / pmc_latest.json (tools/summarize_prof.py); bench.py refuses a figure, or a binary, whose
    digest is not the current one."""
    import __graft_entry__ as ge
    return ge.source_digest()


def recorded_traffic(config):
    """(hbm bytes per launch, source text) from profiles/traffic_*.json -- only when that file was taken on the current
    sources; a stale file yields (None, why)."""
    tj = os.path.join(ROOT, "profiles", "traffic_latest.json" if config == "cfg3" else f"traffic_{config}.json")
    if not os.path.exists(tj):
        return None, None
    try:
        tjd = json.load(open(tj))
    except Exception:
        return None, None
    now = source_digest()
    if tjd.get("source_digest") != now:
        return None, (f"profiles/{os.path.basename(tj)} (tag {tjd.get('tag')}) was recorded on other sources "
                      f"(digest {tjd.get('source_digest')} != {now}): not reported; re-run tools/profile.sh")
    return tjd.get("hbm_bytes_per_launch"), (
        f"profiles/{os.path.basename(tj)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tag {tjd.get('tag')}, source "
        f"digest {now}; recorded by tools/profile.sh, not measured in this run)")


def recorded_counters(launch_ms, live=None):
    """SQ / LDS / L2 counters per launch of the fused kernel -- measured in this run (`live`: measure_pmc_in_run, the SQ
    set only) or from profiles/pmc_latest.json (tools/profile.sh passes, same digest rule as the traffic figure) -- as the
    utilisation ratios SURVEY.md section 8d asks for beside the HBM fraction."""
    if live and live.get("counters", {}).get("SQ_WAVE_CYCLES"):
        c = live["counters"]
        src = "measured in this run: rocprofv3 --pmc child pass of bench.py --profile, per launch of k_win"
    else:
        pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if not os.path.exists(pj):
            return None
        try:
            d = json.load(open(pj))
        except Exception:
            return None
        if d.get("source_digest") != source_digest():
            return {"stale": f"profiles/pmc_latest.json (tag {d.get('tag')}) was recorded on other sources; re-run tools/profile.sh"}
        c = d["counters"]
        src = f"profiles/pmc_latest.json (tag {d.get('tag')}): rocprofv3 --pmc passes of bench.py --profile, per launch of {d.get('kernel')}"
    wc = c.get("SQ_WAVE_CYCLES") or 0.0
    out = {"source": src,
           "valu_wave_instructions": c.get("SQ_INSTS_VALU"), "lds_wave_instructions": c.get("SQ_INSTS_LDS"),
           "salu_wave_instructions": c.get("SQ_INSTS_SALU")}
    if wc:
        out["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc   # saturates near 0.5 at two waves per SIMD
        out["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / wc
        out["wait_inst_any_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
        out["wait_inst_lds_frac"] = c.get("SQ_WAIT_INST_LDS", 0.0) / wc
    if c.get("SQ_INSTS_VALU") and launch_ms:
        # plain VALU issue time at the measured 1.13 ns per wave-instruction and SIMD (tools/probe/valu_forms.hip), 1024 SIMDs;
        # DPP / compare forms cost more: DESIGN.md section 7 puts the weighted figure at 0.70
        out["fp32_valu_issue_frac_of_launch_lower_bound"] = c["SQ_INSTS_VALU"] * 1.13e-6 / 1024.0 / launch_ms
    if c.get("SQ_LDS_IDX_ACTIVE") and c.get("GRBM_GUI_ACTIVE"):
        # (GRBM_GUI_ACTIVE sums the busy cycles of the 8 XCDs, SQ_LDS_IDX_ACTIVE the LDS-array cycles of the 256 CUs)
        out["lds_busy_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256.0)
        out["lds_bank_conflict_frac_of_lds_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    return out


PMC_SETS = {
    # separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: FETCH_SIZE and WRITE_SIZE
    # never share a pass with each other or with the SQ counters
    "FETCH_SIZE": ["FETCH_SIZE"],
    "WRITE_SIZE": ["WRITE_SIZE"],
    "SQ": ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS",
           "SQ_INSTS_LDS", "SQ_INSTS_SALU"],
}


def parse_pmc_csv(path, kernel_sub="k_win"):
    """{counter: mean per full-size launch} of the kernels whose name contains kernel_sub in one rocprofv3
    counter_collection.csv: launches of the largest grid only, and of those the ones at least half as long as the
    longer quarter's shortest (bench.py --profile also runs a 32-window parity leg through the same kernel)."""
    import collections
    import csv
    rows = [r for r in csv.DictReader(open(path)) if kernel_sub in r["Kernel_Name"]]
    if not rows:
        return {}
    gmax = max(int(r["Grid_Size"]) for r in rows)
    rows = [r for r in rows if int(r["Grid_Size"]) == gmax]
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: sum(v) / len(v) for k, v in acc.items()}
    out["launches_sampled"] = len(next(iter(acc.values())))
    return out


def measure_pmc_in_run(budget_s=150.0, steps=8, warmup=3):
    """HBM traffic and SQ counters of the headline kernel measured IN THIS RUN (the figures copied from
    profiles/*.json were a builder's claim): three child processes `rocprofv3 --pmc <one set> -- python3 bench.py
    --profile` (FETCH_SIZE, WRITE_SIZE and the SQ set each in a pass of their own, no trace domain beside them), started
    before this process has touched the GPU, each under a timeout.  Returns {"counters": {...per launch...},
    "hbm_bytes_per_launch": (2 FETCH_SIZE + WRITE_SIZE) KiB (gfx950 counts 64 B per 128-B request of a wide stream:
    MI355X_MICROARCH.md), ...} or {"error": why} -- the caller then falls back to the recorded files."""
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"error": "rocprofv3 not found"}
    t_start = time.perf_counter()
    tmp = tempfile.mkdtemp(prefix="rmx_pmc_", dir="/tmp")
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    env["RMX_BENCH_CHILD"] = "1"
    env["RMX_BENCH_PREWARM"] = "0"       # the children need launches to count, not a settled clock
    counters, notes = {}, []
    try:
        for name, cs in PMC_SETS.items():
            left = budget_s - (time.perf_counter() - t_start)
            if left < 20.0:
                notes.append(f"{name}: skipped (time budget)")
                continue
            d = os.path.join(tmp, name)
            cmd = [rp, "--pmc"] + cs + ["-f", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                                        "--profile", "--steps", str(steps), "--warmup", str(warmup)]
            try:
                r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=min(left, 90.0))
            except subprocess.TimeoutExpired:
                notes.append(f"{name}: timed out")
                continue
            if r.returncode != 0:
                notes.append(f"{name}: rocprofv3 exit {r.returncode}: {(r.stderr or r.stdout)[-200:]}")
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                notes.append(f"{name}: no counter_collection.csv")
                continue
            got = parse_pmc_csv(files[0])
            n = got.pop("launches_sampled", 0)
            counters.update(got)
            notes.append(f"{name}: {n} full-size launches")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {"counters": counters, "passes": notes, "seconds": time.perf_counter() - t_start,
           "method": "rocprofv3 --pmc, one counter set per child run of `bench.py --profile` (separate passes), per full-size launch of k_win"}
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        out["hbm_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
        out["note"] = "FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md: gfx950 counts 64 B per 128-B request; WRITE_SIZE (KiB) as counted"
    elif not counters:
        out["error"] = "; ".join(notes) or "no counters"
    return out


VALU_CLOCK_GHZ = 2.4        # MI355X_MICROARCH.md: max clock; a wave64 VALU instruction occupies its 32-wide SIMD for 2 cycles
N_SIMDS = 256 * 4


def recorded_valu(launch_ms, live=None):
    """`roofline.valu`: the fused kernel against the resource that actually binds it -- fp32 VALU issue.  Wave-instructions
    per launch measured in this run (`live`: measure_pmc_in_run) or, failing that, from profiles/pmc_latest.json (same
    digest rule as the traffic figure), their issue time at the data-sheet rate of 2 cycles per wave64 instruction on each
    of the 1024 SIMDs at 2.4 GHz, and that time over this run's launch."""
    n, src = None, None
    if live and live.get("counters", {}).get("SQ_INSTS_VALU"):
        n, src = live["counters"]["SQ_INSTS_VALU"], "measured in this run (rocprofv3 --pmc child pass, SQ_INSTS_VALU per launch of k_win)"
    else:
        pj = os.path.join(ROOT, "profiles", "pmc_latest.json")
        try:
            d = json.load(open(pj))
        except Exception:
            return None
        if d.get("source_digest") != source_digest():
            return {"stale": f"profiles/pmc_latest.json (tag {d.get('tag')}) was recorded on other sources; re-run tools/profile.sh"}
        n = d.get("counters", {}).get("SQ_INSTS_VALU")
        src = f"profiles/pmc_latest.json (tag {d.get('tag')}, SQ_INSTS_VALU per launch of {d.get('kernel')})"
    if not n or not launch_ms:
        return None
    issue_ms = n * 2.0 / (N_SIMDS * VALU_CLOCK_GHZ * 1e9) * 1e3
    return {"wave_instructions": n, "issue_ms_at_2cyc": issue_ms, "frac_of_launch": issue_ms / launch_ms,
            "clock_ghz_assumed": VALU_CLOCK_GHZ, "simds": N_SIMDS, "source": src}


class Shape:
    """One BASELINE shape on one device: synthetic windows resident in HBM, an engine, a step."""

    def __init__(self, torch, xcorr, name, dev, dev_index, windows, buoys=None, seed_offset=0):
        import numpy as np
        cfg = dict(CONFIGS[name])
        self.torch, self.name, self.cfg, self.dev = torch, name, cfg, dev
        self.B = buoys or cfg["B"]
        self.N, self.fs, self.C = cfg["N"], cfg["fs"], cfg["C"]
        self.W = windows                                # (window, channel) units on this device
        self.P = self.B * (self.B - 1) // 2
        self.caf = "doppler_hz" in cfg
        self.grid = None
        buoy_dop = None
        if self.caf:
            nb = int(round(cfg["doppler_hz"] / cfg["doppler_step_hz"]))
            self.grid = np.arange(-nb, nb + 1) * (cfg["doppler_step_hz"] / self.fs)       # cycles/sample, 21 bins
            rng = np.random.default_rng(cfg["seed"])
            buoy_dop = rng.integers(-8, 9, size=self.B) * (cfg["doppler_step_hz"] / self.fs) * 0.5   # true offsets within +-200 Hz
        self.D = len(self.grid) if self.caf else 1
        W, B, N, P = self.W, self.B, self.N, self.P
        self.x, self.delays = synth_on_device(torch, dev, W, B, N, self.fs, seed=cfg["seed"] + seed_offset, doppler_cps=buoy_dop)
        self.lag = torch.zeros((W, P), dtype=torch.int32, device=dev)
        self.frac = torch.zeros((W, P), dtype=torch.float32, device=dev)
        self.peak = torch.zeros((W, P), dtype=torch.float32, device=dev)
        self.dop = torch.zeros((W, P), dtype=torch.int32, device=dev) if self.caf else None
        self.eng = xcorr.XcorrEngine(B, N, max(W, 1), device=dev_index)
        self.stream = torch.cuda.current_stream()
        self.eng.set_stream(self.stream.cuda_stream)
        self.calls = 0        # full-size engine calls (tools/summarize_prof.py divides counter totals by it)

    def step(self):
        self.calls += 1
        if self.caf:
            self.eng.caf_device(self.x.data_ptr(), self.W, self.grid, self.dop.data_ptr(), self.lag.data_ptr(),
                                self.frac.data_ptr(), self.peak.data_ptr())
        else:
            self.eng.correlate_device(self.x.data_ptr(), self.W, self.lag.data_ptr(), self.frac.data_ptr(), self.peak.data_ptr())

    def host_windows(self, a, b):
        import numpy as np
        return self.x[a:b].cpu().numpy().view(np.complex64).reshape(b - a, self.B, self.N)

    def alg_bytes_per_step(self):
        # algorithmic bytes: 16 N + 12 per pair-window; CAF: 16 N per pair-window-bin + 16 per pair-window
        return self.W * self.P * ((16 * self.N) * self.D + (16 if self.caf else 12))

    def parity(self, cpu_thr, light=False):
        """the timed outputs (lag / frac / peak on the device) against the oracle on a bounded subset + ground truth"""
        import numpy as np
        from oracle import xcorr_ref as orc
        li, lf = self.lag.cpu().numpy(), self.frac.cpu().numpy()
        B, P, N, W = self.B, self.P, self.N, self.W
        pl = orc.pair_list(B)
        true_lag = self.delays[:, pl[:, 1]] - self.delays[:, pl[:, 0]]
        truth_ok = float(np.mean(np.abs(li + lf - true_lag) < 1.0))
        if self.caf:
            # oracle on a bounded subset: window 0, the first pairs, the whole Doppler grid (32 pairs = 672 calls of the
            # reference primitive at cfg5, about 30 s on the box's host cores; 4 pairs in the short form)
            npr = min(4 if light else 32, P)
            rd, ri, rf, rp = orc.caf_batch(self.host_windows(0, 1), self.grid, pl[:npr])
            dgot = self.dop.cpu().numpy()
            ref, got = ri + rf, li[:1, :npr] + lf[:1, :npr].astype(np.float64)
            return {"windows": 1, "pairs": npr, "doppler_bins": self.D,
                    "doppler_idx_mismatches": int(np.sum(dgot[:1, :npr] != rd)),
                    "lag_int_mismatches": int(np.sum(li[:1, :npr] != ri)),
                    "max_lag_err_rel": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))),
                    "lags_within_1_sample_of_truth": truth_ok}
        nchk = min(256 if N <= 4096 else (2 if light and N > 262144 else 8), W)   # >= 256 windows at cfg3; 8 of the long ones
        ri, rf, rp = orc.xcorr_batch_fast(self.host_windows(0, nchk), workers=cpu_thr)
        ref = ri + rf
        got = li[:nchk] + lf[:nchk].astype(np.float64)
        return {"windows": nchk, "pair_windows": int(nchk * P),
                "lag_int_mismatches": int(np.sum(li[:nchk] != ri)),
                "max_lag_err_rel": float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0))),
                "lags_within_1_sample_of_truth": truth_ok}

    def close(self):
        self.eng.close()
        self.x = self.lag = self.frac = self.peak = self.dop = None
        self.torch.cuda.empty_cache()


def timed_steps(torch, sh, steps, sync_all):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides -> (wall seconds, HIP-event ms on the launch stream)"""
    sync_all()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(sh.stream)
    for _ in range(steps):
        sh.step()
    ev1.record(sh.stream)
    sync_all()
    return time.perf_counter() - t0, ev0.elapsed_time(ev1)


def other_config(torch, xcorr, name, dev, dev_index, sync_all, cpu_thr):
    """One of the other BASELINE shapes in the same process (N = 1): a bounded number of steps after the shape's own
    warm-up, the same bracket, parity of the timed outputs on a short subset.  Returns the dict for `other_configs`."""
    cfg = CONFIGS[name]
    steps, warm = OTHER_STEPS[name]
    t_all = time.perf_counter()
    sh = Shape(torch, xcorr, name, dev, dev_index, cfg["W"] * cfg["C"])
    for _ in range(warm):
        sh.step()
    elapsed, region_ms = timed_steps(torch, sh, steps, sync_all)
    ms = elapsed * 1e3 / steps
    alg = sh.alg_bytes_per_step()
    # one more step with every launch bracketed by HIP events (not inside the timed region: two event records per launch
    # are a tenth of cfg1's whole step): ms per kernel family of this shape
    sh.eng.set_option("timing", 1)
    sh.step()
    by_kernel = sh.eng.last_timing_by_kernel()
    sh.eng.set_option("timing", 0)
    out = {"workload": cfg["what"], "steps": steps, "warmup": warm, "ms_per_step": ms, "kernel_ms": by_kernel,
           "value": sh.W * sh.P * sh.N * sh.D / (ms * 1e-3), "unit": "samples/s",
           "alg_bytes_per_step": alg,
           "roofline_frac": alg / (region_ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "whole_path_frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "ingest_samples_per_s": sh.W * sh.B * sh.N / (ms * 1e-3),
           "realtime_factor": sh.W * sh.B * sh.N / (ms * 1e-3) / (sh.B * sh.fs),
           "parity": sh.parity(cpu_thr, light=True)}
    tr, _src = recorded_traffic(name)
    out["traffic"] = tr
    sh.close()
    out["wall_s"] = time.perf_counter() - t_all
    return out


PROJECTION_GPUS = (1, 2, 4, 8)


def projection_table(ms_by_gpus, job_windows):
    """{G: {windows_per_gpu, ms_per_step, efficiency_vs_G1}} from one GPU's ms per step on job_windows / G windows.  The path
    has no collective (windows block-shard, host gather outside the timed region), so G GPUs finish the job when the
    slowest finishes its block: time(G) = ms(job / G); strong-scaling efficiency = time(1) / (G * time(G))."""
    t1 = ms_by_gpus[1]
    return {str(g): {"windows_per_gpu": -(-job_windows // g), "ms_per_step": ms,
                     "efficiency_vs_G1": t1 / (g * ms)} for g, ms in sorted(ms_by_gpus.items())}


def strong_scaling_projection(torch, sh, sync_all, steps=40, warm=5):
    """One GPU stands in for each rank of a G-GPU strong-scaling run of the headline job (no 8-GPU node is available to
    this build): the first ceil(W / G) resident windows through the same device-pointer call, `steps` timed steps after
    `warm` warm-up steps each, G = 1 first.  A PROJECTION: it contains the partial-round and launch-overhead cost of
    small blocks (512 windows = two rounds of the persistent grid), not PCIe / host contention between real ranks."""
    import time as _t
    from radio_mapper_amd.shard import window_shard
    ms = {}
    for g in PROJECTION_GPUS:
        _s, wg = window_shard(sh.W, 0, g)          # rank 0's block is the largest
        if wg < 1:
            continue
        call = lambda: sh.eng.correlate_device(sh.x.data_ptr(), wg, sh.lag.data_ptr(), sh.frac.data_ptr(), sh.peak.data_ptr())
        for _ in range(warm):
            call()
        sync_all()
        t0 = _t.perf_counter()
        for _ in range(steps):
            call()
        sync_all()
        ms[g] = (_t.perf_counter() - t0) * 1e3 / steps
    out = projection_table(ms, sh.W)
    return {"kind": "projection from ONE GPU (no multi-GPU hardware was available): G GPUs = this GPU on ceil(W / G) windows",
            "job_windows": sh.W, "steps": steps, "warmup": warm, "by_gpus": out}


def job_shard(config, scaling, windows, rank, n_gpus):
    """(first unit, units of this rank, units of the whole job); a unit = one (window, channel).
    weak: every rank owns the config's per-GPU batch (`windows` overrides the windows per GPU); strong: the job is the
    config's batch as BASELINE.json states it -- cfg3: 4096 windows; cfg4: 4096 windows x 10 channels; cfg5: 64 windows;
    `windows` overrides the total -- block-sharded over the ranks (remainders to the low ranks); at N = 1 one GPU owns it all."""
    from radio_mapper_amd.shard import window_shard
    cfg = CONFIGS[config]
    C = cfg["C"]
    if scaling == "strong":
        job = {"cfg4": 4096, "cfg5": 64}.get(config, cfg["W"])
        total = (windows or job) * C
        start, count = window_shard(total, rank, n_gpus)
        return start, count, total
    per = (windows or cfg["W"]) * C              # channels are a batch axis
    total = per * n_gpus
    start, count = window_shard(total, rank, n_gpus)
    assert count == per
    return start, count, total


# (timed steps, warm-up) per other shape.  cfg1's step is 45 us: 2000 warm-up steps = the 0.1 s of load the device needs to reach
# its sustained clock (behind 10 warm-up steps the same five launches read 47-50 us instead of 44-45)
OTHER_STEPS = {"cfg1": (200, 2000), "cfg2": (20, 5), "cfg4": (20, 5), "cfg5": (3, 1)}
# (a step of these shapes is 0.2 ... 0.6 ms: 300 warm-up steps = the 0.1 ... 0.2 s of load the device needs to reach its sustained
# clock -- behind 5 warm-up steps the same kernels measure 10 % slower; `warmup` is in each entry)
OTHER_STEPS.update({k: (200, 300) for k in CAPTURE_SHAPES})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank owns a full batch; strong: the config's windows are block-sharded over the ranks")
    ap.add_argument("--windows", type=int, default=None, help="windows per GPU (weak) / in total (strong); default: the config's")
    ap.add_argument("--buoys", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other four BASELINE shapes (profiling runs)")
    ap.add_argument("--no-single-group", action="store_true", help="skip the one-window latency probe (profiling runs)")
    ap.add_argument("--no-projection", action="store_true", help="skip the one-GPU strong-scaling projection (profiling runs)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure HBM traffic / SQ counters in this run (three rocprofv3 --pmc child runs, ~30 s); "
                         "the figures recorded under profiles/ are used instead when their source digest matches")
    ap.add_argument("--no-host-path", action="store_true", help="skip the host-pointer legs (profiling runs)")
    ap.add_argument("--profile", action="store_true",
                    help="the timed path and its parity only: --no-other-configs --no-projection --no-single-group --no-host-path --no-cpu-baseline")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for cpu_baseline")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous + shard + gather only (no compute, no metric): tests the N-rank launcher")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.profile:
        args.no_other_configs = args.no_projection = args.no_single_group = args.no_host_path = args.no_cpu_baseline = True
        args.no_pmc = True

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

    # counters of the headline kernel, measured by child processes BEFORE this process touches the GPU (N = 1, cfg3 only)
    live_pmc = None
    if (world == 1 and rank == 0 and args.config == "cfg3" and not args.no_pmc and not args.launch_check
            and os.environ.get("RMX_BENCH_CHILD") != "1" and args.buoys is None and args.windows is None):
        import __graft_entry__ as ge0
        ge0.build()                       # the children must not each compile the library
        try:
            live_pmc = measure_pmc_in_run()
        except Exception as e:            # never let the measurement take the line with it
            live_pmc = {"error": f"{type(e).__name__}: {e}"}

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
    # which binary runs: the digest compiled into the loaded library against the digest of the sources in this tree
    build_info = xcorr.build_info()
    src_digest = source_digest()
    if build_info.get("source_digest") != src_digest:
        print(f"bench.py: {xcorr.library_path()} was built from other sources (binary digest "
              f"{build_info.get('source_digest')} != source digest {src_digest}); rebuild with "
              f"`python -c 'import __graft_entry__ as g; g.build(force=True)'`", file=sys.stderr)
        sys.exit(3)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    cfg = dict(CONFIGS[args.config])
    C = cfg["C"]
    steps, warm = DEFAULT_STEPS[args.config]
    steps = args.steps if args.steps is not None else steps
    warm = args.warmup if args.warmup is not None else warm
    w_start, W, W_total = job_shard(args.config, args.scaling, args.windows, rank, n_gpus)
    # every rank generates its own block of the job's windows (seed per config, offset by rank)
    sh = Shape(torch, xcorr, args.config, dev, dev_index, W, buoys=args.buoys, seed_offset=7919 * rank)
    B, N, P, D, fs, caf = sh.B, sh.N, sh.P, sh.D, sh.fs, sh.caf
    eng, x, lag, frac, peak = sh.eng, sh.x, sh.lag, sh.frac, sh.peak
    step = sh.step

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            barrier()
            torch.cuda.synchronize()

    def max_over_ranks(v):
        if world == 1:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    region_order = []
    # the other BASELINE shapes, same process, N = 1 only (they are parity-test cases, not the metric: bounded steps).
    # They run BEFORE the headline region: whatever state they leave the device in is the state a service that handles
    # several shapes is in, and the headline region below is then still exactly W warm-up + K timed steps.
    others = None
    if rank == 0 and n_gpus == 1 and args.config == "cfg3" and not args.no_other_configs:
        others = {}
        for name in ("cfg1", "cfg2", "cfg4", "cfg5"):
            try:
                others[name] = other_config(torch, xcorr, name, dev, dev_index, sync_all, cpu_threads())
            except Exception as e:      # a shape that fails must not take the headline line with it
                others[name] = {"error": f"{type(e).__name__}: {e}"}
        region_order.append("other_configs (cfg1, cfg2, cfg4, cfg5: warm-up, timed steps, parity each)")
    capture = None
    if rank == 0 and n_gpus == 1 and args.config == "cfg3" and not args.no_other_configs:
        capture = {}
        for name in CAPTURE_SHAPES:
            try:
                capture[name] = other_config(torch, xcorr, name, dev, dev_index, sync_all, cpu_threads())
            except Exception as e:
                capture[name] = {"error": f"{type(e).__name__}: {e}"}
        region_order.append("reference_capture_lengths (N = 8192 and 16384, 3 and 8 buoys: warm-up, timed steps, parity each)")
    # one-GPU projection of the 1 -> 8 GPU strong-scaling curve: the path has no collective, so G GPUs on
    # the config's job = one GPU on 1/G of its windows; timed on the resident windows of the headline shape
    projection = None
    if rank == 0 and n_gpus == 1 and args.config == "cfg3" and not args.no_projection and not caf:
        projection = strong_scaling_projection(torch, sh, sync_all)
        region_order.append("strong_scaling_projection (cfg3 on W/G windows, G = 1, 2, 4, 8)")

    for _ in range(warm):
        step()
    # (a) HEADLINE: the driver's contract and nothing else -- W warm-up steps of this shape, then exactly K timed steps
    # bracketed by barrier + synchronize; HIP events on the launch stream (the ctx uses torch's current stream) around
    # the same region give the back-to-back launch duration of roofline.achieved
    elapsed, region_ms = timed_steps(torch, sh, steps, sync_all)
    elapsed = max_over_ranks(elapsed)
    ms_per_step = elapsed * 1e3 / steps
    region_order += [f"warmup ({warm} steps)", f"timed HEADLINE region ({steps} steps)"]
    # (b) untimed pre-warm, then the same K steps again: a device that has just been idle needs some hundred
    # milliseconds of load to settle on its sustained clock; "prewarm_steps" counts everything that ran between the
    # headline region and the sustained one
    prewarm = 0
    if args.config == "cfg3":
        prewarm = max(0, int(os.environ.get("RMX_BENCH_PREWARM", "300")) - warm - steps)
    for _ in range(prewarm):
        step()
    sus_elapsed, sus_region_ms = timed_steps(torch, sh, steps, sync_all)
    ms_per_step_sustained = max_over_ranks(sus_elapsed) * 1e3 / steps
    region_order += [f"prewarm ({prewarm} steps)", f"timed sustained region ({steps} steps)"]
    # kernel durations from HIP events bracketing every launch on the launch stream: a few more steps with option
    # "timing" on, a read-back after each (OUTSIDE the timed regions: two event records per launch are a tenth of a
    # cfg1 step, which is five launches of 6-22 us)
    fwd_ms = pair_ms = 0.0
    fwd_n = pair_n = 0
    n_meas = max(min(steps, 10 if not caf else 1), 1)
    by_kernel = None
    eng.set_option("timing", 1)
    for k in range(n_meas):
        step()
        tm = eng.last_timing()
        fwd_ms += tm["fwd_ms"]; fwd_n += tm["fwd_launches"]
        pair_ms += tm["pair_ms"]; pair_n += tm["pair_launches"]
        by_kernel = eng.last_timing_by_kernel()
    eng.set_option("timing", 0)
    seen = ranks_seen()

    # PCIe-inclusive rate of the same step through the host-pointer entry of the C ABI (reported, never `value`):
    # pageable numpy in, numpy out -- complex64 (8 B/sample over the link) and the reference's wire format, raw uint8
    # I/Q (2 B/sample; the synthetic windows sit on the rtl_sdr grid, so raw = x + 127.5 exactly and the lags must agree)
    host_ms = host_u8_ms = dev_u8_ms = None
    host_u8_same = None
    if rank == 0 and n_gpus == 1 and args.config == "cfg3" and not args.no_host_path:
        xh = x.cpu().numpy().view(np.complex64).reshape(W, B, N)
        eng.correlate(xh[:64])
        th = time.perf_counter()
        got_c = eng.correlate(xh)
        sh.calls += 1
        host_ms = (time.perf_counter() - th) * 1e3
        del xh
        raw = (x + 127.5).to(torch.uint8).cpu().numpy().reshape(W, B, 2 * N)
        eng.correlate(raw[:64])
        th = time.perf_counter()
        got_u = eng.correlate(raw)
        sh.calls += 1
        host_u8_ms = (time.perf_counter() - th) * 1e3
        host_u8_same = bool(all(np.array_equal(a, b) for a, b in zip(got_c, got_u)))
        # the same raw bytes resident in HBM (2 B/sample compulsory input instead of 8): device pointers in and out
        raw_d = torch.from_numpy(raw).to(dev)
        for _ in range(3):
            eng.correlate_device(raw_d.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr(), u8=True)
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(20):
            eng.correlate_device(raw_d.data_ptr(), W, lag.data_ptr(), frac.data_ptr(), peak.data_ptr(), u8=True)
        torch.cuda.synchronize()
        dev_u8_ms = (time.perf_counter() - th) * 1e3 / 20
        sh.calls += 23
        del raw, raw_d, got_c, got_u
        step()                    # leave the device-pointer results of the timed path in lag/frac/peak
        torch.cuda.synchronize()

    # host-side gather of the lag scalars (the only exchange of the multi-GPU path)
    li, lf, pk = lag.cpu().numpy(), frac.cpu().numpy(), peak.cpu().numpy()
    if world > 1 and backend == "nccl" and args.scaling == "weak":
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

    # parity in the same run (rank 0) + cpu baseline
    parity = None
    cpu = None
    single = None
    if rank == 0:
        parity = sh.parity(cpu_threads())
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(sh.host_windows, W, B, N, args.cpu_budget, doppler=sh.grid)
        # latency of ONE window of the same shape (the reference's call pattern: one frequency group per call), device
        # pointers, outside the timed region; extra information, not part of the metric
        if not caf and not args.no_single_group:
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
    calls_headline = sh.calls
    fused = (N == 4096 and not caf)
    alg_bytes_per_step_gpu = sh.alg_bytes_per_step()
    headline_workload = cfg["what"]
    sh.close()

    if rank == 0:
        units_per_step = W_total * P * N * D                  # IQ samples cross-correlated per step
        value = units_per_step / (ms_per_step * 1e-3)
        if fused:
            launches_per_step = max(pair_n / n_meas, 1.0)
            kernel = "k_win (fused forward + pair kernel), one launch per step"
            launch_ms = region_ms / (steps * launches_per_step)
            launch_ms_sustained = sus_region_ms / (steps * launches_per_step)
            isolated_launch_ms = pair_ms / max(pair_n, 1)
        else:
            # multi-kernel paths: the figure is for the whole kernel sequence of one step (HIP events on the
            # launch stream around the timed region); per-kernel durations are in profiles/r04_<cfg>_*
            launches_per_step = 1.0
            kernel = ("rmx_caf_batch kernel sequence (un-rotated spectra once; per Doppler bin: de-rotated forward "
                      "g_cols_fwd + g_rows, g_rows_anchor(product, inverse), g_cols_inv, g_final, k_caf_select)" if caf else
                      ("four-step sequence g_cols_fwd + g_rows_fused (forward rows, products, inverse rows) + g_cols_inv "
                       "+ g_final" if B <= 4 and W >= 4 else
                       "four-step sequence g_cols_fwd + g_rows + g_rows(product, inverse) + g_cols_inv + g_final"))
            launch_ms = region_ms / steps
            launch_ms_sustained = sus_region_ms / steps
            isolated_launch_ms = None
        alg_bytes_per_launch = alg_bytes_per_step_gpu / launches_per_step
        achieved = alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
        traffic, traffic_src = recorded_traffic(args.config)
        if live_pmc and live_pmc.get("hbm_bytes_per_launch"):
            traffic = live_pmc["hbm_bytes_per_launch"]
            traffic_src = (f"measured in this run: {live_pmc['method']}; {'; '.join(live_pmc['passes'])}; {live_pmc['note']} "
                           f"({live_pmc['seconds']:.0f} s)")
        elif live_pmc:
            traffic_src = (traffic_src or "no recorded figure") + f" [in-run measurement failed: {live_pmc.get('error') or live_pmc.get('passes')}]"
        ingest = W_total * B * N / (ms_per_step * 1e-3)       # input samples taken in per second, all ranks
        line = {
            "metric": "IQ samples cross-correlated per second" + (" (pair-window-Doppler-bin samples)" if caf else ""),
            "value": value, "unit": "samples/s", "n_gpus": n_gpus, "ranks_seen": seen, "steps": steps,
            "warmup": warm, "engine_calls": calls_headline, "ms_per_step": ms_per_step,
            "ms_per_step_sustained": ms_per_step_sustained, "prewarm_steps": prewarm,
            "region_order": region_order,
            "build": {"binary_digest": build_info.get("source_digest"), "source_digest": src_digest,
                      "library": os.path.relpath(xcorr.library_path(), ROOT), "info": build_info.get("text")},
            "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": headline_workload, "name": args.config,
                       "n_buoys": B, "n_pairs": P, "n_samples": N, "windows_per_gpu": W, "channels": C,
                       "doppler_bins": D if caf else None,
                       "windows_total": W_total, "parallelism": f"windows sharded x{n_gpus}, no collective"},
            # `bound` is the roofline north_star declares for the path and the one `frac` is taken against; `limiter` is the
            # resource the counters say the dominant kernel actually waits on
            "roofline": {"bound": "hbm", "limiter": "fp32-valu-issue" if fused else "hbm (two passes per transform) + CU-side issue",
                         "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": alg_bytes_per_launch,
                         "launch_ms": launch_ms, "launch_ms_sustained": launch_ms_sustained,
                         "frac_sustained": alg_bytes_per_launch / (launch_ms_sustained * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "isolated_launch_ms": isolated_launch_ms,
                         "kernel_ms_per_step": by_kernel,
                         "fwd_kernel_ms_per_step": fwd_ms / n_meas,
                         "pair_kernel_ms_per_step": pair_ms / n_meas,
                         "whole_path_frac": alg_bytes_per_step_gpu / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "whole_path_frac_sustained": alg_bytes_per_step_gpu / (ms_per_step_sustained * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "valu": recorded_valu(launch_ms, live_pmc) if fused else None,
                         "counters": recorded_counters(launch_ms, live_pmc) if fused else None,
                         "fp32_valu_and_lds": "DESIGN.md section 7: the fused kernel is bound by fp32 VALU issue (70 % of the launch "
                                              "with DPP / compare forms weighted), not by HBM"},
            "ingest_samples_per_s": ingest,
            "realtime_factor": ingest / (B * fs),               # seconds of every buoy's signal processed per second
            "host_path_ms_per_step": host_ms,
            "host_path_u8_ms_per_step": host_u8_ms,
            "host_path_u8_identical": host_u8_same,
            "device_u8_ms_per_step": dev_u8_ms,                 # raw uint8 I/Q resident in HBM (RMX_IN_U8 | RMX_IN_DEVICE)
            "single_group": single,
            "strong_scaling_projection": projection,
            "other_configs": others,
            "reference_capture_lengths": capture,
            "cpu_baseline": cpu,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
