#!/usr/bin/env python3
"""Parity soak on the GPU box: random shapes, seeds, input types and pair lists through the C ABI against the oracle.

    python tests/soak_parity.py [--seconds 240] [--seed 0] > gpurun_out/soak.txt

Every case draws (N, B, W, input type, default or custom pair list, noise level) from the ranges the kernels dispatch
on -- the fused N = 4096 kernel and its custom-pair-list kernels, the whole-window kernels (N = 256 ... 16 384), the
LDS two-kernel path, the four-step path -- runs `XcorrEngine.correlate`, and compares with `oracle.xcorr_batch_fast`
(the oracle is the checker here, never the thing measured): integer lag bit-exact, lag within 1e-5 * max(|lag|, 1),
peak within 1e-5 relative.  An integer mismatch is excused only as in tests/test_gpu_parity.py: the oracle's two largest
magnitudes are within 1e-5 relative AND the GPU picked the oracle's second candidate.  A lag beyond 1e-5 is excused
only where the parabola is that ill-conditioned by itself: within four times what ONE float32 ulp on each of the oracle's
own three taps moves its result (a flat peak on a short, noisy window).  Prints one line per case and a summary; exit
code 1 on any unexcused mismatch.  Progress goes to stdout at least every few seconds.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TOL = 1e-5


def soak_detect(args, rng, t_end, xcorr):
    """rmx_detect_batch against oracle.detect_ref (the reference's own scipy calls) with the rule of tests/test_detect.py:
    identical peak sets, or a near-tie (< 1e-3 dB) somewhere in the window that explains the difference; noise floor,
    power and snr within 4e-4 dB on identical sets."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from oracle import detect_ref as dr
    from test_detect import make_windows, _margin
    n_win = n_exact = n_tie = n_bad = n_dyn = n_dyn_floor = 0
    worst = 0.0
    case = 0
    with xcorr.XcorrEngine(2, 4096, 1) as eng:
        while time.time() < t_end:
            case += 1
            N = 1 << int(rng.choice([6, 8, 9, 10, 11, 12, 13, 14]))
            W = int(rng.choice([1, 2, 5, 16, 40]))
            u8 = bool(rng.integers(0, 2))
            tones = int(rng.integers(0, 7))
            dist = int(rng.choice([10, 10, 3, 25]))
            seed = int(rng.integers(1, 2 ** 31 - 1))
            x, raw = make_windows(W, N, seed=seed, tones=tones, u8=u8)
            dc = 10e3 * N / 2.4e6
            ref = dr.detect_batch(x, dc_exclude_bins=dc, distance=dist)
            got = eng.detect(raw if u8 else x, dc_exclude_bins=dc, distance=dist, max_peaks=max(N // 4, 16))
            exact = tie = bad = 0
            for w in range(W):
                rb, rp, rs, rc, rf = ref[w]
                gb, gp, gs, gc, gf = got[w]
                ok_floor = abs(gf - rf) < 2e-4
                if not ok_floor:
                    # the median of the ORACLE's float32 spectrum is itself that far from a float64 one when a tone 60 dB
                    # above the floor sets the transform's error (case 3761 of the round-5 soak: oracle 59.688046, float64
                    # 59.687810, kernel 59.687817): same allowance as for the peak powers below
                    f64 = float(np.median(20.0 * np.log10(np.abs(np.fft.fft(x[w].astype(np.complex128))) + 1e-12)))
                    ok_floor = abs(gf - rf) <= 4.0 * abs(float(rf) - f64) + 1e-5
                    n_dyn_floor += int(ok_floor)
                if np.array_equal(gb, rb):
                    d = max(float(np.abs(gp - rp).max()) if len(rb) else 0.0, float(np.abs(gs - rs).max()) if len(rb) else 0.0)
                    worst = max(worst, d)
                    if ok_floor and d < 4e-4:
                        exact += 1
                    elif ok_floor and len(rb):
                        # a weak peak next to tones 50 dB stronger: the ORACLE's float32 spectrum is itself that far from a
                        # float64 one there (its transform's error rides on the strongest bin); allow four times that
                        p32 = dr.power_spectrum_db(x[w]).astype(np.float64)
                        p64 = 20.0 * np.log10(np.abs(np.fft.fft(x[w].astype(np.complex128))) + 1e-12)
                        own = float(np.abs(p32[rb] - p64[rb]).max())
                        if d <= 4.0 * own + 1e-5:
                            exact += 1
                            n_dyn += 1
                        else:
                            bad += 1
                    else:
                        bad += 1
                elif ok_floor and _margin(dr.power_spectrum_db(x[w]), rf, dist, 0.3) < 1e-3:
                    tie += 1
                else:
                    bad += 1
            n_win += W; n_exact += exact; n_tie += tie; n_bad += bad
            print(f"detect case {case:4d}  N={N:6d} W={W:3d} {'u8 ' if u8 else 'c64'} tones={tones} distance={dist:2d} seed={seed:10d}  "
                  f"identical={exact} near-tie={tie} bad={bad}{'  FAIL' if bad else ''}", flush=True)
    print(f"SUMMARY (detect): {case} cases, {n_win} windows, {n_exact} identical peak sets, {n_tie} explained by a near-tie (< 1e-3 dB), "
          f"{n_bad} failures, worst dB difference on identical sets {worst:.2e} ({n_dyn} windows beyond 4e-4 dB but within four times the "
          f"oracle's own float32 error against a float64 spectrum at those bins; {n_dyn_floor} noise floors beyond 2e-4 dB but within four "
          f"times the oracle's own distance to the float64 median)")
    return 1 if n_bad else 0


def soak_caf(args, rng, t_end, rm, xcorr, orc):
    """rmx_caf_batch against oracle.caf_batch (a Python loop over windows x pairs x hypotheses: small cases only).  A
    (Doppler index, lag) that differs from the oracle's is excused only when the oracle's own peak at the GPU's
    hypothesis is within 1e-5 relative of its best (two float32 transforms may order two such rows differently)."""
    n_cases = n_pw = n_bad = n_excused = 0
    worst_lag = worst_peak = 0.0
    case = 0
    while time.time() < t_end:
        case += 1
        logn = int(rng.choice([6, 8, 9, 10, 11, 12, 12, 13, 14, 15, 16]))
        N = 1 << logn
        B = int(rng.choice([2, 3, 4, 5, 8]))
        P = B * (B - 1) // 2
        D = int(rng.choice([1, 3, 5, 9, 21]))
        W = int(max(1, min(rng.choice([1, 2, 5, 16]), 4.0e7 // (P * D * N))))      # oracle cost ~ W*P*D transforms of 2N
        u8 = bool(rng.integers(0, 2))
        fs = float(rng.choice([2.4e6, 20e6]))
        step = float(rng.choice([25.0, 50.0, 200.0])) / fs
        dop = (np.arange(D) - D // 2) * step
        true = rng.uniform(-(D // 2) * step, (D // 2) * step, size=(W, B)) if D > 1 else np.zeros((W, B))
        seed = int(rng.integers(1, 2 ** 31 - 1))
        out = rm.synth.make_windows(W, B, N, fs, seed=seed, snr_db=float(rng.choice([10.0, 3.0])), return_u8=u8, doppler_cps=true)
        iq, raw = (out[0], out[2]) if u8 else (out[0], None)
        eng = xcorr.XcorrEngine(B, N, W)
        try:
            gd, li, lf, pk = eng.caf(raw if u8 else iq, dop)
        finally:
            eng.close()
        rd, ri, rf, rp = orc.caf_batch(iq, dop)
        bad = (gd != rd) | (li != ri)
        excused = np.zeros_like(bad)
        plist = orc.pair_list(B)
        for w, q in zip(*np.nonzero(bad)):
            i, j = int(plist[q, 0]), int(plist[q, 1])
            y = (iq[w, j] * orc.doppler_phasor(dop[gd[w, q]], N)).astype(np.complex64)
            _, _, alt = orc.xcorr_pair(iq[w, i], y)
            excused[w, q] = abs(float(alt) - float(rp[w, q])) <= TOL * float(rp[w, q])
        ok = ~bad
        ref = ri + rf
        got = li + lf.astype(np.float64)
        lag_err = np.max(np.abs(got - ref)[ok] / np.maximum(np.abs(ref[ok]), 1.0)) if ok.any() else 0.0
        peak_err = np.max(np.abs(pk[ok] - rp[ok]) / np.maximum(np.abs(rp[ok]), 1e-30)) if ok.any() else 0.0
        unexcused = int(np.sum(bad & ~excused))
        fail = unexcused + int(lag_err > TOL) + int(peak_err > TOL)
        n_cases += 1
        n_pw += li.size * D
        n_bad += fail
        n_excused += int(excused.sum())
        worst_lag = max(worst_lag, float(lag_err))
        worst_peak = max(worst_peak, float(peak_err))
        print(f"caf case {case:4d}  N={N:6d} B={B} W={W:2d} D={D:2d} {'u8 ' if u8 else 'c64'} seed={seed:10d}  pair-window-bins={li.size * D:6d} "
              f"mismatch={int(bad.sum())} (excused {int(excused.sum())})  lag_err={lag_err:.2e} peak_err={peak_err:.2e}{'  FAIL' if fail else ''}", flush=True)
    print(f"SUMMARY (caf): {n_cases} cases, {n_pw} pair-window-bins, {n_bad} failures, {n_excused} excused near-ties between rows, "
          f"worst relative lag error {worst_lag:.2e}, worst relative peak error {worst_peak:.2e}")
    return 1 if n_bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-bytes", type=float, default=3.0e8, help="largest input batch in bytes (complex64)")
    ap.add_argument("--device-pointers", action="store_true", help="half of the cases through RMX_IN_DEVICE | RMX_OUT_DEVICE (needs torch)")
    ap.add_argument("--caf", action="store_true", help="soak rmx_caf_batch (Doppler grid) instead of rmx_xcorr_batch")
    ap.add_argument("--detect", action="store_true", help="soak rmx_detect_batch (spectral detection) against oracle/detect_ref.py")
    ap.add_argument("--lengths", type=int, nargs="*", default=None, help="xcorr soak: only these log2 window lengths (e.g. 14 = 16384)")
    args = ap.parse_args()

    import __graft_entry__ as g
    g.build()
    import radio_mapper_amd as rm
    from radio_mapper_amd import xcorr
    from oracle import xcorr_ref as orc

    assert xcorr.device_count() > 0, "no GPU visible"
    rng = np.random.default_rng(args.seed)
    t_end = time.time() + args.seconds
    if args.caf:
        return soak_caf(args, rng, t_end, rm, xcorr, orc)
    if args.detect:
        return soak_detect(args, rng, t_end, xcorr)
    n_cases = n_pw = n_excused = n_bad = n_ill = 0
    worst_lag = worst_peak = 0.0
    by_n = {}
    case = 0
    while time.time() < t_end:
        case += 1
        # window length: weight the lengths with dedicated kernels
        logn = int(rng.choice(args.lengths or [4, 5, 6, 7, 8, 9, 10, 11, 12, 12, 12, 13, 13, 14, 14, 15, 16, 17, 18]))
        N = 1 << logn
        B = int(rng.choice([2, 3, 3, 4, 5, 6, 8, 8, 9, 12, 16])) if logn <= 14 else int(rng.choice([2, 3, 4, 6, 8]))
        per_win = B * N * 8
        w_max = max(1, int(args.max_bytes // per_win))
        W = int(min(w_max, rng.choice([1, 2, 3, 7, 16, 61, 256, 300, 1024] + ([257, 300, 517, 600, 1100] if N == 4096 else []) +
                                          ([9, 24, 33, 40, 257, 264, 520] if N == 16384 else []))))   # (N = 16384: flat / XCD-aware item order, several chunks)
        u8 = bool(rng.integers(0, 2))
        snr = float(rng.choice([10.0, 10.0, 3.0, 0.0, 20.0]))
        fs = float(rng.choice([2.4e6, 10e6, 20e6]))
        P_all = B * (B - 1) // 2
        custom = bool(rng.integers(0, 3) == 0)
        pairs = None
        if custom:
            k = int(rng.integers(1, P_all + 3))
            ij = rng.integers(0, B, size=(k, 2))
            ij[ij[:, 0] == ij[:, 1], 1] = (ij[ij[:, 0] == ij[:, 1], 0] + 1) % B      # i != j, any order, repeats allowed
            pairs = np.ascontiguousarray(ij, dtype=np.int32)
        seed = int(rng.integers(1, 2 ** 31 - 1))
        out = rm.synth.make_windows(W, B, N, fs, seed=seed, snr_db=snr, return_u8=u8)
        iq, raw = (out[0], out[2]) if u8 else (out[0], None)
        # (round 4) N = 4096: a third of the cases run with a small or odd chunk size, and half of the default-pair-list
        # cases through device pointers -- the combination that takes the per-chunk partial-round split of rmx_xcorr_batch
        chunk = 0
        if N == 4096 and rng.integers(0, 3) == 0:
            chunk = int(rng.choice([8, 16, 104, 256, 304, 1000]))
            xcorr.set_default_option("chunk_windows", chunk)
        # (round 5) N = 16384: a third of the cases with the quarter-transform kernels forced whatever the batch (kwin16k = 2:
        # the flat and the XCD-aware item order on small batches) or switched off (0: g_win_eo15 / four-step, the split dispatch)
        if N == 16384 and rng.integers(0, 3) == 0:
            chunk = -1
            xcorr.set_default_option("kwin16k", int(rng.choice([0, 2])))
        dev_ptr = args.device_pointers and bool(rng.integers(0, 2))
        t0 = time.time()
        eng = xcorr.XcorrEngine(B, N, W)
        try:
            if dev_ptr:
                import torch
                dev = torch.device("cuda", 0)
                src = raw if u8 else iq.view(np.float32)
                xd = torch.from_numpy(np.ascontiguousarray(src)).to(dev)
                Pn = B * (B - 1) // 2 if pairs is None else len(pairs)
                lagd = torch.zeros((W, Pn), dtype=torch.int32, device=dev)
                fracd = torch.zeros((W, Pn), dtype=torch.float32, device=dev)
                peakd = torch.zeros((W, Pn), dtype=torch.float32, device=dev)
                eng.set_stream(torch.cuda.current_stream().cuda_stream)
                eng.correlate_device(xd.data_ptr(), W, lagd.data_ptr(), fracd.data_ptr(), peakd.data_ptr(), pairs, u8=u8)
                torch.cuda.synchronize()
                li, lf, pk = lagd.cpu().numpy(), fracd.cpu().numpy(), peakd.cpu().numpy()
                del xd, lagd, fracd, peakd
            else:
                li, lf, pk = eng.correlate(raw if u8 else iq, pairs)
        finally:
            eng.close()
            if chunk:
                xcorr.clear_default_options()
        t_gpu = time.time() - t0
        ri, rf, rp = orc.xcorr_batch_fast(iq, pairs, workers=8)
        bad = li != ri
        excused = np.zeros_like(bad)
        if bad.any():
            plist = orc.pair_list(B) if pairs is None else pairs
            for w, q in zip(*np.nonzero(bad)):
                i, j = int(plist[q, 0]), int(plist[q, 1])
                # (the comparison above is against xcorr_batch_fast, whose own float32 transforms break an exact tie of uint8 data
                # either way -- case 405 of the round-5 soak, seed 1001: N = 16, lags 4 and 10 tie exactly, scipy.signal.correlate
                # and the engine say 4, the fast oracle 10 -- so the engine may hold either candidate, as in the suite's rule)
                margin, first, second = orc.peak_top2(iq[w, i], iq[w, j])
                excused[w, q] = margin <= TOL and li[w, q] in (first, second)
        ok = ~bad
        ref = ri + rf
        got = li + lf.astype(np.float64)
        rel = np.where(ok, np.abs(got - ref) / np.maximum(np.abs(ref), 1.0), 0.0)
        lag_err = float(rel.max()) if ok.any() else 0.0
        peak_err = np.max(np.abs(pk[ok] - rp[ok]) / np.maximum(np.abs(rp[ok]), 1e-30)) if ok.any() else 0.0
        unexcused = int(np.sum(bad & ~excused))
        # a lag beyond 1e-5 is a failure unless the parabola itself is that ill-conditioned: one float32 ulp on the
        # oracle's own taps moves the result by `bound`; four of those are allowed (two roundings per transform)
        ill = 0
        lag_fail = 0
        if lag_err > TOL:
            plist = orc.pair_list(B) if pairs is None else pairs
            for w, q in zip(*np.nonzero(rel > TOL)):
                bound = orc.parabola_ulp_bound(iq[w, int(plist[q, 0])], iq[w, int(plist[q, 1])])
                if rel[w, q] <= 4.0 * bound:
                    ill += 1
                else:
                    lag_fail += 1
        n_ill += ill
        tol_fail = lag_fail + int(peak_err > TOL)
        n_cases += 1
        n_pw += li.size
        n_excused += int(excused.sum())
        n_bad += unexcused + tol_fail
        worst_lag = max(worst_lag, float(lag_err))
        worst_peak = max(worst_peak, float(peak_err))
        by_n[N] = by_n.get(N, 0) + li.size
        print(f"case {case:4d}  N={N:7d} B={B:2d} W={W:5d} {'u8 ' if u8 else 'c64'} snr={snr:4.1f} "
              f"pairs={'all' if pairs is None else len(pairs):>4} {'dev' if dev_ptr else 'hst'} chunk={chunk:4d} seed={seed:10d}  pair-windows={li.size:7d} "
              f"int-mismatch={int(bad.sum())} (excused {int(excused.sum())})  lag_err={lag_err:.2e} peak_err={peak_err:.2e} "
              f"gpu={t_gpu * 1e3:7.1f} ms{'  (%d flat-peak lags within 4 ulp of the taps)' % ill if ill else ''}{'  FAIL' if unexcused or tol_fail else ''}", flush=True)
    print(f"SUMMARY: {n_cases} cases, {n_pw} pair-windows, {n_bad} failures, {n_excused} excused near-ties, {n_ill} flat-peak lags beyond 1e-5 but within 4 ulp of the oracle's taps, "
          f"worst relative lag error {worst_lag:.2e}, worst relative peak error {worst_peak:.2e}")
    print("pair-windows per window length: " + ", ".join(f"{n}: {c}" for n, c in sorted(by_n.items())))
    return 1 if n_bad else 0


if __name__ == "__main__":
    sys.exit(main())
