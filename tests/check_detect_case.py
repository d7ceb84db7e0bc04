"""GPU box helper: replay one case of `tests/soak_parity.py --detect` and print, for every window whose verdict is not
"identical", what differs: the two peak sets, the oracle's float32 and a float64 spectrum around the differing bins,
the noise floors and the near-tie margin of tests/test_detect._margin.

    python tests/check_detect_case.py N W u8(0|1) tones distance seed
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    N, W, u8, tones, dist, seed = (int(a) for a in sys.argv[1:7])
    import __graft_entry__ as g
    g.build()
    from radio_mapper_amd import xcorr
    from oracle import detect_ref as dr
    from test_detect import make_windows, _margin
    x, raw = make_windows(W, N, seed=seed, tones=tones, u8=bool(u8))
    dc = 10e3 * N / 2.4e6
    ref = dr.detect_batch(x, dc_exclude_bins=dc, distance=dist)
    with xcorr.XcorrEngine(2, 4096, 1) as eng:
        got = eng.detect(raw if u8 else x, dc_exclude_bins=dc, distance=dist, max_peaks=max(N // 4, 16))
    for w in range(W):
        rb, rp, rs, rc, rf = ref[w]
        gb, gp, gs, gc, gf = got[w]
        same = np.array_equal(gb, rb)
        d = 0.0
        if same and len(rb):
            d = max(float(np.abs(gp - rp).max()), float(np.abs(gs - rs).max()))
        if same and abs(gf - rf) < 2e-4 and d < 4e-4:
            continue
        p32 = dr.power_spectrum_db(x[w]).astype(np.float64)
        p64 = 20.0 * np.log10(np.abs(np.fft.fft(x[w].astype(np.complex128))) + 1e-12)
        print(f"window {w}: same_set={same} n_ref={len(rb)} n_gpu={len(gb)} floor ref={rf:.6f} gpu={gf:.6f} "
              f"(f64 median {np.median(p64):.6f}) d={d:.3e} margin={_margin(dr.power_spectrum_db(x[w]), rf, dist, 0.3):.3e}")
        if same:
            i = int(np.argmax(np.abs(gp - rp)))
            print(f"   worst power bin {rb[i]}: ref {rp[i]:.6f} gpu {gp[i]:.6f} f64 {p64[rb[i]]:.6f}; "
                  f"strongest bin {p64.max():.3f} dB")
            continue
        only_ref = np.setdiff1d(rb, gb)
        only_gpu = np.setdiff1d(gb, rb)
        print(f"   only oracle: {only_ref.tolist()}   only gpu: {only_gpu.tolist()}")
        for k in sorted(set(only_ref.tolist()) | set(only_gpu.tolist())):
            lo, hi = max(0, k - dist - 1), min(N, k + dist + 2)
            print(f"   around bin {k} (conf32 {(p32[k] - rf) / 20:.6f}):")
            for j in range(lo, hi):
                tag = ("R" if j in rb else " ") + ("G" if j in gb else " ")
                print(f"      {j:6d} {tag} p32 {p32[j]:12.6f}  p64 {p64[j]:12.6f}")


if __name__ == "__main__":
    main()
