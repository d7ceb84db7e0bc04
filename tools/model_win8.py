#!/usr/bin/env python3
"""numpy model of the k_win8 dataflow (1024 threads x 8 complex points, radix-8 x 4 passes).

Not part of the product or the tests' oracle: a design check that runs on the CPU.  It executes the
exact (thread, slot) index maps, twiddle tables and LDS address maps the HIP kernel uses and checks

  * forward + product + inverse + lane-pair radix-2 reproduce numpy's zero-padded FFT correlation;
  * every LDS address map is injective inside its region and free of bank conflicts under the
    ds_write_b64 (16 contiguous lanes, 8-byte bank pairs mod 16) and ds_read_b64 (32-lane halves,
    bank pairs mod 32) rules of MI355X_MICROARCH.md;
  * the 'full'-order index of every (thread, slot).

Decomposition (M = 4096 = 8^4, L = 2M; sub-transform p = bin parity):
    time   n = 512 n3 + 64 n2 + 8 n1 + n0          bin  k = c0 + 8 c1 + 64 d0 + 512 d1   (L-bin 2k + p)
    role 1  T = 2 t1 + p, t1 = n0 + 8 n1 + 64 n2    slots n3 <-> c0     (time side; parity on lane bit 0)
    role 2  T = 512 p + 64 c0 + 8 n1 + n0           slots n2 <-> c1
    role 3  T = 512 p + 64 c0 + 8 c1 + n0           slots n1 <-> d0
    role 4  T = 512 p + 64 c0 + 8 c1 + d0           slots n0 <-> d1     (frequency side)
  exchange 1<->2 crosses waves (workgroup barrier); 2<->3 and 3<->4 stay inside one wave.
"""
import numpy as np

M, L, NT, S = 4096, 8192, 1024, 8
REG = 576                      # complex per wave region of an exchange image


def W(n, e):
    return np.exp(-2j * np.pi * (np.asarray(e) % n) / n)


# ---- tables -------------------------------------------------------------------------------------
def tables():
    T = np.arange(NT)
    t1, p = T >> 1, T & 1
    c = np.arange(S)
    tw1 = W(M, t1[:, None] * c[None, :]) * np.where(p[:, None] == 1, W(L, t1)[:, None], 1.0)   # [T][c0]
    lane = np.arange(64)
    tb = W(512, lane[:, None] * c[None, :])        # [lane = n0 + 8 n1][c1]
    tc = W(64, c[:, None] * c[None, :])            # [n0][d0]
    w16 = W(16, c)                                 # odd parity: W_L^(512 n3) = W16^n3
    return tw1, tb, tc, w16


def dft8(v):                                       # over the slot axis (last), natural in / natural out
    k = np.arange(S)
    return v @ W(8, k[:, None] * k[None, :])


# ---- LDS address maps (complex index inside one exchange image) ---------------------------------
def a43(wave, c1, n0, d0):   # role 4 (lane d0 + 8 c1, slot n0)  <->  role 3 (lane n0 + 8 c1, slot d0)
    return wave * REG + c1 * 72 + n0 * 9 + d0


def a32(wave, c1, n1, n0):   # role 3 (lane n0 + 8 c1, slot n1)  <->  role 2 (lane n0 + 8 n1, slot c1)
    return wave * REG + n1 * 72 + c1 * 8 + n0


def a21(p, c0, n2, n1, n0):  # role 2 (wave 8p + c0, lane n0 + 8 n1, slot n2)  <->  role 1 (T = 2 t1 + p, slot c0)
    # (the 16 p skew separates the two parities of a role-1 half wave on reads, the xor on writes)
    return (8 * p + c0) * REG + 16 * p + ((64 * n2 + 8 * n1 + n0) ^ (8 * p))


def check_banks():
    lane = np.arange(64)
    lo, hi = lane & 7, lane >> 3
    ok = True

    def wr(addr, what):      # ds_write_b64: 4 groups of 16 contiguous lanes, 16 bank pairs
        nonlocal ok
        for g in range(4):
            a = addr[16 * g:16 * g + 16] % 16
            if len(set(a.tolist())) != 16:
                ok = False
                print("WRITE conflict", what, g)

    def rd(addr, what):      # ds_read_b64: 2 halves of 32 lanes, 32 bank pairs
        nonlocal ok
        for g in range(2):
            a = addr[32 * g:32 * g + 32] % 32
            if len(set(a.tolist())) != 32:
                ok = False
                print("READ conflict", what, g)

    for s in range(S):
        wr(a43(3, hi, s, lo), "43 write slot n0")       # role 4 writes slot n0 = s
        rd(a43(3, hi, lo, s), "43 read slot d0")        # role 3 reads slot d0 = s
        wr(a43(3, hi, lo, s), "43 write (forward) slot d0")
        rd(a43(3, hi, s, lo), "43 read (forward) slot n0")
        wr(a32(5, hi, s, lo), "32 write slot n1")       # role 3 writes slot n1 = s
        rd(a32(5, s, hi, lo), "32 read slot c1")        # role 2 reads slot c1 = s
        wr(a32(5, s, hi, lo), "32 write (forward)")
        rd(a32(5, hi, s, lo), "32 read (forward)")
        for p in range(2):
            wr(a21(p, 2, s, hi, lo), "21 write slot n2")   # role 2 writes slot n2 = s
        for wv in range(16):                                # role 1 wave wv reads slot c0 = s
            T = 64 * wv + lane
            t1, p = T >> 1, T & 1
            rd(a21(p, s, t1 >> 6, (t1 >> 3) & 7, t1 & 7), "21 read slot c0")
            wr(a21(p, s, t1 >> 6, (t1 >> 3) & 7, t1 & 7), "21 write (forward) slot c0")
        for p in range(2):
            rd(a21(p, 2, s, hi, lo), "21 read (forward) slot n2")
    # injectivity + own-region property
    seen = set()
    for p in range(2):
        for c0 in range(8):
            for n2 in range(8):
                for n1 in range(8):
                    for n0 in range(8):
                        a = a21(p, c0, n2, n1, n0)
                        assert a // REG == 8 * p + c0 and a not in seen
                        seen.add(a)
    for f in (a43, a32):
        seen = set()
        for x in range(8):
            for y in range(8):
                for z in range(8):
                    a = f(0, x, y, z)
                    assert 0 <= a < REG and a not in seen
                    seen.add(a)
    return ok


# ---- the transform network ------------------------------------------------------------------------
def role4_index():
    """bin (sub-transform index k) held by thread T (team layout) in slot d1, and its parity."""
    T = np.arange(NT)
    p, s = T >> 9, T & 511
    c0, c1, d0 = s >> 6, (s >> 3) & 7, s & 7
    d1 = np.arange(S)
    k = c0[:, None] + 8 * c1[:, None] + 64 * d0[:, None] + 512 * d1[None, :]
    return p, k


def forward(x, tab):
    """x: complex[M] window -> spectrum in role-4 register layout [T][d1] (L-bin 2k + p)."""
    tw1, tb, tc, w16 = tab
    T = np.arange(NT)
    # role 1: T = 2 t1 + p, slot n3
    t1, p1 = T >> 1, T & 1
    v = x[512 * np.arange(S)[None, :] + t1[:, None]].astype(complex)
    v = np.where(p1[:, None] == 1, v * w16[None, :], v)          # odd: x * W_L^(512 n3); W_L^t1 rides on tw1
    v = dft8(v) * tw1                                            # slots c0
    img = np.zeros(16 * REG, complex)
    img[a21(p1[:, None], np.arange(S)[None, :], (t1 >> 6)[:, None], ((t1 >> 3) & 7)[:, None], (t1 & 7)[:, None])] = v
    # role 2: T = 512 p + 64 c0 + 8 n1 + n0, slot n2
    p, s = T >> 9, T & 511
    wave, lane = T >> 6, T & 63
    c0, hi, lo = s >> 6, (s >> 3) & 7, s & 7
    v = img[a21(p[:, None], c0[:, None], np.arange(S)[None, :], hi[:, None], lo[:, None])]
    v = dft8(v) * tb[lane]                                       # slots c1, post-twiddle W_512^(c1 (8 n1 + n0))
    img2 = np.zeros(16 * REG, complex)
    img2[a32(wave[:, None], np.arange(S)[None, :], hi[:, None], lo[:, None])] = v        # (c1 = slot, n1 = hi, n0 = lo)
    # role 3: lane = n0 + 8 c1, slot n1
    v = img2[a32(wave[:, None], hi[:, None], np.arange(S)[None, :], lo[:, None])]
    v = dft8(v) * tc[lo]                                         # slots d0, post-twiddle W_64^(d0 n0)
    img3 = np.zeros(16 * REG, complex)
    img3[a43(wave[:, None], hi[:, None], lo[:, None], np.arange(S)[None, :])] = v        # (c1 = hi, n0 = lo, d0 = slot)
    # role 4: lane = d0 + 8 c1, slot n0
    v = img3[a43(wave[:, None], hi[:, None], np.arange(S)[None, :], lo[:, None])]
    return dft8(v)                                               # slots d1


def inverse_mag(R, tab):
    """R: product spectrum in role-4 layout [T][d1] -> (|r|^2 per (T, slot n3) in role-1 layout, 'full' index)."""
    tw1, tb, tc, w16 = tab
    T = np.arange(NT)
    p, s = T >> 9, T & 511
    wave, lane = T >> 6, T & 63
    hi, lo = (s >> 3) & 7, s & 7
    sw = lambda z: z.imag + 1j * z.real
    v = dft8(sw(R))                                              # role 4: d1 -> n0
    img = np.zeros(16 * REG, complex)
    img[a43(wave[:, None], hi[:, None], np.arange(S)[None, :], lo[:, None])] = v         # (c1 = hi, n0 = slot, d0 = lo)
    v = img[a43(wave[:, None], hi[:, None], lo[:, None], np.arange(S)[None, :])]         # role 3: lane n0 + 8 c1, slot d0
    v = dft8(v * tc[lo])                                         # d0 -> n1
    img[a32(wave[:, None], hi[:, None], np.arange(S)[None, :], lo[:, None])] = v         # (c1 = hi, n1 = slot, n0 = lo)
    v = img[a32(wave[:, None], np.arange(S)[None, :], hi[:, None], lo[:, None])]         # role 2: lane n0 + 8 n1, slot c1
    v = dft8(v * tb[lane])                                       # c1 -> n2
    c0 = s >> 6
    img2 = np.zeros(16 * REG, complex)
    img2[a21(p[:, None], c0[:, None], np.arange(S)[None, :], hi[:, None], lo[:, None])] = v
    t1, p1 = T >> 1, T & 1
    v = img2[a21(p1[:, None], np.arange(S)[None, :], (t1 >> 6)[:, None], ((t1 >> 3) & 7)[:, None], (t1 & 7)[:, None])]
    v = dft8(v * tw1)                                            # c0 -> n3 ; odd lanes carry W_L^t1
    v = np.where(p1[:, None] == 1, v * w16[None, :], v)          # odd lanes: * W16^n3
    partner = v.reshape(NT // 2, 2, S)[:, ::-1, :].reshape(NT, S)
    sgn = np.where(p1 == 1, -1.0, 1.0)[:, None]
    r = sgn * v + partner                                        # even: e + o' = r[n]; odd: o' - e = -r[n + M]
    n = 512 * np.arange(S)[None, :] + t1[:, None]
    kfull = np.where(p1[:, None] == 0, n + M - 1, n - 1)
    return np.abs(sw(r)) / L, kfull


def main():
    assert check_banks(), "bank conflicts"
    tab = tables()
    rng = np.random.default_rng(1)
    xi = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    xj = np.roll(xi, 37) + 0.3 * (rng.standard_normal(M) + 1j * rng.standard_normal(M))
    Xi, Xj = forward(xi, tab), forward(xj, tab)
    p, k = role4_index()
    ref = np.fft.fft(xi, L)
    assert np.allclose(Xi, ref[2 * k + p[:, None]], atol=1e-8), "forward spectrum"
    mag, kfull = inverse_mag(Xj * np.conj(Xi), tab)
    r = np.fft.ifft(np.fft.fft(xj, L) * np.conj(np.fft.fft(xi, L)))
    full = np.abs(np.concatenate([r[L - (M - 1):], r[:M]]))
    got = np.full(2 * M - 1, np.nan)
    valid = kfull >= 0
    got[kfull[valid]] = mag[valid]
    assert not np.isnan(got).any() and np.allclose(got, full, atol=1e-9), "inverse / 'full' order"
    print("model_win8: forward, inverse, 'full' order and LDS maps OK; peak at lag", int(np.argmax(got)) - (M - 1))


if __name__ == "__main__":
    main()
