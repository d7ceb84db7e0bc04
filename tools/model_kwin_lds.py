#!/usr/bin/env python3
"""numpy/python model of k_win's second-generation LDS exchanges (fft_r16.hpp): executes the kernel's index maps with
symbolic payloads and checks (1) that every thread of the next role receives exactly the 16 values the transform
needs, in slot order, in both directions, (2) that the wave-local exchange only touches the wave's own region, and
(3) the bank behaviour of every wave-instruction under the rules of the MI355X guide (lane groups per instruction,
64 banks for ds_read_b64 / b128, 32 for stores).   usage: model_kwin_lds.py"""
import itertools
import sys

K_BC_HALF = 544                      # complex per k0 row of the barrier image
LOC_ROW, LOC_PLANE, LOC_WAVE = 272, 16 * 272, 2 * 16 * 272


def loc_pos(s):
    return 2 * s if s < 4 else (2 * (s - 8) if s >= 12 else 2 * (s - 4) + 1)


# ---- role maps: thread id -> digits ------------------------------------------------------------------------------
def role_a(t):   # (n1, n0, p)
    p, u = t & 1, t >> 1
    return u >> 4, u & 15, p


def role_b(t):   # (k0, n0, p)
    return t >> 5, t & 15, (t >> 4) & 1


def role_c(t):   # (k0, k1, p)
    lane, wave = t & 63, t >> 6
    return 2 * wave + ((lane >> 4) & 1), lane & 15, lane >> 5


def xa2(t, k0):                       # complex index of role A's slot k0
    n1, n0, p = role_a(t)
    return k0 * K_BC_HALF + 32 * n1 + 16 * p + n0


def xb2(t, n1):                       # complex index of role B's slot n1
    k0, n0, p = role_b(t)
    return k0 * K_BC_HALF + 32 * n1 + 16 * p + n0


def loc_write_addr(t, slot, plane):   # byte address (addtid: M0 = wave region, offset = row, + 4 * lane)
    lane, wave = t & 63, t >> 6
    return wave * LOC_WAVE + plane * LOC_PLANE + loc_pos(slot) * LOC_ROW + 4 * lane


def loc_read_addr(t, q, plane):       # byte address of float q of this lane's run
    lane, wave = t & 63, t >> 6
    return wave * LOC_WAVE + plane * LOC_PLANE + loc_pos(lane & 15) * LOC_ROW + (((lane >> 4) & 1) * 2 + (lane >> 5)) * 64 + 4 * q


def check_dataflow():
    T = range(512)
    # forward: A --barrier--> B --local--> C
    img = {}
    for t in T:
        n1, n0, p = role_a(t)
        for k0 in range(16):
            assert xa2(t, k0) not in img
            img[xa2(t, k0)] = (n1, n0, p, k0)
    for t in T:
        k0, n0, p = role_b(t)
        for n1 in range(16):
            assert img[xb2(t, n1)] == (n1, n0, p, k0), ("A->B", t, n1)
    loc = {}
    for t in T:
        k0, n0, p = role_b(t)
        for k1 in range(16):
            a = loc_write_addr(t, k1, 0)
            assert a not in loc and (t >> 6) * LOC_WAVE <= a < ((t >> 6) + 1) * LOC_WAVE
            loc[a] = (k0, n0, p, k1)
    for t in T:
        k0, k1, p = role_c(t)
        for n0 in range(16):
            assert loc[loc_read_addr(t, n0, 0)] == (k0, n0, p, k1), ("B->C", t, n0)
    # inverse: C --local--> B --barrier--> A
    loc = {}
    for t in T:
        k0, k1, p = role_c(t)
        for n0 in range(16):
            loc[loc_write_addr(t, n0, 1)] = (k0, k1, p, n0)
    for t in T:
        k0, n0, p = role_b(t)
        for k1 in range(16):
            assert loc[loc_read_addr(t, k1, 1)] == (k0, k1, p, n0), ("C->B", t, k1)
    img = {}
    for t in T:
        k0, n0, p = role_b(t)
        for n1 in range(16):
            img[xb2(t, n1)] = (k0, n0, p, n1)
    for t in T:
        n1, n0, p = role_a(t)
        for k0 in range(16):
            assert img[xa2(t, k0)] == (k0, n0, p, n1), ("B->A", t, k0)
    # the wave-local region is the wave's own two k0 rows of the barrier image (what it alone writes in role B)
    for t in T:
        k0, _, _ = role_b(t)
        lo, hi = (t >> 6) * LOC_WAVE, ((t >> 6) + 1) * LOC_WAVE
        for n1 in range(16):
            assert lo <= 8 * xb2(t, n1) < hi
    assert 16 * K_BC_HALF * 8 == 8 * LOC_WAVE
    print("data flow ok: A->B->C and C->B->A deliver the right 16 values to every thread; local regions are private")


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def worst(addrs_by_lane, groups, width, banks):
    """largest number of distinct addresses on one bank inside one lane group (1 = conflict free)"""
    w = 0
    for g in groups:
        per_bank = {}
        for l in g:
            a = addrs_by_lane[l]
            for d in range(width // 4):
                per_bank.setdefault(((a // 4) + d) % banks, set()).add(a + 4 * d)
        w = max(w, max(len(v) for v in per_bank.values()))
    return w


def check_banks():
    res = {}
    for wave in range(8):
        lanes = [wave * 64 + l for l in range(64)]
        for s in range(16):
            # barrier image: ds_read_b64 = 2 groups of 32 lanes over 64 banks; ds_write_b64 = 4 groups of 16 contiguous lanes over 32 banks
            rd_groups = [list(range(0, 32)), list(range(32, 64))]
            wr_groups = [list(range(16 * g, 16 * g + 16)) for g in range(4)]
            res["A read  (b64)"] = max(res.get("A read  (b64)", 0), worst([8 * xa2(t, s) for t in lanes], rd_groups, 8, 64))
            res["A write (b64, forward only)"] = max(res.get("A write (b64, forward only)", 0), worst([8 * xa2(t, s) for t in lanes], wr_groups, 8, 32))
            res["B read  (b64)"] = max(res.get("B read  (b64)", 0), worst([8 * xb2(t, s) for t in lanes], rd_groups, 8, 64))
            res["B write (b64)"] = max(res.get("B write (b64)", 0), worst([8 * xb2(t, s) for t in lanes], wr_groups, 8, 32))
            # local image: addtid stores are lane-contiguous dwords (2 groups of 32 over 32 banks)
            res["local write (addtid)"] = max(res.get("local write (addtid)", 0),
                                              worst([loc_write_addr(t, s, 0) for t in lanes], [list(range(0, 32)), list(range(32, 64))], 4, 32))
        for j in range(4):
            res["local read (b128)"] = max(res.get("local read (b128)", 0),
                                           worst([loc_read_addr(t, 4 * j, 0) for t in lanes], B128_GROUPS, 16, 64))
    for k, v in res.items():
        print("%-30s %d-way%s" % (k, v, "" if v == 1 else "   <-- conflict"))
    assert all(v == 1 for k, v in res.items() if "forward only" not in k)
    assert res["A write (b64, forward only)"] == 2


def brute_force_local():
    """the search behind the local layout: row pad in bank quads, S / S^c row interleave, reader group permutation"""
    best = []
    for stride_q in range(16):
        for perm in itertools.permutations(range(4)):
            for interleave in (False, True):
                w = 0
                for g in B128_GROUPS[:2]:
                    for j in range(4):
                        cnt = {}
                        for l in g:
                            d = l & 15
                            pos = loc_pos(d) if interleave else d
                            q = (pos * (16 + stride_q) + 4 * perm[l >> 4] + j) % 16
                            cnt[q] = cnt.get(q, 0) + 1
                        w = max(w, max(cnt.values()))
                best.append((w, stride_q, interleave, perm))
    best.sort()
    free = [b for b in best if b[0] == 1]
    print("local b128 layouts searched: %d, conflict free: %d, e.g. row pad %d quads, interleave %s, reader groups %s"
          % (len(best), len(free), free[0][1], free[0][2], free[0][3]))
    assert (1, 1, True, (0, 2, 1, 3)) in free      # the one the kernel uses: group = 2 * bit4 + bit5


if __name__ == "__main__":
    check_dataflow()
    check_banks()
    brute_force_local()
    print("model ok")
