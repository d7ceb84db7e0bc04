#!/usr/bin/env python3
"""Model of k_win2's pair schedule (two resident anchors): executes the state machine the kernel runs with
symbolic register contents and checks, for every buoy count 2..32, that each pair (i<j) is produced exactly once
from the right spectra, that every spectrum a pair reads has been requested (and nothing it still needs was
overwritten), and counts the global traffic per window.   usage: model_kwin2_schedule.py"""
import sys


def out_of(B, i, j):
    return i * B - i * (i + 1) // 2 + (j - i - 1)


def run(B):
    sa = sb = sc = None          # register contents: ('X', b) spectrum of buoy b, ('x', b) samples
    stored = set()
    loads = stores = samples = 0
    done = {}

    def pair(anchor, stream, out):
        assert anchor[0] == 'X' and stream[0] == 'X', (B, anchor, stream)
        i, j = anchor[1], stream[1]
        assert i < j, (B, i, j)
        assert out == out_of(B, i, j), (B, i, j, out)
        assert (i, j) not in done
        done[(i, j)] = out

    # ---- phase 1: X_0, X_1 resident; X_e (e >= 2) transformed once, stored once, used twice
    samples += 1; sa = ('x', 0)
    if B > 1:
        samples += 1; sb = ('x', 1)
    sa = ('X', 0)                                    # fwd(sa)
    if B > 1:
        sb = ('X', 1)                                # fwd(sb), not stored
        sc = sb                                      # COPY
        pair(sa, sb, out_of(B, 0, 1))                # prefetch during its h1a: samples of buoy 2 -> sb (or BLOCK(2): never, B == 2 has no phase 2)
        if B > 2:
            samples += 1; sb = ('x', 2)
    flipped = 0
    for e in range(2, B):
        assert sb == ('x', e)
        sb = ('X', e); stores += 1; stored.add(e)    # fwd + store
        pair(sa, sb, out_of(B, flipped, e))
        sa, sc = sc, sa; flipped ^= 1                # SWAP
        pair(sa, sb, out_of(B, flipped, e))
        if e + 1 < B:
            samples += 1; sb = ('x', e + 1)
        elif B >= 4:                                 # BLOCK(2): sa <- X_2, sb <- X_3
            assert 2 in stored and 3 in stored
            loads += 2; sa = ('X', 2); sb = ('X', 3)
    # ---- phase 2: blocks of two resident anchors (a, a+1), streams j = B-1 .. a+2
    a = 2
    while a + 1 <= B - 1:
        assert sa == ('X', a) and sb == ('X', a + 1), (B, a, sa, sb)
        sc = sb; flipped = 0                         # COPY
        n = B - a - 2
        pair(sa, sb, out_of(B, a, a + 1))
        nxt_block = a + 3 <= B - 1
        if n > 0:
            loads += 1; sb = ('X', B - 1)            # SB(j_1)
        elif nxt_block:
            loads += 2; sa = ('X', a + 2); sb = ('X', a + 3)
        for k in range(1, n + 1):
            j = B - k
            assert sb == ('X', j)
            pair(sa, sb, out_of(B, a + flipped, j))
            sa, sc = sc, sa; flipped ^= 1
            pair(sa, sb, out_of(B, a + flipped, j))
            if k < n:
                loads += 1; sb = ('X', B - k - 1)
            elif nxt_block:
                loads += 2; sa = ('X', a + 2); sb = ('X', a + 3)
        a += 2
    want = {(i, j) for i in range(B) for j in range(i + 1, B)}
    assert set(done) == want, (B, sorted(want - set(done)))
    return loads, stores, samples


if __name__ == '__main__':
    for B in range(2, 33):
        l, s, x = run(B)
        old_l = max(0, (B - 1) * (B - 2) // 2 + (B - 2)) if B > 2 else 0
        print('B=%2d pairs=%3d  spectrum loads %3d (k_win %3d)  stores %2d (k_win %2d)  sample loads %2d' %
              (B, B * (B - 1) // 2, l, old_l, s, B - 1, x))
    print('schedule ok for B = 2..32')
