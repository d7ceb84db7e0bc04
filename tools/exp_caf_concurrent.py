#!/usr/bin/env python3
"""GPU box experiment: do two rmx_caf_batch calls on two engines (two streams, two host threads) overlap usefully?
One engine with W windows against two engines with W/2 each, run concurrently.  usage: exp_caf_concurrent.py B N W D"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import radio_mapper_amd as rm
from radio_mapper_amd import xcorr
B, N, W, D = [int(a) for a in sys.argv[1:5]]
fs = 20e6
grid = (np.arange(D) - D // 2) * (50.0 / fs)
iq, _ = rm.synth.make_windows(W, B, N, fs, seed=1005)
one = xcorr.XcorrEngine(B, N, W)
one.caf(iq, grid)
t0 = time.perf_counter(); one.caf(iq, grid); t_one = time.perf_counter() - t0
one.close()
h = W // 2
engs = [xcorr.XcorrEngine(B, N, h), xcorr.XcorrEngine(B, N, h)]
parts = [iq[:h], iq[h:]]
for e, p in zip(engs, parts): e.caf(p, grid)
def run(i): engs[i].caf(parts[i], grid)
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
for t in th: t.start()
for t in th: t.join()
t_two = time.perf_counter() - t0
t0 = time.perf_counter(); run(0); run(1); t_seq = time.perf_counter() - t0
print(f"one engine, {W} windows: {t_one*1e3:.1f} ms; two engines x {h} concurrently: {t_two*1e3:.1f} ms; the two one after the other: {t_seq*1e3:.1f} ms")
