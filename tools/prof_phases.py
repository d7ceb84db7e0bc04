#!/usr/bin/env python3
"""GPU box: per-phase cycle accounting of k_win's pair body (build: tools/prof_build.sh, -DRMX_PROF)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RMX_LIBRARY"] = os.path.join(ROOT, "radio-mapper_amd/csrc/librmx_prof.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_probe
from radio_mapper_amd import xcorr
gpu_probe.timing(chunk=4096, reps=3)
lib = xcorr.load_library()
buf = (C.c_ulonglong * 32)()
assert lib.rmx_debug_read_prof(buf) == 0
names = ["premul+passC", "prefetch+W wr/rd issue", "passB (incl W wait)", "X write issue", "drain+barrier+resolve",
         "X read issue", "passA (incl X wait)", "w32+radix2+mag", "reduce+record"]
for blk in range(2):
    v = [buf[blk * 16 + i] for i in range(10)]
    n = max(v[9], 1)
    tot = sum(v[:9])
    print(f"block {blk}: pairs {v[9]}  total {tot / n:.0f} ticks/pair (s_memtime ticks = 100 MHz? see ratio)")
    for i in range(9):
        print(f"   {names[i]:28s} {v[i] / n:9.1f}  {100.0 * v[i] / max(tot,1):5.1f}%")
