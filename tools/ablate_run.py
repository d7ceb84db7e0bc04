#!/usr/bin/env python3
"""GPU box: time k_win at the cfg3 shape under each ablation mask (needs tools/ablate.sh's build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RMX_LIBRARY"] = os.path.join(ROOT, "radio-mapper_amd/csrc/librmx_ablate.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_probe
masks = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 12, 16, 3, 15, 31]
for m in masks:
    gpu_probe.timing(chunk=4096, dbg=m, reps=7)
