#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels whose mangled name contains <filter>, from
   hipcc ... -Rpass-analysis=kernel-resource-usage 2> res.txt      usage: kres.py res.txt [filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split(" ")[0]
    if flt not in name:
        continue
    g = lambda k: re.search(re.escape(k) + r": (\d+)", b).group(1)
    try:
        dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.split("(")[0]
    except Exception:
        dn = name
    print(f"{dn:70s} VGPR {g('VGPRs'):>3} spill {g('VGPRs Spill'):>3} scratch {g('ScratchSize [bytes/lane]'):>4} waves/SIMD {g('Occupancy [waves/SIMD]')}")
