#!/usr/bin/env python3
"""GPU box helper: per-kernel median duration of the full-size launches in a rocprofv3 --kernel-trace output dir.
usage: kstats.py <dir> [name-filter]"""
import collections, csv, glob, sys
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else "g_"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0][:48]
    if flt in n:
        agg[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print(k, len(v), "med_us", v[len(v) // 2] / 1e3)
