#!/bin/bash
# GPU box: SQ counters of the four-step kernels on one shape (separate rocprofv3 --pmc passes, no tracing).
# (FETCH_SIZE / WRITE_SIZE / GRBM go in their own passes: tools/profile.sh; combined with each other they hung a run)
# usage: tools/pmc_cfg.sh <tag> B N W   -> gpurun_out/pmc_<tag>/*.csv + a per-kernel summary on stdout
tag=$1; shift
R=$PWD
out=$R/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set -f csv -d "$out/s$i" -o pmc -- python3 $R/tools/bench_cfg.py "$@" 2 > "$out/s$i.log" 2>&1 || echo "pmc set $i failed"
done
cd $R
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
out = {}
for k, d in agg.items():
    if "g_" not in k: continue
    print(k)
    out[k] = {}
    for c, v in sorted(d.items()):
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
        out[k][c] = sum(v) / len(v)
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
