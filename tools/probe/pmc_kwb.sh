#!/bin/bash
# GPU box: SQ counters of every kernel variant in tools/probe/kwin_bench (separate --pmc passes, short runs)
set -o pipefail
out=$PWD/gpurun_out/${KWB_OUT:-pmc_kwb}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BIN=${KWB_BIN:-$PWD/tools/probe/kwin_bench}   # KWB_BIN=...: another build of the harness (tools/probe/kwb_*)
cd /tmp
[ -f $out/../counters.txt ] || rocprofv3 -L > $out/../counters.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "$@"; do
  [ -z "$set" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -f csv -d "$out/p$i" -o pmc -- $BIN 4096 12 6 > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_win' not in k: continue
        k = k.split("(")[0].replace("void rmx::", "")
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in acc for c in acc[k]})
for k in sorted(acc):
    print(k)
    for c in names:
        v = acc[k].get(c)
        if v: print('   %-24s %.4g (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
