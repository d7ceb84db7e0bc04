#!/bin/bash
# GPU box: matrix-pipe counters of tools/probe/mfma_coexec (separate --pmc passes) -> gpurun_out/pmc_mfma.txt
set -o pipefail
out=$PWD/gpurun_out/pmc_mfma
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BIN=$PWD/tools/probe/mfma_coexec
cd /tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -f csv -d "$out/p$i" -o pmc -- $BIN 1 > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'k<(\d+), (\d+), (\d+)>', k)
        if not m: continue
        acc[tuple(int(x) for x in m.groups())][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in acc for c in acc[k]})
print('shape mode NV  ' + '  '.join(names))
for k in sorted(acc):
    print('%d %d %3d  ' % k + '  '.join('%s=%.4g' % (c, max(acc[k][c])) for c in names if c in acc[k]))
PY
rm -rf "$out"/p*/
