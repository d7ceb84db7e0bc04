#!/bin/bash
# GPU box: L2 / fabric counters of tools/probe/xcd_l2_probe, one dispatch pair (warm-up + timed) per size and mode
set -o pipefail
out=$PWD/gpurun_out/pmc_xcd
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BIN=$PWD/tools/probe/xcd_l2_probe
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -f csv -d "$out/p$i" -o pmc -- $BIN 100 > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'xcd_probe' not in r['Kernel_Name']: continue
        c = r['Counter_Name']
        n[c] += 1
        rows[(n[c] - 1) // 2][c] = float(r['Counter_Value'])     # the second (timed) dispatch of each pair overwrites the first
sizes = [1, 2, 3, 4, 6, 8, 16, 64]
names = sorted({c for d in rows.values() for c in d})
print('%-12s' % 'S/XCD mode', ' '.join('%18s' % c for c in names))
for k in sorted(rows):
    print('%3d MiB  m%d ' % (sizes[k % 8], k // 8), ' '.join('%18.4g' % rows[k].get(c, float('nan')) for c in names))
PY
