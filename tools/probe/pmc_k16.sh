#!/bin/bash
# GPU box: L2 / SQ counters of k16_fwd and k16_pairs in tools/probe/k16_bench (separate --pmc passes, short runs)
#   tools/probe/pmc_k16.sh [k16_bench arguments ...]
set -o pipefail
out=$PWD/gpurun_out/${K16_PMC_OUT:-pmc_k16}
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
BIN=${K16_BIN:-$PWD/tools/probe/k16_bench}
args=${@:-8 256 1 256 32 1}
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -f csv -d "$out/p$i" -o pmc -- $BIN $args > "$out/p$i.log" 2>&1 || echo "pmc pass $i failed"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k16_' not in k: continue
        k = 'k16_fwd' if 'k16_fwd' in k else 'k16_pairs'
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print('   %-26s %.5g (n=%d)' % (c, sum(v) / len(v), len(v)))
PY
