#!/bin/bash
# GPU box: does a four-step batch run faster when it is cut into chunks small enough for the rows<->columns round trip to
# stay in the 256 MiB memory-side cache?    For every shape and every chunk size: the call's time
# (tools/bench_cfg.py) and, from a rocprofv3 --kernel-trace of the same command, the per-kernel medians (tools/kstats.py).
#   usage: tools/exp_gen_chunk.sh            -> gpurun_out/gen_chunk.txt
set -o pipefail
R=$PWD
out=$R/gpurun_out/gen_chunk
mkdir -p "$out"
export TMPDIR=/tmp
log=$R/gpurun_out/gen_chunk.txt
: > "$log"
cd /tmp
for shape in "3 1048576 64" "3 262144 64" "8 262144 8" "4 1048576 16"; do
  for ch in 0 1 2 4 8; do
    tag=$(echo $shape | tr ' ' '_')_c$ch
    echo "=== shape $shape  gen_chunk $ch (0 = default: one chunk)" >> "$log"
    if [ $ch = 0 ]; then unset RMX_GEN_CHUNK; else export RMX_GEN_CHUNK=$ch; fi
    timeout -k 10 120 python3 $R/tools/bench_cfg.py $shape 7 >> "$log" 2>&1 || echo "bench failed" >> "$log"
    timeout -k 10 150 rocprofv3 --kernel-trace -f csv -d "$out/$tag" -o kt -- python3 $R/tools/bench_cfg.py $shape 3 > "$out/$tag.log" 2>&1 || echo "trace failed" >> "$log"
    python3 $R/tools/kstats.py "$out/$tag" g_ >> "$log" 2>&1
    rm -rf "$out/$tag"      # (raw traces: gpurun copies back at most 64 MiB)
  done
done
unset RMX_GEN_CHUNK
tail -5 "$log"
