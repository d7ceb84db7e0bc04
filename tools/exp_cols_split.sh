#!/bin/bash
# GPU box: g_cols_inv's loads alone (-DRMX_EXP_COLS_NOCOMP) and its arithmetic alone (-DRMX_EXP_COLS_NOLOAD) against the
# kernel as it is: can overlapping a tile's loads with the previous tile's passes (LDS-DMA double buffering) pay?
# Timing-only builds (wrong results), made with:
#   hipcc <HIP_FLAGS of __graft_entry__.py> -DRMX_EXP_COLS_NOLOAD -o tools/probe/librmx_noload.so radio-mapper_amd/csrc/rmx_hip.hip   (and _NOCOMP)
set -o pipefail
R=$PWD
out=$R/gpurun_out/cols_split
mkdir -p $out
log=$R/gpurun_out/cols_split.txt
: > $log
export TMPDIR=/tmp
cd /tmp
for shape in "3 1048576 64" "8 262144 8" "3 262144 64"; do
  for lib in "" tools/probe/librmx_noload.so tools/probe/librmx_nocomp.so; do
    if [ -z "$lib" ]; then unset RMX_LIBRARY; name=product; else export RMX_LIBRARY=$R/$lib; name=$(basename $lib .so); fi
    tag=$(echo $shape | tr ' ' '_')_$name
    echo "=== shape $shape  library $name" >> $log
    timeout -k 10 150 rocprofv3 --kernel-trace -f csv -d "$out/$tag" -o kt -- python3 $R/tools/bench_cfg.py $shape 3 > "$out/$tag.log" 2>&1 || echo "trace failed" >> $log
    python3 $R/tools/kstats.py "$out/$tag" g_ >> $log 2>&1
    rm -rf "$out/$tag"
  done
done
unset RMX_LIBRARY
cat $log
