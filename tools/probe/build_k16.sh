#!/bin/bash
# build tools/probe/k16_bench (extra -D switches as arguments) and print the kernels' register / spill counts
cd "$(dirname "$0")"
out=${K16_OUT:-k16_bench}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false -Wno-inline-asm \
  -I../../radio-mapper_amd/csrc -save-temps=obj "$@" -o $out k16_bench.hip || exit 1
f=$(ls ${out}-hip-amdgcn*.s 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls k16_bench-hip-amdgcn*.s | head -1)
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|private_segment_fixed_size|name):" $f | paste - - - - | grep k16 | sed 's/_ZN3rmx3k16//; s/EPK.*\.private/ .private/; s/ILb.*\.private/ .private/'
mkdir -p /tmp/k16 && mv -f *-hip-amdgcn*.s /tmp/k16/ 2>/dev/null
rm -f k16_bench-hip-* k16_bench-host-* ${out}-hip-* ${out}-host-*
