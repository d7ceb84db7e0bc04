#!/bin/bash
# builds tools/probe/kwin_bench (+ the ISA of its kernels under /tmp/kwb for inspection)
set -e
cd "$(dirname "$0")"
out=kwin_bench
case "$1" in -*|"") ;; *) out=$1; shift;; esac
mkdir -p /tmp/kwb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
  -save-temps=obj -Rpass-analysis=kernel-resource-usage "$@" -o $out kwin_bench.hip 2> /tmp/kwb/res.txt || { tail -30 /tmp/kwb/res.txt; exit 1; }
mv -f kwin_bench-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/kwb/$out.s
rm -f kwin_bench-hip-* kwin_bench-host-* kwin_bench.hip-hip-*
python3 ../kres.py /tmp/kwb/res.txt k_win 2>/dev/null || grep -A8 "Function Name: .*k_win" /tmp/kwb/res.txt | grep -i "name\|VGPRs:\|Spill\|Occupancy" 
