#!/bin/bash
# Timing-only ablation build of the fused window kernel (results are wrong by construction):
# -DRMX_ABLATE makes k_win honour the runtime "dbg" option; tools/ablate_run.py loops the masks
#   1 no workgroup barrier   2 no peak search/resolve   4 no B<->C (wave-local) exchange
#   8 no A<->B (cross-wave) exchange   16 no global loads inside the pair loop
set -e
cd "$(dirname "$0")/../radio-mapper_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -mllvm -simplifycfg-sink-common=false \
  -DRMX_ABLATE -I../../include -o librmx_ablate.so rmx_hip.hip
