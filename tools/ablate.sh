#!/bin/bash
# timing-only ablation builds of the fused window kernel (results are wrong by construction)
cd "$(dirname "$0")/../radio-mapper_amd/csrc"
for m in 0 1 2 4 8 12 16 32 48 63; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -DRMX_ABLATE=$m -o librmx_ablate_$m.so rmx_hip.hip || exit 1
done
