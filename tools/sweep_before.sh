#!/bin/bash
# GPU box: the LDS-resident lengths of tools/sweep.sh with the whole-window kernels switched off (per-transform kernels / four-step)
cd $GRAFT_REPO_ROOT
export RMX_WFUSED=0 RMX_WSCR=0
for s in "3 256 16384" "3 512 16384" "3 1024 8192" "3 2048 4096" "3 8192 1024" "4 2048 4096" "8 256 8192" "8 512 8192" "8 1024 4096" "8 2048 2048" "8 8192 512" "16 2048 1024"; do
  python tools/bench_cfg.py $s 2>/dev/null
done
