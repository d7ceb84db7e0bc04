#!/bin/bash
# GPU box: small batches with the whole-window kernel forced (RMX_WSCR=2) and off (RMX_WSCR=0): where is the crossover?
cd $GRAFT_REPO_ROOT
for s in "3 8192 32" "3 8192 64" "3 8192 96" "3 8192 128" "3 8192 192" "8 2048 32" "8 2048 64" "8 2048 96" "8 2048 128" "8 2048 192" "8 8192 32" "8 8192 64" "8 8192 128" "8 8192 192" "8 512 256" "8 512 512" "8 512 1024"; do
  echo "--- $s"
  RMX_WSCR=2 python tools/bench_cfg.py $s 9 2>/dev/null
  RMX_WSCR=0 RMX_WFUSED=0 python tools/bench_cfg.py $s 9 2>/dev/null
done
