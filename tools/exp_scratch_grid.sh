#!/bin/bash
# GPU box: is g_win_eo15 (N = 16384) / g_win_scr14 (N = 8192) bound by its spectrum scratch leaving the 256 MiB memory-side
# cache?    The persistent grid is capped with the `ncus` option: G workgroups keep G x B x 8L bytes of
# scratch live.  If the scratch traffic were the limit, the time per window and workgroup would drop once G x scratch fits.
#   usage: tools/exp_scratch_grid.sh   -> gpurun_out/scratch_grid.txt
out=$PWD/gpurun_out/scratch_grid.txt
: > $out
for shape in "8 16384 512" "16 16384 256" "8 8192 1024" "3 16384 1024"; do
  for g in 256 224 192 160 128 112 96 64 32; do
    echo "--- B N W = $shape   ncus=$g" >> $out
    RMX_NCUS=$g timeout -k 10 120 python3 tools/bench_cfg.py $shape 7 2>&1 | grep "^B=" >> $out
  done
done
cat $out
