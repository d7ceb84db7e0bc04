cd $GRAFT_REPO_ROOT
for s in "3 8192 1" "3 8192 16" "3 8192 64" "3 8192 256" "3 2048 1" "3 2048 16" "3 2048 128" "8 2048 1" "8 2048 32" "8 2048 256" "8 8192 8" "8 8192 64"; do
  echo "--- $s"
  RMX_WSCR=1 RMX_WFUSED=1 python tools/bench_cfg.py $s 9 2>/dev/null
  RMX_WSCR=0 RMX_WFUSED=0 python tools/bench_cfg.py $s 9 2>/dev/null
done
