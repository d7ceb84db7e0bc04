cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r02_final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_final_pytest.log
for c in cfg3 cfg4; do python bench.py --config $c > gpurun_out/r02_bench_$c.log 2>&1; echo "$c rc=$?"; tail -1 gpurun_out/r02_bench_$c.log | cut -c1-200; done
tools/profile.sh r02 > gpurun_out/profile_r02.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_r02 r02 | tail -2
tools/profile.sh r02_cfg4 cfg4 > gpurun_out/profile_r02_cfg4.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_r02_cfg4 r02_cfg4 cfg4 | tail -2
cp profiles/r02* profiles/traffic_* gpurun_out/ 2>/dev/null
