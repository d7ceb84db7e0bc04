#!/bin/bash
# Profiling recipe for the GPU box (run through gpurun from the repo root):
#   tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/{stats,pmc_*}  (copy summaries into profiles/)
# Kernel-trace/stats and every PMC set are separate rocprofv3 runs (never combined).
# bench.py runs with --profile: the timed path and its parity leg only (no single-group probe, no host-pointer legs, no
# other shapes), so that counter totals divided by `engine_calls` are the timed path's (traffic_cfg2.json had
# the probe's 220 one-window calls in its sums).
set -o pipefail
#   tools/profile.sh <tag> cfg2   -> the same for another BASELINE shape (bench.py --config), stats + HBM counters only
tag=${1:-r01}
cfg=${2:-cfg3}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ "$cfg" = cfg3 ]; then
  BENCH="python3 $PWD/bench.py --steps 100 --warmup 20 --profile"
else
  BENCH="python3 $PWD/bench.py --config $cfg --profile"
fi
cd /tmp
rocprofv3 --kernel-trace --stats -T -f csv -d "$out/stats" -o stats -- $BENCH > "$out/stats.log" 2>&1 || echo "stats run failed"
sets=("FETCH_SIZE" "WRITE_SIZE")
[ "$cfg" = cfg3 ] && sets+=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum")
for set in "${sets[@]}"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -T -f csv -d "$out/pmc_$name" -o pmc -- $BENCH > "$out/pmc_$name.log" 2>&1 || echo "pmc $name failed"
done
ls -R "$out" | head -50
exit 0
