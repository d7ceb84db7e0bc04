#!/bin/bash
# GPU box (one GPU): same-device rehearsal of the N-rank bench -- two ranks on cuda:0 over gloo (RCCL refuses two ranks on
# one device), every BASELINE shape that is "sharded across 8 GPUs", weak and strong.  Not a scaling measurement (the two
# ranks share one device): it shows that the launcher, the shard arithmetic, the barrier/MAX timing and the gather work for
# those shapes before a real 8-GPU node runs them.   usage: tools/rehearse_multi.sh <tag>  -> profiles/<tag>_bench_2rank_same_device.log
set -o pipefail
tag=${1:-r04}
out=profiles/${tag}_bench_2rank_same_device.log
: > $out
export RMX_BENCH_SAME_DEVICE=1
for args in "--config cfg3 --steps 20 --warmup 5" "--config cfg3 --scaling strong --steps 20 --warmup 5" \
            "--config cfg4 --steps 5 --warmup 2" "--config cfg4 --scaling strong --windows 1024 --steps 5 --warmup 2" \
            "--config cfg5 --steps 1 --warmup 1" "--config cfg5 --scaling strong --windows 16 --steps 1 --warmup 1"; do
  echo "### RMX_BENCH_SAME_DEVICE=1 python bench.py --gpus 2 $args --no-cpu-baseline" >> $out
  timeout -k 10 400 python bench.py --gpus 2 $args --no-cpu-baseline 2> gpurun_out/rehearse_err.log | grep '^{' | cut -c1-1400 >> $out || { echo "FAILED (see stderr below)" >> $out; tail -5 gpurun_out/rehearse_err.log >> $out; }
done
cp $out gpurun_out/
cut -c1-260 $out
