#!/bin/bash
# GPU box: the evidence run of a round -- GPU test suite, bench lines of every BASELINE shape, rocprofv3 kernel stats and
# PMC passes; summaries land in profiles/<tag>_* (copied back through gpurun_out/).   usage: tools/evidence.sh r04
set -o pipefail
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${tag}_final_pytest.log
# (the default line = cfg3 with the other four shapes inside it; the per-shape lines below are each shape's own full line)
for c in cfg3 cfg2 cfg4 cfg5 cfg1; do python bench.py --config $c > gpurun_out/${tag}_bench_$c.log 2>&1; echo "$c rc=$?"; tail -1 gpurun_out/${tag}_bench_$c.log | cut -c1-200; tail -1 gpurun_out/${tag}_bench_$c.log > profiles/${tag}_bench_$c.json; done
tools/profile.sh $tag > gpurun_out/profile_$tag.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_$tag $tag | tail -2
for c in cfg2 cfg4 cfg5; do tools/profile.sh ${tag}_$c $c > gpurun_out/profile_${tag}_$c.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_${tag}_$c ${tag}_$c $c | tail -2; done
cp profiles/${tag}* profiles/traffic_* gpurun_out/ 2>/dev/null
