#!/bin/bash
# GPU box: the evidence run of a round -- GPU test suite, bench lines of every BASELINE shape, rocprofv3 kernel stats and
# PMC passes; summaries land in profiles/<tag>_* (copied back through gpurun_out/).   usage: tools/evidence.sh r04
set -o pipefail
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
part=${EVIDENCE_PART:-all}      # a: tests, bench lines, profiles and counters of the BASELINE shapes; b: the N = 8192 / 16384 tables; all: both
if [ "$part" != b ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_final_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${tag}_final_pytest.log
# (the default line = cfg3 with the other four shapes inside it; the per-shape lines below are each shape's own full line)
for c in cfg3 cfg2 cfg4 cfg5 cfg1; do python bench.py --config $c > gpurun_out/${tag}_bench_$c.log 2>&1; echo "$c rc=$?"; tail -1 gpurun_out/${tag}_bench_$c.log | cut -c1-200; tail -1 gpurun_out/${tag}_bench_$c.log > profiles/${tag}_bench_$c.json; done
tools/profile.sh $tag > gpurun_out/profile_$tag.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_$tag $tag | tail -2
for c in cfg2 cfg4 cfg5 cfg1; do tools/profile.sh ${tag}_$c $c > gpurun_out/profile_${tag}_$c.log 2>&1; python tools/summarize_prof.py gpurun_out/prof_${tag}_$c ${tag}_$c $c | tail -2; done
# cfg1 (one window per call, five launches): durations and gaps between consecutive kernels of a step from the kernel trace
python - <<PY
import csv, glob, statistics as st
tr = glob.glob("gpurun_out/prof_${tag}_cfg1/stats/**/*kernel_trace.csv", recursive=True)
if tr:
    rows = sorted((r for r in csv.DictReader(open(tr[0])) if r["Kernel_Name"].startswith(("g_cols", "g_rows", "g_final"))), key=lambda r: int(r["Start_Timestamp"]))
    names = ["g_cols_fwd", "g_rows", "g_rows", "g_cols_inv", "g_final"]
    steps = []
    i = 0
    while i + 5 <= len(rows):
        blk = rows[i:i + 5]
        if all(b["Kernel_Name"].startswith(n) for n, b in zip(names, blk)):
            steps.append(blk); i += 5
        else:
            i += 1
    out = ["# cfg1: one rmx_xcorr_batch call = five launches; medians over %d steps of bench.py --config cfg1 --profile under rocprofv3 --kernel-trace" % len(steps)]
    if steps:
        for k in range(5):
            d = st.median(int(s[k]["End_Timestamp"]) - int(s[k]["Start_Timestamp"]) for s in steps)
            g = st.median(int(s[k + 1]["Start_Timestamp"]) - int(s[k]["End_Timestamp"]) for s in steps) if k < 4 else float("nan")
            out.append("%-12s duration %7.2f us   gap to the next kernel %6.2f us" % (names[k].rstrip("<"), d / 1e3, g / 1e3))
        span = st.median(int(s[4]["End_Timestamp"]) - int(s[0]["Start_Timestamp"]) for s in steps)
        gaps = [g for g in (int(b[0]["Start_Timestamp"]) - int(a[4]["End_Timestamp"]) for a, b in zip(steps, steps[1:])) if g < 20000]   # back-to-back calls only
        nxt = st.median(gaps) if gaps else float("nan")
        out.append("first start -> last end %.2f us; gap to the next call's first kernel %.2f us" % (span / 1e3, nxt / 1e3))
    open("profiles/${tag}_cfg1_gaps.txt", "w").write("\n".join(out) + "\n")
    print("\n".join(out))
PY
cp profiles/${tag}* profiles/traffic_* profiles/pmc_latest.json gpurun_out/ 2>/dev/null
fi
if [ "$part" = a ]; then exit 0; fi
# round 5: the N = 8192 kernel (k_win8kl) -- sweep against g_win_scr14, forward / pair split, rocprofv3 kernel stats of a full batch;
# the 32-column-tile experiment of the four-step column kernels
python tools/exp_k8_sweep.py 2>&1 | grep -v amdgpu.ids > profiles/${tag}_n8192_sweep.txt; tail -3 profiles/${tag}_n8192_sweep.txt
{ python tools/exp_k8_split.py 8 512; python tools/exp_k8_split.py 8 128; python tools/exp_k8_split.py 3 1024; python tools/exp_k8_split.py 8 256 16384; } 2>&1 | grep -v amdgpu.ids > profiles/${tag}_n8192_split.txt
python tools/exp_k8_small.py 2>&1 | grep -v amdgpu.ids > profiles/${tag}_n8192_small.txt
python tools/exp_cols32.py 2>&1 | grep -v amdgpu.ids > profiles/${tag}_cols32.txt
( export TMPDIR=/tmp; R=$PWD; cd /tmp && RMX_WSCR=2 rocprofv3 --kernel-trace --stats -T -f csv -d $R/gpurun_out/prof_${tag}_n8192 -o stats -- python3 $R/tools/bench_cfg.py 8 8192 512 30 > $R/gpurun_out/prof_${tag}_n8192.log 2>&1 )
python - <<PY
import csv, glob
st = glob.glob("gpurun_out/prof_${tag}_n8192/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.reader(open(st[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "k_win8" in r[0] or "g_win" in r[0]]
    csv.writer(open("profiles/${tag}_n8192_kernel_stats.csv", "w", newline="")).writerows(keep)
    print(keep[1][:4] if len(keep) > 1 else "no k_win8kl row")
PY
cp profiles/${tag}* profiles/traffic_* profiles/pmc_latest.json gpurun_out/ 2>/dev/null
cp profiles/${tag}* profiles/traffic_* profiles/pmc_latest.json gpurun_out/ 2>/dev/null
python tools/exp_tail_split.py 2>&1 | grep -v amdgpu.ids > profiles/${tag}_tail_split.txt
python tools/exp_k8_u8.py 2>&1 | grep -v amdgpu.ids > profiles/${tag}_u8_ingest.txt
# N = 16384 (k16_fwd + k16_pairs, kwin16k.hpp): sweep against g_win_eo15 / the four-step kernels, rocprofv3 kernel stats of a full batch
python tools/exp_k16_sweep.py 1 4 16 64 128 256 300 1024 2>&1 | grep -v amdgpu.ids > profiles/${tag}_n16384_sweep.txt; tail -3 profiles/${tag}_n16384_sweep.txt
( export TMPDIR=/tmp; R=$PWD; cd /tmp && rocprofv3 --kernel-trace --stats -T -f csv -d $R/gpurun_out/prof_${tag}_n16384 -o stats -- python3 $R/tools/bench_cfg.py 8 16384 256 30 > $R/gpurun_out/prof_${tag}_n16384.log 2>&1 )
python - <<PY
import csv, glob
st = glob.glob("gpurun_out/prof_${tag}_n16384/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.reader(open(st[0])))
    keep = [rows[0]] + [r for r in rows[1:] if "k16_" in r[0] or "g_win" in r[0]]
    csv.writer(open("profiles/${tag}_n16384_kernel_stats.csv", "w", newline="")).writerows(keep)
    print([r[:4] for r in keep[1:]])
PY
# ... and the L2 / HBM / SQ counters of the two kernels in the standalone harness (8 buoys x 256 windows, one launch each per pass)
tools/probe/build_k16.sh > gpurun_out/build_k16.log 2>&1 && { timeout -k 10 60 tools/probe/k16_bench 8 256 30 256 | tail -1; timeout -k 10 60 tools/probe/k16_bench 3 256 30 256 | tail -1; timeout -k 10 60 tools/probe/k16_bench 16 256 10 256 | tail -1; tools/probe/pmc_k16.sh; } > profiles/${tag}_n16384_harness_pmc.txt 2>&1
cp profiles/${tag}* gpurun_out/ 2>/dev/null
