#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into profiles/<tag>_*.{csv,json}.

  python tools/summarize_prof.py gpurun_out/prof_<tag> <tag> [cfgN]

Writes  profiles/<tag>_kernel_stats.csv   (rocprofv3 --kernel-trace --stats, our kernels only)
        profiles/<tag>_pmc.json           (per-kernel averages of every collected counter)
        profiles/traffic_latest.json      (HBM bytes per launch of the dominant kernel, for bench.py)
HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are in KiB and come
from separate --pmc passes; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced
stream, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_digest   # noqa: E402  (stdlib-only at import: ties a traffic file to the sources it was taken on)


def main():
    src, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    ours = ("k_win", "k_fwd", "k_pair", "g_cols", "g_rows", "g_final", "g_fwd_small", "g_pair_small", "k_caf")
    cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
    # cfg3: the per-launch figures are the fused kernel's; bench.py's single-group probe (one window through k_fwd +
    # k_pair_res, 220 tiny launches) stays in the stats table but out of the full-size selection
    sel = ("k_win",) if cfg == "cfg3" else ours
    st = glob.glob(os.path.join(src, "stats", "*kernel_stats.csv"))
    if st:
        rows = list(csv.reader(open(st[0])))
        keep = [rows[0]] + [r for r in rows[1:] if any(k in r[0] for k in ours)]
        with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            csv.writer(f).writerows(keep)
    # full-size launches only: bench.py also launches the same kernel on 32 windows (parity leg) and on
    # 512-window pieces (host-pointer leg, same persistent grid), so launches are kept by grid size AND
    # by duration (at least half the median of the longer half)
    def full_size(rows, gkey):
        if not rows:
            return []
        gmax = max(int(r[gkey]) for r in rows)
        rows = [r for r in rows if int(r[gkey]) == gmax]
        d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
        ref = d[(len(d) * 3) // 4]
        return [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 0.5 * ref]

    tr = glob.glob(os.path.join(src, "stats", "*kernel_trace.csv"))
    if tr:
        rows = full_size([r for r in csv.DictReader(open(tr[0])) if any(k in r["Kernel_Name"] for k in sel)], "Grid_Size_X")
        if rows:
            d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
            with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "a", newline="") as f:
                csv.writer(f).writerow([f"# full-size launches only: n={len(d)} avg_ns={sum(d) / len(d):.0f} "
                                        f"median_ns={d[len(d) // 2]} min_ns={d[0]} max_ns={d[-1]}"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob(os.path.join(src, "pmc_*", "*counter_collection.csv")):
        rows = [r for r in csv.DictReader(open(p)) if any(k in r["Kernel_Name"] for k in sel)]
        by_disp = collections.defaultdict(list)
        for r in rows:
            by_disp[r["Dispatch_Id"]].append(r)
        keep = {r["Dispatch_Id"] for r in full_size([v[0] for v in by_disp.values()], "Grid_Size")}
        for r in rows:
            if r["Dispatch_Id"] in keep:
                name = r["Kernel_Name"].split("(")[0].split("<")[0]
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches_sampled": len(next(iter(d.values())))}
           for k, d in agg.items()}
    json.dump(pmc, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
    if cfg != "cfg3":
        # multi-kernel shapes: HBM bytes of ALL our kernels per engine call (= bench.py's step): counter totals over
        # every dispatch of the FETCH_SIZE / WRITE_SIZE passes divided by the engine calls bench.py reports
        tot = collections.defaultdict(float)
        ncalls = {}
        for cname in ("FETCH_SIZE", "WRITE_SIZE"):
            for p in glob.glob(os.path.join(src, f"pmc_{cname}", "*counter_collection.csv")):
                for r in csv.DictReader(open(p)):
                    if any(k in r["Kernel_Name"] for k in ours) and r["Counter_Name"] == cname:
                        tot[cname] += float(r["Counter_Value"])
            log = os.path.join(src, f"pmc_{cname}.log")
            for ln in open(log) if os.path.exists(log) else []:
                if ln.startswith("{"):
                    ncalls[cname] = json.loads(ln).get("engine_calls")
        if tot and all(ncalls.get(k) for k in ("FETCH_SIZE", "WRITE_SIZE")):
            f, w = tot["FETCH_SIZE"] / ncalls["FETCH_SIZE"], tot["WRITE_SIZE"] / ncalls["WRITE_SIZE"]
            traffic = {"kernel": "all kernels of one engine call", "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                       "fetch_size_kib_raw": f, "write_size_kib": w, "engine_calls": ncalls,
                       "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)", "tag": tag,
                       "source_digest": source_digest()}
            json.dump(traffic, open(os.path.join(out, f"traffic_{cfg}.json"), "w"), indent=1)
            print(traffic)
        print("kernels:", list(pmc))
        return
    dom = max(pmc, key=lambda k: pmc[k].get("SQ_WAVE_CYCLES", 0)) if pmc else None
    if dom and "FETCH_SIZE" in pmc[dom] and "WRITE_SIZE" in pmc[dom]:
        fetch, write = pmc[dom]["FETCH_SIZE"], pmc[dom]["WRITE_SIZE"]
        traffic = {"kernel": dom, "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                   "fetch_size_kib_raw": fetch, "write_size_kib": write,
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128-B request)",
                   "tag": tag, "source_digest": source_digest()}
        json.dump(traffic, open(os.path.join(out, "traffic_latest.json"), "w"), indent=1)
        print(traffic)
    if dom:
        # the dominant kernel's counters per launch, tied to the sources like the traffic figure: bench.py reports ratios of
        # them in roofline.counters (fp32-VALU and LDS utilisation beside the HBM fraction)
        json.dump({"kernel": dom, "counters": pmc[dom], "tag": tag, "source_digest": source_digest()},
                  open(os.path.join(out, "pmc_latest.json"), "w"), indent=1, sort_keys=True)
    print("kernels:", list(pmc))


if __name__ == "__main__":
    main()
