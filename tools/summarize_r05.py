#!/usr/bin/env python3
"""gpurun_out/prof_r05 (tools/profile_r05.sh) -> the tracked round-5 summaries under profiles/.  One generic pass per profiled command
(`tag` = "" default config 3, "_baryon", "_df1", "_c5"):
  r05_kernel_stats<tag>.csv        rocprofv3 --kernel-trace --stats of `bench.py <args>`
  r05_pmc_summary<tag>.csv         per kernel and counter, averaged per launch (separate --pmc passes)
  r05_bench<tag>_under_rocprof.json   the bench line of the traced run (HIP-event kernel time in the same process as rocprofv3's)
  r05_pmc_traffic.json[key]        HBM-side bytes per launch of the main kernel, corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE
                                   in KiB; on gfx950 FETCH_SIZE counts half the bytes of a 16-B-per-lane stream -> x 2) -- what bench.py's `roofline.traffic` quotes
usage: summarize_r05.py [tag ...]     (default: every tag with a trace directory)"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r05")
dst = os.path.join(ROOT, "profiles")
KEY = {"": "config3", "_baryon": "config3_baryon", "_df1": "config3_df1", "_c5": "config5"}


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None


def one(tag):
    f = newest(os.path.join(src, "trace" + tag, "*", "*_kernel_stats.csv"))
    if not f:
        return False
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, "r05_kernel_stats%s.csv" % tag), "w", newline="") as o:
        w = csv.writer(o)
        for r in rows:
            r[0] = r[0][:140]
            w.writerow(r)
    main_avg_ns = next((float(r[3]) for r in rows[1:] if "cf_main" in r[0]), None)
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for d in sorted(glob.glob(os.path.join(src, "pmc%s_*" % tag))):
        g = newest(os.path.join(d, "*", "*_counter_collection.csv")) if os.path.isdir(d) else None
        if not g:
            continue
        for r in csv.DictReader(open(g)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if name.startswith("is3d::"):
                agg[(name[:70], r["Counter_Name"])] += float(r["Counter_Value"])
                cnt[(name[:70], r["Counter_Name"])] += 1
    if agg:
        with open(os.path.join(dst, "r05_pmc_summary%s.csv" % tag), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["kernel", "counter", "launches", "value_per_launch"])
            for k in sorted(agg):
                w.writerow([k[0], k[1], cnt[k], "%.6g" % (agg[k] / cnt[k])])
    bj = os.path.join(src, "trace%s_bench.json" % tag)
    b = json.load(open(bj)) if os.path.exists(bj) and os.path.getsize(bj) else None
    if b:
        json.dump(b, open(os.path.join(dst, "r05_bench%s_under_rocprof.json" % tag), "w"), indent=1)
    main = {k[1]: agg[k] / cnt[k] for k in agg if "cf_main" in k[0]}
    if "FETCH_SIZE" in main and "WRITE_SIZE" in main:
        tp = os.path.join(dst, "r05_pmc_traffic.json")
        t = json.load(open(tp)) if os.path.exists(tp) else {}
        gui = main.get("GRBM_GUI_ACTIVE")
        keep = {kk: vv for kk, vv in t.get(KEY[tag], {}).items() if kk in ("shards", "shards_note")}
        t[KEY[tag]] = dict(keep, cells=b["config"]["cells_per_gpu"] if b else None, kernel=b["config"]["kernel"] if b else None,
                           FETCH_SIZE_KiB=main["FETCH_SIZE"], WRITE_SIZE_KiB=main["WRITE_SIZE"], fetch_correction=2.0,
                           hbm_bytes_per_launch=(2.0 * main["FETCH_SIZE"] + main["WRITE_SIZE"]) * 1024.0,
                           rocprof_avg_kernel_ms=main_avg_ns / 1e6 if main_avg_ns else None,
                           bench_hip_event_kernel_ms=b["kernel_ms"]["main"] if b else None,
                           effective_clock_GHz=(gui / 8.0 / (main_avg_ns * 1e-9) / 1e9) if (main_avg_ns and gui) else None,
                           valu_busy_frac_of_simd_cycles=(main["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * gui / 8.0)) if gui and main.get("SQ_ACTIVE_INST_VALU") else None,
                           sq_counters_per_launch={k: v for k, v in sorted(main.items()) if k not in ("FETCH_SIZE", "WRITE_SIZE")})
        json.dump(t, open(tp, "w"), indent=1)
        print(tag or "(default)", json.dumps({k: v for k, v in t[KEY[tag]].items() if k != "sq_counters_per_launch"}, indent=1))
    return True


for tag in (sys.argv[1:] or list(KEY)):
    tag = "" if tag in ("default", "a") else tag
    if not one(tag):
        print("no trace for tag %r" % tag)
