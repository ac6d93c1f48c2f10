#!/usr/bin/env python3
"""Condense a tools/profile_rNN.sh output directory (gpurun_out/prof_rNN) into the tracked files under profiles/:
  rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `bench.py`
  rNN_pmc_summary.csv    per-kernel, per-counter values averaged per launch (separate --pmc passes)
  rNN_pmc_traffic.json   HBM-side bytes per launch of the main kernel, corrected as MI355X_MICROARCH.md section HBM
                         prescribes (FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 1/2 of the bytes
                         of a 16-B-per-lane coalesced stream -> x2; WRITE_SIZE taken as is).
usage: summarize_prof.py r01 config3"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "config3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# gpurun merges every call's files into the same local directory: take the newest of each kind
stats = max(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, tag + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)
main_avg_ns = None
for r in rows[1:]:
    if "cf_main" in r[0]:
        main_avg_ns = float(r[3])

agg = collections.defaultdict(float)
cnt = collections.Counter()
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in [max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)]:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not name.startswith("is3d::"):
                continue
            k = (name[:60], r["Counter_Name"])
            agg[k] += float(r["Counter_Value"])
            cnt[k] += 1
with open(os.path.join(dst, tag + "_pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "launches", "value_per_launch"])
    for k in sorted(agg):
        w.writerow([k[0], k[1], cnt[k], "%.6g" % (agg[k] / cnt[k])])


def per_launch(kern, counter):
    """value per launch of the LAST matching kernel name (stale merged directories may hold older kernels)"""
    hit = None
    for k in sorted(agg):
        if kern in k[0] and k[1] == counter:
            hit = agg[k] / cnt[k]
    return hit


fetch, write = per_launch("cf_main", "FETCH_SIZE"), per_launch("cf_main", "WRITE_SIZE")
bench = json.load(open(os.path.join(src, "trace_bench.json")))
gui = per_launch("cf_main", "GRBM_GUI_ACTIVE")
out = {workload: dict(cells=bench["config"]["cells_per_gpu"], kernel=bench["config"]["kernel"],
                      FETCH_SIZE_KiB=fetch, WRITE_SIZE_KiB=write, fetch_correction=2.0,
                      hbm_bytes_per_launch=(2.0 * fetch + write) * 1024.0,
                      rocprof_avg_kernel_ms=main_avg_ns / 1e6 if main_avg_ns else None,
                      bench_hip_event_kernel_ms=bench["kernel_ms"]["main"],
                      GRBM_GUI_ACTIVE=gui,
                      effective_clock_GHz=(gui / 8.0 / (main_avg_ns * 1e-9) / 1e9) if (main_avg_ns and gui) else None,
                      SQ_INSTS_VALU=per_launch("cf_main", "SQ_INSTS_VALU"),
                      SQ_ACTIVE_INST_VALU_quadcycles=per_launch("cf_main", "SQ_ACTIVE_INST_VALU"))}
tp = os.path.join(dst, tag + "_pmc_traffic.json")
old = json.load(open(tp)) if os.path.exists(tp) else {}
for k_, v_ in out.items():   # keep what other tools put beside it (tools/summarize_shards.py: "shards")
    old[k_] = dict({kk: vv for kk, vv in old.get(k_, {}).items() if kk in ("shards", "shards_note")}, **v_)
json.dump(old, open(tp, "w"), indent=1)
json.dump(bench, open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
# optional second workload of the same script (tools/profile_r02.sh: BASELINE config 2)
c2 = glob.glob(os.path.join(src, "trace_c2", "*", "*_kernel_stats.csv"))
if c2:
    rows = list(csv.reader(open(max(c2, key=os.path.getmtime))))
    with open(os.path.join(dst, tag + "_kernel_stats_config2.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:120]
            w.writerow(r)
    b2 = json.load(open(os.path.join(src, "trace_c2_bench.json")))
    json.dump(b2, open(os.path.join(dst, tag + "_bench_config2_under_rocprof.json"), "w"), indent=1)
    agg2, cnt2 = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(src, "c2pmc_sq", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "cf_main" in r["Kernel_Name"]:
                agg2[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt2[r["Counter_Name"]] += 1
    old = json.load(open(tp))
    fw2 = {}
    for ctr, sub in (("FETCH_SIZE", "c2pmc_fetch"), ("WRITE_SIZE", "c2pmc_write")):   # separate passes, one launch of the main kernel each
        fs = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
        if fs:
            v = [float(r["Counter_Value"]) for r in csv.DictReader(open(max(fs, key=os.path.getmtime))) if "cf_main" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
            if v:
                fw2[ctr] = sum(v) / len(v)
    traffic2 = {}
    if len(fw2) == 2:
        traffic2 = dict(FETCH_SIZE_KiB=fw2["FETCH_SIZE"], WRITE_SIZE_KiB=fw2["WRITE_SIZE"], fetch_correction=2.0,
                        hbm_bytes_per_launch=(2.0 * fw2["FETCH_SIZE"] + fw2["WRITE_SIZE"]) * 1024.0)
    old["config2"] = dict(cells=b2["config"]["cells_per_gpu"], kernel=b2["config"]["kernel"], kernel_variant=b2["config"]["kernel_variant"],
                          bench_hip_event_kernel_ms=b2["kernel_ms"]["main"], **traffic2,
                          rocprof_avg_kernel_ms=[float(r[3]) / 1e6 for r in rows[1:] if "cf_main" in r[0]][:1],
                          sq_counters_per_launch={k: agg2[k] / cnt2[k] for k in sorted(agg2)})
    json.dump(old, open(tp, "w"), indent=1)

# optional third workload (tools/profile_r03.sh: the smooth leg of BASELINE config 5, cf_main_vah3): kernel stats, every counter of
# its c5pmc_* passes, and the same traffic derivation
c5 = glob.glob(os.path.join(src, "trace_c5", "*", "*_kernel_stats.csv"))
if c5:
    rows = list(csv.reader(open(max(c5, key=os.path.getmtime))))
    with open(os.path.join(dst, tag + "_kernel_stats_vah.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:120]
            w.writerow(r)
    b5 = json.load(open(os.path.join(src, "trace_c5_bench.json")))
    json.dump(b5, open(os.path.join(dst, tag + "_bench_config5_under_rocprof.json"), "w"), indent=1)
    agg5, cnt5 = collections.defaultdict(float), collections.Counter()
    for d in sorted(glob.glob(os.path.join(src, "c5pmc_*"))):
        if not os.path.isdir(d):
            continue
        for f in [max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)]:   # gpurun merges every call's files: newest only
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"].split("(")[0].replace("void ", "")
                if name.startswith("is3d::"):
                    agg5[(name[:60], r["Counter_Name"])] += float(r["Counter_Value"])
                    cnt5[(name[:60], r["Counter_Name"])] += 1
    with open(os.path.join(dst, tag + "_pmc_summary_vah.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "value_per_launch"])
        for k in sorted(agg5):
            w.writerow([k[0], k[1], cnt5[k], "%.6g" % (agg5[k] / cnt5[k])])
    main5 = {k[1]: agg5[k] / cnt5[k] for k in agg5 if "cf_main" in k[0]}
    avg5 = [float(r[3]) for r in rows[1:] if "cf_main" in r[0]][:1]
    old = json.load(open(tp))
    if "FETCH_SIZE" in main5 and "WRITE_SIZE" in main5:
        old["config5"] = dict(cells=b5["config"]["cells_per_gpu"], kernel=b5["config"]["kernel"], FETCH_SIZE_KiB=main5["FETCH_SIZE"],
                              WRITE_SIZE_KiB=main5["WRITE_SIZE"], fetch_correction=2.0,
                              hbm_bytes_per_launch=(2.0 * main5["FETCH_SIZE"] + main5["WRITE_SIZE"]) * 1024.0,
                              rocprof_avg_kernel_ms=avg5[0] / 1e6 if avg5 else None, bench_hip_event_kernel_ms=b5["kernel_ms"]["main"],
                              sq_counters_per_launch={k: v for k, v in sorted(main5.items()) if k not in ("FETCH_SIZE", "WRITE_SIZE")})
    json.dump(old, open(tp, "w"), indent=1)
