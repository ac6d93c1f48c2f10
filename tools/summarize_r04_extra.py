#!/usr/bin/env python3
"""gpurun_out/prof_r04 (tools/profile_r04.sh) -> the tracked round-4 summaries that tools/summarize_prof.py does not make:
  profiles/r04_kernel_stats_feqmod.csv, r04_pmc_summary_feqmod.csv     the modified-equilibrium kernel (df_mode 4) on the config-3 surface
  profiles/r04_kernel_stats_sampler.csv, r04_pmc_summary_sampler.csv   bench.py --workload config5-sampler
  profiles/r04_pmc_sampler.json      what bench.py's sampler line quotes: HBM-side bytes per step (FETCH_SIZE x 2 + WRITE_SIZE over all sampler kernels) and, per
                                     kernel, the share of SIMD cycles that issue a VALU instruction
  profiles/r04_kernel_stats_vah2d.csv, r04_pmc_summary_vah2d.csv       bench.py --workload config5 --dimension 2
  and the feqmod / sampler entries of profiles/r04_pmc_traffic.json."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_r04")
dst = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = glob.glob(pattern)
    return max(fs, key=os.path.getmtime) if fs else None


def copy_stats(trace_dir, name):
    f = newest(os.path.join(src, trace_dir, "*", "*_kernel_stats.csv"))
    if not f:
        return None
    rows = list(csv.reader(open(f)))
    with open(os.path.join(dst, name), "w", newline="") as o:
        w = csv.writer(o)
        for r in rows:
            r[0] = r[0][:140]
            w.writerow(r)
    return rows


def counters(prefix):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for d in sorted(glob.glob(os.path.join(src, prefix + "*"))):
        f = newest(os.path.join(d, "*", "*_counter_collection.csv")) if os.path.isdir(d) else None
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not (name.startswith("is3d::") or "rocprim" in name):
                continue
            name = name[:70]
            agg[(name, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[(name, r["Counter_Name"])] += 1
    return agg, cnt


def write_summary(agg, cnt, name):
    with open(os.path.join(dst, name), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "counter", "launches", "value_per_launch"])
        for k in sorted(agg):
            w.writerow([k[0], k[1], cnt[k], "%.6g" % (agg[k] / cnt[k])])


tp = os.path.join(dst, "r04_pmc_traffic.json")
traffic = json.load(open(tp)) if os.path.exists(tp) else {}

# ---- modified equilibrium
rows = copy_stats("trace_fq", "r04_kernel_stats_feqmod.csv")
agg, cnt = counters("fqpmc_")
if agg:
    write_summary(agg, cnt, "r04_pmc_summary_feqmod.csv")
    main = {k[1]: agg[k] / cnt[k] for k in agg if "cf_main_feqmod" in k[0]}
    bj = os.path.join(src, "trace_fq_bench.json")
    b = json.load(open(bj)) if os.path.exists(bj) else None
    avg = [float(r[3]) for r in (rows or [])[1:] if "cf_main_feqmod" in r[0]][:1]
    if "FETCH_SIZE" in main and "WRITE_SIZE" in main:
        traffic["config3_feqmod4"] = dict(cells=b["config"]["cells_per_gpu"] if b else None, kernel="cf_main_feqmod", FETCH_SIZE_KiB=main["FETCH_SIZE"], WRITE_SIZE_KiB=main["WRITE_SIZE"],
                                          fetch_correction=2.0, hbm_bytes_per_launch=(2.0 * main["FETCH_SIZE"] + main["WRITE_SIZE"]) * 1024.0,
                                          rocprof_avg_kernel_ms=avg[0] / 1e6 if avg else None, bench_hip_event_kernel_ms=b["kernel_ms"]["main"] if b else None,
                                          sq_counters_per_launch={k: v for k, v in sorted(main.items()) if k not in ("FETCH_SIZE", "WRITE_SIZE")})
    if b:
        json.dump(b, open(os.path.join(dst, "r04_bench_config3_feqmod4_under_rocprof.json"), "w"), indent=1)

# ---- sampler
rows = copy_stats("trace_smp", "r04_kernel_stats_sampler.csv")
agg, cnt = counters("smppmc_")
if agg:
    write_summary(agg, cnt, "r04_pmc_summary_sampler.csv")
    per_kernel = collections.defaultdict(dict)
    for (name, ctr), v in agg.items():
        per_kernel[name][ctr] = v / cnt[(name, ctr)]
        per_kernel[name]["launches_in_pass"] = cnt[(name, ctr)]
    bj = os.path.join(src, "trace_smp_bench.json")
    b = json.load(open(bj)) if os.path.exists(bj) else None
    # bytes per STEP: every launch of every sampler kernel of the profiled process / the number of plan executes in it
    # (bench.py --steps 1 --warmup 0: one count-only execute + one timed step + the host entry's two calls = 4 executes of the kernels)
    tot_f = sum(agg[k] for k in agg if k[1] == "FETCH_SIZE")
    tot_w = sum(agg[k] for k in agg if k[1] == "WRITE_SIZE")
    dens_launches = max([cnt[k] for k in cnt if "cf_sampler_density" in k[0] and k[1] == "FETCH_SIZE"] or [1])
    out = dict(source="profiles/r04_pmc_summary_sampler.csv (rocprofv3 --pmc passes of bench.py --workload config5-sampler; not measured in this run)",
               executes_in_profiled_process=dens_launches,
               hbm_bytes_per_step=(2.0 * tot_f + tot_w) * 1024.0 / dens_launches,
               kernels={})
    for name, c in sorted(per_kernel.items()):
        e = {}
        if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            e["valu_busy_frac_of_simd_cycles"] = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        for k in ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE"):
            if k in c:
                e[k] = c[k]
        out["kernels"][name] = e
    json.dump(out, open(os.path.join(dst, "r04_pmc_sampler.json"), "w"), indent=1)
    if b:
        json.dump(b, open(os.path.join(dst, "r04_bench_sampler_under_rocprof.json"), "w"), indent=1)

# ---- 2+1D anisotropic hydro
copy_stats("trace_v2", "r04_kernel_stats_vah2d.csv")
agg, cnt = counters("v2pmc_")
if agg:
    write_summary(agg, cnt, "r04_pmc_summary_vah2d.csv")
bj = os.path.join(src, "trace_v2_bench.json")
if os.path.exists(bj):
    json.dump(json.load(open(bj)), open(os.path.join(dst, "r04_bench_config5_dim2_under_rocprof.json"), "w"), indent=1)
json.dump(traffic, open(tp, "w"), indent=1)
print("wrote round-4 extra summaries")
