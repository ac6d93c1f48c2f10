#!/bin/bash
# SQ counters of cf_main_feqmod (bench.py --df-mode 4, one step): how busy the vector pipe is
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_feqmod
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --df-mode 4 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $B > $OUT/sq.json 2> $OUT/sq.err || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/misc -- $B > $OUT/misc.json 2> $OUT/misc.err || exit 2
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(float); cnt=collections.Counter()
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "cf_main_feqmod" in r["Kernel_Name"]:
            agg[r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[r["Counter_Name"]]+=1
disp=collections.Counter()
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    seen=set()
    for r in csv.DictReader(open(f)):
        if "cf_main_feqmod" in r["Kernel_Name"]: seen.add(r["Dispatch_Id"])
    for k in agg: pass
    print(f.split("/")[-3], "dispatches", len(seen))
for k in sorted(agg): print(k, agg[k])
PY
