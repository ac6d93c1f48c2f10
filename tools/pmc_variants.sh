#!/bin/bash
# Dev recipe (GPU box, through gpurun): SQ counters of the main kernel for two kernel variants, culling off, 2e5 cells.
# usage: tools/pmc_variants.sh "3 5" [zero_skip]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_variants
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ZS=${2:-2}
for V in $1; do
  B="python3 $R/bench.py --steps 1 --warmup 0 --cells 200000 --variant $V --zero-skip $ZS --no-cpu-baseline --no-clock-probe --no-cull-check"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/v${V}_a -- $B > $OUT/v${V}_a.json 2> $OUT/v${V}_a.err || exit 1
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES --kernel-trace --output-format csv -d $OUT/v${V}_b -- $B > $OUT/v${V}_b.json 2> $OUT/v${V}_b.err || exit 2
done
python3 - <<PY
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(os.path.join(out, "v*_[ab]"))):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "cf_main" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    print(os.path.basename(d), {k: "%.4g" % (agg[k] / cnt[k]) for k in sorted(agg)}, "launches", dict(cnt).get("SQ_WAVES", list(cnt.values())[:1]))
PY
