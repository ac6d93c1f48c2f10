#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py --df-mode 4 and 3 (modified equilibrium on the config-3 surface): which of the prep-side kernels
# (cf_prep_feqmod, cf_feqmod_renorm, cf_feqmod_compact, cf_feqmod_linear) the "prep" of kernel_ms is made of
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_feqmod
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 4 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_m$m -- python3 $R/bench.py --df-mode $m --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check > $OUT/bench_m$m.json 2> $OUT/trace_m$m.err || exit 1
  f=$(find $OUT/trace_m$m -name "*kernel_stats.csv" | head -1)
  echo "== df_mode $m"; cut -c1-150 $f | head -12
done
