#!/bin/bash
# Round-1 kernel-trace stats for the "next" rows (modified-equilibrium kernel, particle sampler); run through gpurun.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r01x
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/feqmod4 -- python3 $R/bench.py --df-mode 4 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check > $OUT/feqmod4_bench.json 2> $OUT/feqmod4.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/feqmod3 -- python3 $R/bench.py --df-mode 3 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check > $OUT/feqmod3_bench.json 2> $OUT/feqmod3.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sampler -- python3 $R/tests/bench_sampler.py --cpu-cells 2000 > $OUT/sampler_bench.json 2> $OUT/sampler.err || exit 3
find $OUT -name "*kernel_stats.csv"
