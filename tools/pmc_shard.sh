#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03_125k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cells 125000 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $B > $OUT/f.json 2> $OUT/f.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $B > $OUT/w.json 2> $OUT/w.err || exit 3
B2="python3 $R/bench.py --cells 1000000 --cell-chunks 2296 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_1e6_short -- $B2 > $OUT/f2.json 2> $OUT/f2.err || exit 4
grep -h "cf_main" $OUT/*/*/*counter_collection.csv | awk -F, '{print $0}' | cut -c1-400 | head -20
cat $OUT/f2.json | head -c 600
