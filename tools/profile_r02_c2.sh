#!/bin/bash
# The BASELINE config 2 half of tools/profile_r02.sh on its own (same output directory).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH2="python3 $R/bench.py --workload config2 --steps 5 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- $BENCH2 > $OUT/trace_c2_bench.json 2> $OUT/trace_c2.err || exit 6
BENCH21="python3 $R/bench.py --workload config2 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/c2pmc_sq -- $BENCH21 > $OUT/c2pmc_sq.json 2> $OUT/c2pmc_sq.err || exit 7
