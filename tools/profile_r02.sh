#!/bin/bash
# Round-2 profiling recipe (run on the GPU box through gpurun): kernel-trace stats of the bench command, then
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ counters in their own pass).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace_bench.json 2> $OUT/trace.err || exit 1
BENCH1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $BENCH1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- $BENCH1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 3
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $BENCH1 > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || exit 4
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_misc -- $BENCH1 > $OUT/pmc_misc.json 2> $OUT/pmc_misc.err || exit 5
find $OUT -name "*.csv" | head -40
# BASELINE config 2 (2+1D, pi/K/p): kernel-trace stats + SQ counters of the unit-strided-lane kernel
BENCH2="python3 $R/bench.py --workload config2 --steps 5 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- $BENCH2 > $OUT/trace_c2_bench.json 2> $OUT/trace_c2.err || exit 6
BENCH21="python3 $R/bench.py --workload config2 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/c2pmc_sq -- $BENCH21 > $OUT/c2pmc_sq.json 2> $OUT/c2pmc_sq.err || exit 7
find $OUT -name "*kernel_stats.csv"
