#!/bin/bash
# Round-3 profiling recipe (run on the GPU box through gpurun): kernel-trace stats of the bench command, then separate --pmc passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ counters in their own pass) -- for BASELINE config 3 (default bench) and for
# the smooth leg of config 5 (anisotropic hydro, cf_main_vah3).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03
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
echo "config 3 done"
# BASELINE config 5, smooth leg: kernel-trace stats + the same counter passes for cf_main_vah3
B5="python3 $R/bench.py --workload config5 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5 -- $B5 > $OUT/trace_c5_bench.json 2> $OUT/trace_c5.err || exit 6
B51="python3 $R/bench.py --workload config5 --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/c5pmc_fetch -- $B51 > $OUT/c5pmc_fetch.json 2> $OUT/c5pmc_fetch.err || exit 7
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/c5pmc_write -- $B51 > $OUT/c5pmc_write.json 2> $OUT/c5pmc_write.err || exit 8
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/c5pmc_sq -- $B51 > $OUT/c5pmc_sq.json 2> $OUT/c5pmc_sq.err || exit 9
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/c5pmc_misc -- $B51 > $OUT/c5pmc_misc.json 2> $OUT/c5pmc_misc.err || exit 10
find $OUT -name "*kernel_stats.csv"
