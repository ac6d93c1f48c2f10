#!/bin/bash
# Round-5 profiling recipe (GPU box, through gpurun): rocprofv3 --kernel-trace --stats of the bench command, then separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass; SQ counters in their own passes; never together with a trace domain other than --kernel-trace).
#   part a: BASELINE config 3 (default bench)
#   part b: config 3 with include_baryon = 1 (SURVEY.md 8f rank 1 at the bench)      -> profiles/r05_*_baryon*
#   part c: config 3 with df_mode 1 (14-moment: config 1's physics at config-3 size) -> profiles/r05_*_df1*
#   part v: config 5 (cf_main_vah3)
# tools/summarize_r05.py condenses gpurun_out/prof_r05 into profiles/r05_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r05
PART=${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-clock-probe --no-cull-check"
run_stats() { local d=$1; shift; rm -rf $OUT/$d; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$d -- python3 $R/bench.py "$@" --steps 8 --warmup 1 $Q > $OUT/${d}_bench.json 2> $OUT/$d.err || exit 1; }
run_pmc() { local d=$1; local ctr=$2; shift 2; rm -rf $OUT/$d; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$d -- python3 $R/bench.py "$@" --steps 1 --warmup 0 $Q > $OUT/$d.json 2> $OUT/$d.err || exit 2; }
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
SQ2="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
full() {   # full <tag> <bench args...>
  local t=$1; shift
  run_stats trace$t "$@"
  run_pmc pmc${t}_fetch FETCH_SIZE "$@"
  run_pmc pmc${t}_write WRITE_SIZE "$@"
  run_pmc pmc${t}_sq "$SQ1" "$@"
  run_pmc pmc${t}_misc "$SQ2" "$@"
}
case "$PART" in
  a) full "" ;;
  b) full _baryon --include-baryon ;;
  c) full _df1 --df-mode 1 ;;
  v) full _c5 --workload config5 ;;
  *) echo "unknown part $PART"; exit 3 ;;
esac
find $OUT -name "*kernel_stats.csv" | head -20
