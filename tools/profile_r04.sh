#!/bin/bash
# Round-4 profiling recipe (GPU box, through gpurun): rocprofv3 --kernel-trace --stats of the bench command, then separate --pmc passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass; SQ counters in their own passes; never together with a trace domain other than --kernel-trace).
#   part a: BASELINE config 3 (default bench) and the modified-equilibrium kernel on the same surface (--df-mode 4)
#   part b: config 5 -- the smooth leg (cf_main_vah3), the sampler leg (--workload config5-sampler), the 2+1D anisotropic-hydro kernel
# tools/summarize_prof.py r04 + tools/summarize_r04_extra.py condense gpurun_out/prof_r04 into profiles/r04_*.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04
PART=${1:-a}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-clock-probe --no-cull-check"
run_stats() { local d=$1; shift; rm -rf $OUT/$d; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$d -- python3 $R/bench.py "$@" --steps 2 --warmup 1 $Q > $OUT/${d}_bench.json 2> $OUT/$d.err || exit 1; }
run_pmc() { local d=$1; local ctr=$2; shift 2; rm -rf $OUT/$d; rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$d -- python3 $R/bench.py "$@" --steps 1 --warmup 0 $Q > $OUT/$d.json 2> $OUT/$d.err || exit 2; }
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
SQ2="GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ "$PART" = "a" ]; then
  run_stats trace
  run_pmc pmc_fetch FETCH_SIZE
  run_pmc pmc_write WRITE_SIZE
  run_pmc pmc_sq "$SQ1"
  run_pmc pmc_misc "$SQ2"
  echo "config 3 done"
  run_stats trace_fq --df-mode 4
  run_pmc fqpmc_fetch FETCH_SIZE --df-mode 4
  run_pmc fqpmc_write WRITE_SIZE --df-mode 4
  run_pmc fqpmc_sq "$SQ1" --df-mode 4
  run_pmc fqpmc_misc "$SQ2" --df-mode 4
  echo "feqmod done"
elif [ "$PART" = "d" ]; then   # only the sampler leg (after a change to its kernels)
  run_stats trace_smp --workload config5-sampler
  run_pmc smppmc_fetch FETCH_SIZE --workload config5-sampler
  run_pmc smppmc_write WRITE_SIZE --workload config5-sampler
  run_pmc smppmc_sq "$SQ1" --workload config5-sampler
  run_pmc smppmc_misc "$SQ2" --workload config5-sampler
elif [ "$PART" = "e" ]; then   # the 2+1D kernels on the config-2 surface: delta-f (BASELINE config 2) and modified equilibrium (df_mode 4)
  run_stats trace_c2 --workload config2
  run_pmc c2pmc_fetch FETCH_SIZE --workload config2
  run_pmc c2pmc_write WRITE_SIZE --workload config2
  run_pmc c2pmc_sq "$SQ1" --workload config2
  run_pmc c2pmc_misc "$SQ2" --workload config2
  run_stats trace_c2fq --workload config2 --df-mode 4
  run_pmc c2fqpmc_sq "$SQ1" --workload config2 --df-mode 4
  run_pmc c2fqpmc_misc "$SQ2" --workload config2 --df-mode 4
elif [ "$PART" = "c" ]; then   # only the kernel-trace stats of the anisotropic-hydro workloads (after a change to cf_prep_vah)
  run_stats trace_c5 --workload config5
  run_stats trace_v2 --workload config5 --dimension 2
  run_pmc v2pmc_sq "$SQ1" --workload config5 --dimension 2
else
  run_stats trace_c5 --workload config5
  run_pmc c5pmc_fetch FETCH_SIZE --workload config5
  run_pmc c5pmc_write WRITE_SIZE --workload config5
  run_pmc c5pmc_sq "$SQ1" --workload config5
  run_pmc c5pmc_misc "$SQ2" --workload config5
  echo "config 5 smooth leg done"
  run_stats trace_smp --workload config5-sampler
  run_pmc smppmc_fetch FETCH_SIZE --workload config5-sampler
  run_pmc smppmc_write WRITE_SIZE --workload config5-sampler
  run_pmc smppmc_sq "$SQ1" --workload config5-sampler
  run_pmc smppmc_misc "$SQ2" --workload config5-sampler
  echo "sampler done"
  run_stats trace_v2 --workload config5 --dimension 2
  run_pmc v2pmc_sq "$SQ1" --workload config5 --dimension 2
  echo "2+1D vah done"
fi
find $OUT -name "*kernel_stats.csv" | head -20
