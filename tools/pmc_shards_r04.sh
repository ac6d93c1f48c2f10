#!/bin/bash
# Round 4: HBM-side traffic of the dominant kernel at the shard sizes a rank of BASELINE config 4 holds at N = 1, 2, 4, 8 (1e6, 5e5, 2.5e5, 1.25e5 cells
# of the config-3 surface): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes (they cannot share one), nothing else in the pass.
# tools/summarize_shards.py turns the counter files into profiles/r04_pmc_traffic.json ("shards": keyed by cells), which bench.py looks up by
# the rank's shard size.  Run on the GPU box: gpurun -- bash tools/pmc_shards_r04.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04_shards
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1000000 500000 250000 125000; do
  B="python3 $R/bench.py --cells $n --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$n -- $B > $OUT/fetch_$n.json 2> $OUT/fetch_$n.err || exit 2
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$n -- $B > $OUT/write_$n.json 2> $OUT/write_$n.err || exit 3
  echo "shard $n done"
done
