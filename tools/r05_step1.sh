#!/bin/bash
# round 5, first GPU call: the new bench contract cases, the cycle accounting of cf_main_tile3e on the current build (PROF instantiation of the developer library),
# the full-size bench lines of --include-baryon and --df-mode 1
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_bench.py -m gpu -x -q > $O/bench_tests.log 2>&1 || { tail -30 $O/bench_tests.log; exit 1; }
tail -3 $O/bench_tests.log
IS3D_USE_DEV_LIB=1 IS3D_DEV_PROF=1 python tools/gpu_ab.py --cells 1000000 --rounds 1 --sets "variant=6;variant=6,zero_skip=2" > $O/cycle_accounting_tile3e.out 2> $O/cycle_accounting_tile3e.log || { tail -20 $O/cycle_accounting_tile3e.log; exit 2; }
cat $O/cycle_accounting_tile3e.out; grep prof3e $O/cycle_accounting_tile3e.log | tail -4
python bench.py --include-baryon --steps 5 --warmup 1 > $O/bench_baryon.json 2> $O/bench_baryon.err || { tail -20 $O/bench_baryon.err; exit 3; }
python bench.py --df-mode 1 --steps 5 --warmup 1 > $O/bench_df1.json 2> $O/bench_df1.err || { tail -20 $O/bench_df1.err; exit 4; }
python - <<'PY'
import json
for n in ("baryon", "df1"):
    d = json.load(open("gpurun_out/r05/bench_%s.json" % n))
    print(n, d["ms_per_step"], d["value"], d["kernel_ms"], d["roofline_valu"]["frac"], d["roofline_valu"]["frac_at_shader_clock"], d["config"]["species_classes_evaluated"], d["config"]["culled_rows_change_no_bit"], d["config"]["workspace_GB"])
PY
