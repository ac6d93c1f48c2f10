#!/bin/bash
# round 5: cf_main_vah3 after max_j d_j moved to an SGPR pair -- parity tests, then the config-5 bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_vah.py -m gpu -x -q > $O/vah_tests.log 2>&1 || { tail -30 $O/vah_tests.log; exit 1; }
tail -2 $O/vah_tests.log
python bench.py --workload config5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5_dmax_sgpr.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 2; }
python -c "
import json; d=json.load(open('$O/bench_c5_dmax_sgpr.json')); print(d['ms_per_step'], d['kernel_ms'], d['roofline_valu']['frac'], d['roofline_valu']['frac_at_shader_clock'], d['roofline_valu']['shader_clock_ghz'], d['config']['culled_rows_change_no_bit'])"
