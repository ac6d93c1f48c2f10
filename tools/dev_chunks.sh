#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for c in 0 438 876 1314 1752 0; do
python3 $R/bench.py --cell-chunks $c --steps 5 --warmup 2 --no-cpu-baseline --no-cull-check --no-clock-probe 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunks', $c, 'step %.2f'%d['ms_per_step'], {k:round(v,2) for k,v in d['kernel_ms'].items() if v}, 'culled %.4f'%d['roofline_valu']['wave_rows_culled_frac'], 'frac %.4f'%d['roofline_valu']['frac'], 'ws %.1f'%d['config']['workspace_GB'])"
done
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r03_chunks; rm -rf $OUT; mkdir -p $OUT
for c in 876 1752; do
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f$c -- python3 $R/bench.py --cell-chunks $c --steps 1 --warmup 0 --no-cpu-baseline --no-clock-probe --no-cull-check > $OUT/f$c.json 2> $OUT/f$c.err
grep -h "cf_main" $OUT/f$c/*/*counter_collection.csv | awk -F'"FETCH_SIZE",' '{print "chunks '$c' FETCH_SIZE KiB", $2}' | cut -d, -f1-1 | head -2
done
