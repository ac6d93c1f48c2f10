#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_probe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-clock-probe --no-cull-check --steps 1 --warmup 0"
for v in 7 8 2; do
  rm -rf $OUT/v$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/v$v -- python3 $R/bench.py --workload config2 --variant $v $Q > $OUT/v$v.json 2> $OUT/v$v.err || exit 2
done
rm -rf $OUT/v7b
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/v7b -- python3 $R/bench.py --workload config2 --variant 7 $Q > $OUT/v7b.json 2> $OUT/v7b.err || exit 3
python3 - <<'PY'
import csv,glob,collections,os
out=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/fetch_probe'
for d in sorted(glob.glob(out+'/v*/')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d+'/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:50]][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items():
        if 'cf_main' in k: print(d.split('/')[-2], k, dict(v))
PY
