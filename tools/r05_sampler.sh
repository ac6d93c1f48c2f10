#!/bin/bash
# round 5: the sampler with blocked running sums -- list parity tests, then the sampler bench line
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_sampler.py tests/test_gpu_fuzz.py tests/test_gpu_cli.py -m gpu -x -q -k "sampler or cli" > $O/sampler_tests.log 2>&1 || { tail -30 $O/sampler_tests.log; exit 1; }
tail -2 $O/sampler_tests.log
python bench.py --workload config5-sampler --steps 10 --warmup 2 > $O/bench_sampler_blocked.json 2> $O/bench_sampler.err || { tail -20 $O/bench_sampler.err; exit 2; }
python -c "
import json; d=json.load(open('$O/bench_sampler_blocked.json')); print(d['ms_per_step'], d['value'], d['kernel_ms'], d['cpu_baseline'].get('same_list_on_the_slice'), d['same_list_as_host_entry'], d['particles_per_step'])"
