#!/bin/bash
# the -m gpu suite as the driver runs it, with timings of the slowest tests
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
python -m pytest tests/ -x -q -m gpu --durations=15 > $O/gpu_tests.log 2>&1
rc=$?
tail -40 $O/gpu_tests.log
exit $rc
