#!/bin/bash
# round 5: cf_main_tile3s (variant 9, unit records on the scalar path) against cf_main_tile3e (variant 6) in one process, same resident surface
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
CELLS=${1:-300000}
timeout -k 10 500 python tools/gpu_ab.py --cells $CELLS --rounds 3 --sets "variant=6;variant=9;variant=6,zero_skip=2;variant=9,zero_skip=2;variant=6,waves_per_group=1" > $O/ab_tile3s.log 2>&1
rc=$?
cat $O/ab_tile3s.log | grep -v amdgpu.ids
exit $rc
