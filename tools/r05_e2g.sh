#!/bin/bash
# round 5: cf_main_tile3e<E2G> (variant 10, developer build: the E2 column from global memory, records-only LDS batches) against the default, one process
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
CELLS=${1:-300000}
IS3D_USE_DEV_LIB=1 timeout -k 10 500 python tools/gpu_ab.py --cells $CELLS --rounds 3 --sets "variant=6;variant=10;variant=10,waves_per_group=1;variant=10,waves_per_group=4;variant=6,zero_skip=2;variant=10,zero_skip=2;variant=10,waves_per_group=1,zero_skip=2" > $O/ab_e2g.log 2>&1
rc=$?
grep -v amdgpu.ids $O/ab_e2g.log
exit $rc
