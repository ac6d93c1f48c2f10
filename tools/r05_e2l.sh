#!/bin/bash
# round 5: cf_main_tile3e<E2L> (variant 12, developer build: the E2 tables built per workgroup in LDS from the staged records) against the default, one process
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
CELLS=${1:-300000}
IS3D_USE_DEV_LIB=1 timeout -k 10 500 python tools/gpu_ab.py --cells $CELLS --rounds 3 --sets "variant=6;variant=12;variant=12,waves_per_group=4;variant=12,waves_per_group=8;variant=6,waves_per_group=4;variant=6,zero_skip=2;variant=12,waves_per_group=4,zero_skip=2" > $O/ab_e2l.log 2>&1
rc=$?
grep -v amdgpu.ids $O/ab_e2l.log
exit $rc
