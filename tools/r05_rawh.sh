#!/bin/bash
# round 5: cf_main_tile3e<RAWH> (variant 11, developer build: raw header values as FMA operands) against the default, one process
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
CELLS=${1:-300000}
IS3D_USE_DEV_LIB=1 timeout -k 10 500 python tools/gpu_ab.py --cells $CELLS --rounds 5 --df ${2:-2} --sets "variant=6;variant=11;variant=6,zero_skip=2;variant=11,zero_skip=2" > $O/ab_rawh_df${2:-2}.log 2>&1
rc=$?
grep -v amdgpu.ids $O/ab_rawh_df${2:-2}.log
exit $rc
