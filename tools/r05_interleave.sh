#!/bin/bash
# round 5: the two waves of a cf_main_tile3e workgroup on interleaved slots of the mT order (IS3D_LANE_INTERLEAVE=1, developer build) against halves:
# two processes (the switch is read once per process), same box, same surface
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
CELLS=${1:-300000}
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export IS3D_LANE_INTERLEAVE=1; else unset IS3D_LANE_INTERLEAVE; fi
  echo "== interleave $v"
  IS3D_USE_DEV_LIB=1 timeout -k 10 300 python tools/gpu_ab.py --cells $CELLS --rounds 3 --sets "variant=6;variant=6,zero_skip=2;variant=6,waves_per_group=4" 2>&1 | grep -v amdgpu.ids
done > $O/ab_interleave.log 2>&1
cat $O/ab_interleave.log
