#!/bin/bash
# Same-box A/B of the shipped library against a development build with extra -D flags on some sources, alternating processes.
# usage (GPU box): bash tools/ab_libs.sh <source[,source]> "<-D flags>" [B] [tile] [pairs]      e.g. fused_kernels "-DOSF_V2_PREFETCH_BLOCK" 65536 0 3
R=${GRAFT_REPO_ROOT:-/root/repo}; SRC=$1; FLAGS=$2; B=${3:-65536}; TILE=${4:-0}; N=${5:-3}
LIB=$(bash $R/tools/ts_lib.sh ab_$$ liboptistate_ab.so $SRC $FLAGS | tail -1) || exit 1
cd $R
for i in $(seq $N); do
  python3 tools/time_fused.py $B $TILE
  OPTISTATE_HIP_LIB=$LIB python3 tools/time_fused.py $B $TILE
done
rm -rf $R/build_ab/ab_$$
