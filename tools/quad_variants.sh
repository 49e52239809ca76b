#!/bin/bash
# A/B of development builds of mpc_quad.hip (extra -D flags) against the shipped library on one box.
#   build (container):  bash tools/quad_variants.sh build <name> "<-D flags>" [<name> "<flags>" ...]   -> build_ab/qv_<name>/
#   run (GPU box):      bash tools/quad_variants.sh run [passes]                                       -> ms per QP launch, every variant
R=${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 bench.py --mode mpc --steps 3 --warmup 1 --cpu-seconds 0 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s mpc launch %.4f ms   %.3e steps/s  iters mean %.2f max %d' % ('$1', d['kernels']['mpc']['ms_per_launch'], d['value'], d['qp_iterations_mean'], d['qp_iterations_max']))"; }
if [ "$1" = build ]; then
  shift
  while [ $# -ge 2 ]; do
    bash $R/tools/ts_lib.sh qv_$1 liboptistate_qv.so mpc_quad $2 | tail -1; shift 2
  done
else
  cd $R
  for p in $(seq ${2:-2}); do
    line shipped
    for d in build_ab/qv_*/; do n=$(basename $d); OPTISTATE_HIP_LIB=$R/$d/liboptistate_qv.so line ${n#qv_}; done
  done
fi
