#!/bin/bash
# Development build of the library with in-kernel timestamps in mpc_solve_quad_kernel (-DOSQ_TS) and one --mode mpc bench pass:
# cycles per wave-iteration and phase (lane 0 of workgroup 0).  usage (GPU box): bash tools/quad_ts.sh [extra -D flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/qts
bash $R/tools/ts_lib.sh qts liboptistate_qts.so mpc_quad -DOSQ_TS $* > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_qts.so python3 bench.py --mode mpc --steps 1 --warmup 1 --cpu-seconds 0 2>&1 | grep "cycles per" | tail -3
