#!/bin/bash
# Development build with in-kernel timestamps in kf_run_sym_kernel (-DOS_SYM_TS); the kernel prints cycles per step and phase.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/sts
bash $R/tools/ts_lib.sh sts liboptistate_sts.so kf_kernels -DOS_SYM_TS > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_sts.so python3 tools/run_fused_once.py 2 2>&1 | grep "kf_run_sym cycles"
