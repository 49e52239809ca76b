#!/bin/bash
# Development build of the library with in-kernel timestamps in fused_kf_gru_kernel_v2 (-DOS_FUSED_TS) and one run at the bench
# shape: the kernel prints cycles per step and phase (lane 0 of workgroup 0).  usage (GPU box): bash tools/fused_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/fts
bash $R/tools/ts_lib.sh fts liboptistate_fts.so fused_kernels -DOS_FUSED_TS > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 2
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 1 3     # fused_kf_gru_bf16_kernel<., 3>
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 1 2     # <., 2>
# round 6: the shard shapes (B, trajectories per CU): v2 with 32 per wave, v3 with 1 / 2 / 4 waves per 16-trajectory tile
for bt in "32768 128" "16384 64" "8192 32" "4096 16"; do OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 1 0 $bt; done
