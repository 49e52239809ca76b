#!/bin/bash
# Development build of the library with in-kernel timestamps in the GRU layer kernels (-DOS_LAYER_TS) and runs of the reference's
# real model shape GRU(188,128,4) at B = 65,536, T = 100 with gru_layer_stage_kernel (default) and with gru_layer_kernel<2,2>
# (OS_GRU_STAGE=0): each layer prints its cycles per step and phase.   usage (GPU box): bash tools/layer_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/lts
bash $R/tools/ts_lib.sh lts liboptistate_lts.so gru_kernels -DOS_LAYER_TS > /dev/null || exit 1
cd $R
echo "== gru_layer_stage_kernel"; OPTISTATE_HIP_LIB=$D/liboptistate_lts.so python3 tools/run_ref_shape_once.py 1
echo "== gru_layer_kernel<2,2> (OS_GRU_STAGE=0)"; OS_GRU_STAGE=0 OPTISTATE_HIP_LIB=$D/liboptistate_lts.so python3 tools/run_ref_shape_once.py 1
echo "== training forward, 8192 x 10 (OS_GRU_AHEAD=0: gru_layer_split_kernel)"; OS_GRU_AHEAD=0 OPTISTATE_HIP_LIB=$D/liboptistate_lts.so python3 tools/train_fwd_ts.py 2>&1 | grep "cycles per step" | tail -4
