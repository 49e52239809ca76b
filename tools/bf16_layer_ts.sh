#!/bin/bash
# Development build with in-kernel phase timestamps in the split-bf16 layer kernel (-DOS_LAYER_TS; further -D switches: see the
# kernel's source) linked against the shipped objects, then one run per mode at the reference's model shape.
#   usage (GPU box, after `python -m optistate_amd.build`): bash tools/bf16_layer_ts.sh [extra -D flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/bfts
bash $R/tools/ts_lib.sh bfts liboptistate_bfts.so gru_bf16_kernels -DOS_LAYER_TS "$@" > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_bfts.so python3 tools/bf16_layer_probe.py 65536 20 2>&1 | grep "cycles per step" | awk 'NR%8==1||NR%8==2||NR%8==4' | head -12
