#!/bin/bash
# Development build with in-kernel phase timestamps in the split-bf16 layer kernel (-DOS_LAYER_TS; further -D switches: see the
# kernel's source) linked against the shipped objects, then one run per mode at the reference's model shape.
#   usage (GPU box, after `python -m optistate_amd.build`): bash tools/bf16_layer_ts.sh [extra -D flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/bfts; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_dense_rows kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do cp build/$f.o $D/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed -DOS_LAYER_TS "$@" -DOS_BUILD_ID='"ts-build"' -c gru_bf16_kernels.hip -o $D/gru_bf16_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_bfts.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_bfts.so python3 tools/bf16_layer_probe.py 65536 20 2>&1 | grep "cycles per step" | awk 'NR%8==1||NR%8==2||NR%8==4' | head -12
