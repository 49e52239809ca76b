#!/bin/bash
# Development build with in-kernel timestamps in the split-bf16 layer kernel (-DOS_LAYER_TS) + one run per mode at the reference's
# model shape; then a counter pass of the shipped build.   usage (GPU box): bash tools/bf16_layer_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/bfts; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_dense_rows kf_step gru_kernels gru_bf16_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  X=; [ $f = gru_bf16_kernels ] && X=-DOS_LAYER_TS; [ $f = kf_rows_kernel ] && X="-fno-slp-vectorize"
  if [ -f $R/optistate_amd/csrc/build/$f.o -a $f != gru_bf16_kernels ]; then cp $R/optistate_amd/csrc/build/$f.o $D/$f.o
  else /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"ts-build"' -c $f.hip -o $D/$f.o & fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_bfts.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_bfts.so python3 tools/bf16_layer_probe.py 65536 20 2>&1 | grep "cycles per step\|mode" 
