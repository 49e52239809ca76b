#!/bin/bash
# Development build of the library with in-kernel timestamps in fused_kf_gru_kernel_v2 (-DOS_FUSED_TS) and one run at the bench
# shape: the kernel prints cycles per step and phase (lane 0 of workgroup 0).  usage (GPU box): bash tools/fused_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/fts; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  [ $f = fused_kernels ] && X=-DOS_FUSED_TS || X=
  [ -f $D/$f.o -a $f != fused_kernels ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"ts-build"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_fts.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 2
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 1 3     # fused_kf_gru_bf16_kernel<., 3>
OPTISTATE_HIP_LIB=$D/liboptistate_fts.so python3 tools/run_fused_once.py 1 2     # <., 2>
