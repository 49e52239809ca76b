#!/bin/bash
# Development build with in-kernel timestamps in kf_run_sym_kernel (-DOS_SYM_TS); the kernel prints cycles per step and phase.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/sts; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  [ $f = kf_kernels ] && X=-DOS_SYM_TS || X=
  [ -f $D/$f.o -a $f != kf_kernels ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"ts-build"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_sts.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_sts.so python3 tools/run_fused_once.py 2 2>&1 | grep "kf_run_sym cycles"
