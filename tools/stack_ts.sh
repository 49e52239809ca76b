#!/bin/bash
# In-kernel timestamps of gru_stack_kernel's workgroups (-DOS_LAYER_TS build of gru_kernels.hip): GRU(188,128,4), B = 64, T = 10.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/lts; mkdir -p $D
cd $R/optistate_amd/csrc
for f in capi kf_kernels kf_rows_kernel kf_step gru_kernels fused_kernels gru_train_kernels vit_kernels mpc_kernels; do
  X=; [ $f = gru_kernels ] && X=-DOS_LAYER_TS; [ $f = kf_rows_kernel ] && X="-fno-slp-vectorize"
  [ -f $D/$f.o -a $f != gru_kernels ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-pass-failed $X -DOS_BUILD_ID='"ts-build"' -c $f.hip -o $D/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/liboptistate_lts.so $D/*.o -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
cd $R
OS_GRU_VEC=0 OPTISTATE_HIP_LIB=$D/liboptistate_lts.so python3 - <<PY
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
m = RNN(188, 128, 4, 24, torch.device("cpu"))
e = Engine(0)
e.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
x = torch.rand(64, 10, 188).cuda()
for _ in range(3):
    e.gru_forward(x)
torch.cuda.synchronize()
PY
