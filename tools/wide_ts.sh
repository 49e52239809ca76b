#!/bin/bash
# In-kernel phase timestamps of gru_wide_kernel (-DOS_LAYER_TS build of gru_wide_kernel.hip): RNN(188,128,4,24), B = 64, T = 100.
# usage (GPU box, after `python -m optistate_amd.build`): bash tools/wide_ts.sh [extra -D flags]
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/wts
bash $R/tools/ts_lib.sh wts liboptistate_wts.so gru_wide_kernel -DOS_LAYER_TS "$@" > /dev/null || exit 1
cd $R
OS_GRU_VEC=0 OPTISTATE_HIP_LIB=$D/liboptistate_wts.so python3 - <<PY
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
torch.manual_seed(0)
m = RNN(188, 128, 4, 24, torch.device("cpu"))
e = Engine(0)
e.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
x = torch.rand(64, 100, 188).cuda()
for _ in range(3):
    e.gru_forward(x)
torch.cuda.synchronize()
PY
