#!/bin/bash
# In-kernel phase timestamps of bwd_sweep_wide_kernel (-DOS_LAYER_TS build of gru_train_kernels.hip): RNN(188,128,4,24), B = 64, T = 100.
# usage (GPU box, after `python -m optistate_amd.build`): bash tools/wide_bwd_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/wbts
bash $R/tools/ts_lib.sh wbts liboptistate_wbts.so gru_train_kernels -DOS_LAYER_TS "$@" > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_wbts.so python3 - <<PY
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
torch.manual_seed(0)
m = RNN(188, 128, 4, 24, torch.device("cpu"))
e = Engine(0)
e.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
x = torch.rand(64, 100, 188).cuda(); y = torch.rand(64, 12).cuda()
for _ in range(2):
    o = e.gru_forward_train(x)
    _, dout, _ = e.gru_loss(o, y, want_target=True)
    e.gru_backward(x, o, dout)
torch.cuda.synchronize()
PY
