#!/bin/bash
# In-kernel timestamps of gru_vec_kernel (-DOS_LAYER_TS build of gru_kernels.hip) on the reference's window: GRU(188,128,4), B = 1, T = 10.
# usage (GPU box): bash tools/vec_ts.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/build_ab/lts
bash $R/tools/ts_lib.sh lts liboptistate_lts.so gru_kernels -DOS_LAYER_TS > /dev/null || exit 1
cd $R
OPTISTATE_HIP_LIB=$D/liboptistate_lts.so python3 - <<PY
import torch, time
from optistate_amd import Engine, RNN, flatten_state_dict
for (B, T) in ((1, 10), (4, 10)):
    m = RNN(188, 128, 4, 24, torch.device("cpu"))
    e = Engine(0)
    e.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
    x = torch.rand(B, T, 188).cuda()
    for _ in range(3):
        e.gru_forward(x)
    torch.cuda.synchronize()
PY
