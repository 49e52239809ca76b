#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_gru.py -x -q -m gpu -k "vector or window" 2>&1 | tail -3
timeout 600 bash tools/vec_ts.sh 2>&1 | grep "gru_vec_kernel" | tail -4 > $O/r04s_vec_ts.txt; cat $O/r04s_vec_ts.txt
timeout 300 python3 - <<PY
import torch, time
from optistate_amd import Engine, RNN, flatten_state_dict
m = RNN(188, 128, 4, 24, torch.device("cpu"))
e = Engine(0)
e.load_gru(flatten_state_dict(m.state_dict(), 4, "cuda"), 188, 128, 4, 24)
for (B, T) in ((1, 10), (2, 10), (4, 10), (1, 24), (1, 48)):
    x = torch.rand(B, T, 188).cuda()
    for _ in range(20): e.gru_forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): e.gru_forward(x)
    torch.cuda.synchronize(); print(B, T, "us per forward (back to back)", (time.perf_counter() - t0) / 500 * 1e6, e.kernel_name("gru_layer"))
PY
