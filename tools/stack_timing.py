"""Small-batch GRU forward: one pipelined stack launch (gru_stack_kernel) against a launch per layer, same process, same box.
usage: python3 tools/stack_timing.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import Engine, RNN, flatten_state_dict   # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
os.environ["OS_GRU_VEC"] = "0"          # B <= 4 would take gru_vec_kernel in both columns (tools/dropin_rnn_latency.py times that one)
SHAPES = [(188, 128, 4, 1, 10), (188, 128, 4, 64, 10), (188, 128, 4, 128, 8), (188, 128, 4, 128, 100), (188, 128, 4, 1024, 10),
          (188, 128, 4, 2048, 10), (60, 64, 4, 64, 100), (60, 64, 4, 1024, 100), (60, 128, 4, 512, 100)]
print("| I | H | L | B | T | per-layer launches, us | stack launch, us | ratio |")
print("|---|---|---|---|---|---|---|---|")
for (I, H, L, B, T) in SHAPES:
    torch.manual_seed(1)
    m = RNN(I, H, L, 24, torch.device("cpu"))
    x = torch.rand(B, T, I, device="cuda")
    res = {}
    for stack in ("0", "1"):
        os.environ["OS_GRU_STACK"] = stack
        e = Engine(0)
        e.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, 24)
        for _ in range(10):
            e.gru_forward(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            e.gru_forward(x)
        torch.cuda.synchronize()
        res[stack] = (time.perf_counter() - t0) / reps * 1e6
        name = e.kernel_name("gru_layer")
    print(f"| {I} | {H} | {L} | {B} | {T} | {res['0']:.1f} | {res['1']:.1f} ({name}) | {res['0'] / res['1']:.2f} |")
