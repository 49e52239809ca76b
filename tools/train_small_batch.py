#!/usr/bin/env python3
"""The reference's OWN training step (gru/gru_train.py:32-37, :231-249: RNN(188,128,4,24), batch_size = 64, windows of 10, Adam 1e-4)
on the drop-in trainer: milliseconds per step with forward and backward sweep on four CUs per (layer, tile) (gru_wide_kernel /
bwd_sweep_wide_kernel, the default) and on one (OS_GRU_WIDE=0: gru_stack_kernel / bwd_sweep_stack_kernel).  The reference's torch-CPU step on the build container's eight cores: 18.9 ms.
usage: python3 tools/train_small_batch.py [steps]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = {"steps": steps, "reference_torch_cpu_ms_per_step": 18.9}
for tag, env in (("wide", {}), ("stack", {"OS_GRU_WIDE": "0"})):
    os.environ.pop("OS_GRU_WIDE", None)
    os.environ.update(env)
    from optistate_amd import engine as eng_mod, train
    eng_mod.reset_default_engines()
    for B in (64, 512):
        torch.manual_seed(0)
        from optistate_amd import RNN
        m = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda")
        tr = train.DataParallelTrainer(m, lr=1e-4)
        x = torch.rand(B, 10, 188, device="cuda"); y = torch.rand(B, 12, device="cuda")
        for _ in range(10):
            tr.step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            tr.step(x, y)
        torch.cuda.synchronize()
        res[f"{tag}_B{B}_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
print(json.dumps(res))
