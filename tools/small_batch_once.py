#!/usr/bin/env python3
"""The reference's own call shapes, a fixed number of times each, for a kernel trace: 200 evaluation calls RNN(188,128,4,24) on one
window of 10 steps (gru/gru_test.py:171-177) and 200 training steps at batch 64 (gru/gru_train.py:231-249)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import RNN, train                    # noqa: E402

torch.manual_seed(0)
m = RNN(188, 128, 4, 24, torch.device("cuda"), evaluate=True).to("cuda").eval()
x1 = torch.rand(1, 10, 188, device="cuda")
with torch.no_grad():
    for _ in range(200):
        m(x1)
torch.cuda.synchronize()
mt = RNN(188, 128, 4, 24, torch.device("cuda")).to("cuda")
tr = train.DataParallelTrainer(mt, lr=1e-4)
x = torch.rand(64, 10, 188, device="cuda"); y = torch.rand(64, 12, device="cuda")
for _ in range(200):
    tr.step(x, y)
torch.cuda.synchronize()
print("done")
