#!/usr/bin/env python3
"""Development aid (tools/layer_ts.sh builds the -DOS_LAYER_TS library): one training forward at BASELINE configs[3]'s per-GPU shape
(8192 windows x 10, RNN(188,128,4,24)) so that the instrumented layer kernels print their cycles per step and phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import RNN
from optistate_amd.train import DataParallelTrainer
torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = RNN(188, 128, 4, 24, dev).to(dev)
tr = DataParallelTrainer(m, lr=1e-4)
x = torch.rand(8192, 10, 188, device=dev); y = torch.rand(8192, 12, device=dev)
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
