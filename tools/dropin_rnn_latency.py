#!/usr/bin/env python3
"""Latency of the drop-in RNN at B = 1, as the reference's evaluation loop measures it (gru/gru_test.py:171-177: one window per
call, `computation_time` = the time of `outputs = model(inputs)`; here with the .cpu() read of the prediction the loop does next,
so the GPU work is inside the interval).  The reference's torch-CPU path on the build container's cores: ~1.6 ms per call.
usage: python3 tools/dropin_rnn_latency.py [calls]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optistate_amd import RNN                                  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
res = {"calls": n, "reference_torch_cpu_us_per_call": 1600.0}
for tag, env in (("vec", {}), ("stack", {"OS_GRU_VEC": "0"}), ("per_layer", {"OS_GRU_VEC": "0", "OS_GRU_STACK": "0"})):
    for k in ("OS_GRU_VEC", "OS_GRU_STACK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    from optistate_amd import engine as eng_mod
    eng_mod.reset_default_engines()            # tuning knobs are read when an Engine is created
    torch.manual_seed(0)
    m = RNN(188, 128, 4, 24, torch.device("cuda"), evaluate=True).to("cuda").eval()
    xs = torch.rand(n, 1, 10, 188)
    with torch.no_grad():
        for rep in range(2):
            t_model = 0.0
            t0 = time.perf_counter()
            for i in range(n):
                inputs = xs[i].to("cuda")
                a = time.perf_counter()
                outputs = m(inputs)
                pred = outputs[0, 0:12].cpu().numpy()
                t_model += time.perf_counter() - a
            el = time.perf_counter() - t0
    res[tag + "_us_per_call"] = t_model / n * 1e6
    res[tag + "_us_per_iteration"] = el / n * 1e6
    res[tag + "_kernel"] = m._engine.kernel_name("gru_layer")
print(json.dumps(res))
