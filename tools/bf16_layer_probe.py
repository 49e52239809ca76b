"""GPU: the opt-in split-bf16 layer kernel against the exact-fp32 stage kernel at the reference's model shape.
    python tools/bf16_layer_probe.py [B] [T]      (default 65536 x 100, RNN(188,128,4,24))
Prints ms per forward (whole model: 4 layer launches + head), per-layer kernel time from the library's HIP events, and the
distance between the two paths' outputs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optistate_amd import RNN  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    I = int(sys.argv[3]) if len(sys.argv) > 3 else 188
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    m = RNN(I, 128, 4, 24, dev).to(dev).eval()
    xs = torch.rand(T, I, B, device=dev)              # SoA stream [T][K][B]: the layout the Kalman kernels write
    with torch.no_grad():
        m(torch.rand(8, 2, I, device=dev))            # binds the engine and loads the weights
    eng = m._engine

    def run():
        return eng.gru_forward_soa(xs)

    def timeit(n=3):
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            out = run()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, out

    res = {}
    for mode in (0, 3, 2):
        eng.set_gru_split_bf16(mode)
        ms, out = timeit()
        eng.profile(True); run(); torch.cuda.synchronize(); pr = eng.profile_read(); eng.profile(False)
        res[mode] = out.float().cpu()
        k = pr.get("gru_layer", {})
        print(f"mode {mode}: {ms:.3f} ms per forward = {B * T / ms / 1e3:.3e} steps/s | layer kernel {eng.kernel_name('gru_layer')}: {k}")
    eng.set_gru_split_bf16(0)
    for mode in (3, 2):
        print(f"mode {mode} vs exact fp32: l-inf {float((res[mode] - res[0]).abs().max()):.3e}")


if __name__ == "__main__":
    main()
