"""GPU: DataParallelTrainer (device-side target + MSE, backward, flat bucket, fused Adam; world size 1) over several steps on random
model / batch shapes against the reference's loop (gru/gru_train.py:232-249) in float64 torch on the CPU.
    python tools/fuzz_trainer.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd import RNN  # noqa: E402
from optistate_amd.train import DataParallelTrainer  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for case in range(n):
        H = int(rng.choice([32, 64, 128, 128]))
        L = int(rng.integers(1, 5))
        I = int(rng.choice([3, 20, 60, 64, 100, 188]))
        C = 24
        B = int(rng.choice([1, 5, 32, 33, 64, 100, 300, 1000]))
        T = int(rng.choice([1, 3, 10]))
        steps, lr = 4, 1e-3
        torch.manual_seed(100 + case)
        m = RNN(I, H, L, C, torch.device("cuda")).to("cuda")
        sd0 = {k: v.detach().double().cpu().clone() for k, v in m.state_dict().items()}
        xs = [torch.rand(B, T, I) * 2 - 1 for _ in range(steps)]
        ys = [torch.rand(B, C // 2) for _ in range(steps)]
        tr = DataParallelTrainer(m, lr=lr)
        losses = [float(tr.step(x.cuda(), y.cuda()).item()) for x, y in zip(xs, ys)]
        torch.cuda.synchronize()
        # the reference's loop, float64
        gru = torch.nn.GRU(I, H, L, batch_first=True).double()
        fc = torch.nn.Linear(H, C).double()
        gru.load_state_dict({k[4:]: v for k, v in sd0.items() if k.startswith("gru.")})
        fc.load_state_dict({k[3:]: v for k, v in sd0.items() if k.startswith("fc.")})
        opt = torch.optim.Adam(list(gru.parameters()) + list(fc.parameters()), lr=lr)
        ref_losses = []
        for x, y in zip(xs, ys):
            out = torch.sigmoid(fc(gru(x.double())[0][:, -1]))
            target = torch.cat([y.double(), (out[:, :C // 2].detach() - y.double()).abs()], dim=1)
            loss = torch.nn.MSELoss()(out, target)
            opt.zero_grad(); loss.backward(); opt.step()
            ref_losses.append(loss.item())
        e_l = max(abs(a - b) / max(abs(b), 1e-12) for a, b in zip(losses, ref_losses))
        worst, frac = 0.0, 0.0
        sd1 = m.state_dict()
        refsd = {**{"gru." + k: v for k, v in gru.state_dict().items()}, **{"fc." + k: v for k, v in fc.state_dict().items()}}
        tot = out_of = 0
        for k in sd1:
            diff = (sd1[k].double().cpu() - refsd[k]).abs()
            worst = max(worst, float(diff.max()))
            out_of += int((diff > 2e-5).sum()); tot += diff.numel()
        # Adam divides by sqrt(v): an entry whose gradient is at rounding level may step the other way; everything else follows to 2e-5
        ok = e_l < 1e-4 and out_of <= 1e-3 * tot + 2
        print(f"case {case}: RNN({I},{H},{L},{C}) B={B} T={T}, {steps} Adam steps: loss rel {e_l:.1e}, weights worst {worst:.1e}, entries off by > 2e-5: {out_of}/{tot}"
              + ("" if ok else "   <-- ABOVE THE BAR"), flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
