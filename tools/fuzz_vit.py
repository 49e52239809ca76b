"""GPU: the ViT encoder (Transformer_Autoencoder.forward_encoder) on random frame counts -- tile tails of the GEMM / MLP / attention
kernels -- against the float64 restatement (oracle/vit_oracle.py) on a few frames of each batch.
    python tools/fuzz_vit.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from optistate_amd.transformer_model import Transformer_Autoencoder  # noqa: E402
from oracle import vit_oracle  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    torch.manual_seed(3)
    m = Transformer_Autoencoder().to("cuda")
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.1 * torch.randn_like(p))
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    bad = 0
    for case in range(n):
        N = int(rng.choice([1, 2, 3, 7, 31, 63, 64, 65, 100, 127, 128, 129, 255, 257, 513, 1000, 1024, 1027]))
        g = torch.Generator().manual_seed(100 + case)
        x = torch.rand(N, 1, 224, 224, generator=g)
        lat = m.forward_encoder(x.cuda()).cpu().numpy()[:, 0]
        pick = np.unique(np.r_[0, N // 3, N - 1])
        ref = vit_oracle.encode(x[pick, 0].numpy(), sd)
        err = float(np.abs(lat[pick] - ref).max())
        ok = np.isfinite(lat).all() and err < 2e-5
        print(f"case {case}: {N} frames: latent l-inf vs float64 on frames {pick.tolist()}: {err:.1e}" + ("" if ok else "   <-- ABOVE THE BAR"), flush=True)
        bad += 0 if ok else 1
    print(f"{n} cases, {bad} above the bars")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
