"""GPU (development): training gradient of one shape against fp64 autograd, per parameter tensor.
    python tools/debug_train_grad.py I H L C B T"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from optistate_amd import RNN  # noqa: E402


def check(I, H, L, C, B, T, verbose=False):
    torch.manual_seed(25)
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda").eval()
    x = torch.rand(B, T, I) * 2 - 1
    with torch.no_grad():
        m(x[:1].cuda())
    eng = m._engine
    xg = x.cuda()
    y = torch.rand(B, C // 2, device="cuda")
    o = eng.gru_forward_train(xg)
    _, dout, _ = eng.gru_loss(o, y, want_target=True)
    g = eng.gru_backward(xg, o, dout).double().cpu()
    md = torch.nn.GRU(I, H, L, batch_first=True).double()
    fc = torch.nn.Linear(H, C).double()
    sd = {k: v.double().cpu() for k, v in m.state_dict().items()}
    md.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gru.")})
    fc.load_state_dict({k[3:]: v for k, v in sd.items() if k.startswith("fc.")})
    hseq, _ = md(x.double())
    od = torch.sigmoid(fc(hseq[:, -1]))
    od.backward(dout.double().cpu())
    params = list(md.named_parameters()) + list(fc.named_parameters())
    ref_g = torch.cat([p.grad.reshape(-1) for _, p in params])
    scale = float(ref_g.abs().max())
    worst = float((g - ref_g).abs().max()) / scale
    line = f"RNN({I},{H},{L},{C}) B={B} T={T}: grad rel err {worst:.2e} [{eng.kernel_name('train_sweep')} / {eng.kernel_name('train_dw')} / fwd {eng.kernel_name('gru_layer')}]"
    if verbose or worst > 1e-4:
        off = 0
        for n, p in params:
            k = p.numel()
            e = float((g[off:off + k] - p.grad.reshape(-1)).abs().max()) / scale
            line += f"\n     {n}: {e:.2e}"
            off += k
    print(line, flush=True)
    return worst


if __name__ == "__main__":
    if len(sys.argv) > 6:
        check(*[int(v) for v in sys.argv[1:7]], verbose=True)
    else:
        for B in (31, 32, 33, 34, 40, 64, 65):
            for T in (3, 10, 25):
                for I in (64, 100):
                    check(I, 128, 1, 24, B, T)
