"""Development probe: where a pass of the window-stream inference mode goes (device time by torch events; per-kernel by the library's events)."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from optistate_amd import RNN
from optistate_amd import pipeline as pl
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = RNN(188, 128, 4, 24, dev).to(dev).eval()
N, T = 8192, 10
rows = torch.rand(N + T - 1, 188, device=dev)
mn, mx = torch.zeros(12, device=dev), torch.ones(12, device=dev)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    print("forward_windows only      : %.4f ms" % timeit(lambda: m.forward_windows(rows, T)))
    print("predict_rows (+ 3 denorm) : %.4f ms" % timeit(lambda: pl.predict_rows(m, rows, T, mn, mx)))
    eng = m._engine
    print("engine.gru_forward_windows: %.4f ms" % timeit(lambda: eng.gru_forward_windows(rows, T)))
    w = rows.unfold(0, T, 1).permute(0, 2, 1).contiguous()
    print("materialised forward      : %.4f ms" % timeit(lambda: m(w)))
    print("unfold+contiguous         : %.4f ms" % timeit(lambda: rows.unfold(0, T, 1).permute(0, 2, 1).contiguous()))
