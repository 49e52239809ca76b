#!/usr/bin/env python3
"""One-GPU sweep of the fused KF + GRU path over the shard sizes of a FIXED 65,536-trajectory batch.

SURVEY 8(e) partitions inference as "GPU g gets trajectories [g B/G, (g+1) B/G)": at N = 2 / 4 / 8 a rank runs 32,768 / 16,384 /
8,192 trajectories.  For every shard size this runs each tile shape of the single fused kernel (os_fused_set_tile) and the
two-kernel path, checks each against the 256-per-CU kernel's result on the same trajectories, and writes a markdown table:
kernel, ms per pass (HIP events around `--iters` passes), fraction of the fp32 MFMA peak on the GRU flops, the shape os_fused_run
picks by itself, and the projected N-GPU rate and strong-scaling efficiency of a fixed 65,536 batch (no collective on the data path).

    python tools/shard_sweep.py [--T 100] [--iters 5] [--out profiles/r06_shard_sweep.md]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optistate_amd import Engine, RNN, flatten_state_dict
from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=65536)
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--sizes", type=str, default="65536,32768,16384,8192,4096")
ap.add_argument("--out", type=str, default="")
a = ap.parse_args()

GRU_FLOP = 47616.0                   # per (trajectory, step): 2 . 3 . 64 . (60 + 64)   
PEAK = 157.3e12                      # fp32 MFMA (MI355X_MICROARCH.md)
eng = Engine(0)
eng.set_noise(Q_DEFAULT, R_DEFAULT)
d = synth_torch(a.B, a.T, "cuda", seed=2026)
d["contact_p"] = eng.contact_soa_to_packed(d["contact"])
torch.manual_seed(0)
m = RNN(60, 64, 1, 24, torch.device("cpu"))
eng.load_gru(flatten_state_dict(m.state_dict(), 1), 60, 64, 1, 24)
mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).cuda()


def shard(n):
    s = {k: d[k][:, :, :n].contiguous() for k in ("p", "f", "dp", "imu", "accel")}
    s["contact_p"] = d["contact_p"][:, :n].contiguous()
    s["x0"], s["P0"] = d["x0"][:, :n].contiguous(), d["P0"][:, :n].contiguous()
    return s


def run(s, tile, two_kernel=None):
    eng.set_fused_tile(tile)
    x, P = s["x0"].clone(), s["P0"].clone()
    r = eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, x, P, two_kernel=two_kernel)
    return r, x, P


def timeit(s, tile, two_kernel=None):
    eng.set_fused_tile(tile)
    x, P = s["x0"].clone(), s["P0"].clone()
    call = lambda: eng.fused_run(s["p"], s["f"], s["dp"], s["imu"], s["contact_p"], s["accel"], mm, x, P, two_kernel=two_kernel)
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters


rows, best = [], {}
for n in [int(v) for v in a.sizes.split(",")]:
    if n > a.B:
        continue
    s = shard(n)
    ref, xr, Pr = run(s, 256, two_kernel=False)
    torch.cuda.synchronize()
    for label, tile, tk in (("256", 256, False), ("128", 128, False), ("64", 64, False), ("32", 32, False), ("16", 16, False),
                            ("two-kernel", 0, True), ("auto", 0, None)):
        r, x, P = run(s, tile, tk)
        torch.cuda.synchronize()
        ex = (r["x_out"] - ref["x_out"]).abs().max().item()
        eo = (r["out"] - ref["out"]).abs().max().item()
        eP = (P - Pr).abs().max().item()
        bad = int((r["status"] != 0).sum())
        ms = timeit(s, tile, tk)
        name = eng.kernel_name(3) if not tk else eng.kernel_name(0) + " + " + eng.kernel_name(1)      # OS_PHASE_FUSED | OS_PHASE_KF + OS_PHASE_GRU_LAYER
        frac = n * a.T * GRU_FLOP / (ms * 1e-3) / PEAK
        rows.append(dict(B=n, shape=label, ms=ms, frac=frac, dx=ex, dout=eo, dP=eP, status_bad=bad, kernel=name))
        print(json.dumps(rows[-1]), flush=True)
        if label not in ("auto",):
            if n not in best or ms < best[n][0]:
                best[n] = (ms, label)
eng.set_fused_tile(0)

auto = {r["B"]: r for r in rows if r["shape"] == "auto"}
lines = ["# r06 shard sweep: fused KF + GRU(60 -> 64, L = 1) at the shard sizes of a fixed 65,536 batch (one MI355X)", "",
         f"T = {a.T}, {a.iters} timed passes per cell (HIP events, inputs resident), `tools/shard_sweep.py`.  `frac` = GRU flops / time / 157.3 TF.",
         "`dx` / `dout` / `dP`: l-inf distance of x_out / the head outputs / the final P to the 256-per-CU kernel's result on the same trajectories",
         "(the filter arithmetic is the same lane code in every shape: 0; the gate sums differ in their k order).", "",
         "| B | shape (trajectories per CU) | kernel | ms per pass | frac of fp32 MFMA peak | dx | dout | dP |", "|---|---|---|---|---|---|---|---|"]
for r in rows:
    lines.append(f"| {r['B']} | {r['shape']} | `{r['kernel']}` | {r['ms']:.3f} | {r['frac']:.3f} | {r['dx']:.1e} | {r['dout']:.1e} | {r['dP']:.1e} |")
lines += ["", "## what os_fused_run picks, and the projected fixed-batch curve", "",
          "| N GPUs | shard | auto ms | best forced (shape) | projected steps/s at a fixed 65,536 batch | efficiency vs N = 1 |", "|---|---|---|---|---|---|"]
base = auto.get(a.B)
for N in (1, 2, 4, 8, 16):
    n = a.B // N
    if n in auto and base:
        ms = auto[n]["ms"]
        lines.append(f"| {N} | {n} | {ms:.3f} | {best[n][0]:.3f} ({best[n][1]}) | {a.B * a.T / (ms * 1e-3):.3e} | {base['ms'] / (N * ms):.2f} |")
lines += ["", "(projection: every rank runs its shard concurrently, no data-path collective -- SURVEY 8(e); the driver's multi-GPU run measures it.)"]
text = "\n".join(lines) + "\n"
print(text)
if a.out:
    with open(a.out, "w") as fh:
        fh.write(text)
