#!/usr/bin/env python3
"""bench.py -- KF+GRU timesteps/s (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (Kalman predict/update + feature pack + GRU + head) over one batch of
synthetic input: B = 65,536 trajectories x T = 100 timesteps per GPU (BASELINE.json configs[2]).  Inputs are
resident in HBM before the timed region.  Trajectories are independent, so N GPUs run N disjoint batches with
no data-path collective (weak scaling); the only cross-rank traffic is the barrier and the max-over-ranks of
the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA dense peak
BYTES_PER_STEP_FUSED = 244   # 196 B read + 48 B KF-state write per (trajectory, timestep)
BYTES_PER_STEP_KF = 220


def gru_flops_per_step(I, H, L):
    return sum(2 * 3 * H * ((I if l == 0 else H) + H) for l in range(L))


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota (a container usually gets fewer
    than os.cpu_count(); oversubscribing the quota only adds scheduler noise)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(H, L, target_seconds):
    """The float64 C oracle (oracle/kf_oracle.c + gru_oracle.c: a port of the reference's algorithm) timed on the host
    cores over a bounded sample of the same workload (same distributions, T = 100): trajectories split over all cores
    (OpenMP), plus the single-thread figure of the scalar port."""
    import numpy as np
    import torch
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from optistate_amd import RNN
    T = 100
    torch.manual_seed(0)
    w = orc.flatten_state_dict(RNN(60, H, L, 24, torch.device("cpu")).state_dict(), L)

    def run(Bs):
        # timed: the two C entry points (filter, then GRU); the numpy glue between them (feature pack + normalise, single
        # threaded) is outside the clock so that the all-cores figure measures the port, not numpy
        d = synth_numpy(Bs, T, seed=77)
        a64 = {k: np.ascontiguousarray(d[k], dtype=np.float64) for k in ("p", "f", "dp", "imu", "accel", "x0")}
        P0 = np.tile(Q_DEFAULT, (Bs, 1, 1))
        t0 = time.perf_counter()
        r = orc.kf_run_batch(a64["p"], a64["f"], a64["dp"], a64["imu"], d["contact"], a64["x0"], P0, Q_DEFAULT, R_DEFAULT)
        t1 = time.perf_counter()
        rows = np.concatenate([r["x"], a64["accel"], a64["f"], r["p_rot"], a64["dp"], a64["imu"]], axis=2)
        rows = (rows + 30.0) / 60.0
        t2 = time.perf_counter()
        orc.gru_forward(rows, w, 60, H, L, 24)
        return (t1 - t0) + (time.perf_counter() - t2)

    probe_B = 16
    orc.set_threads(1)
    t_probe = run(probe_B)
    B1 = min(max(probe_B, int(probe_B * 0.3 * target_seconds / max(t_probe, 1e-6))), 20000)
    el1 = run(B1)
    cores = orc.set_threads(usable_cores())
    Bs = min(max(cores * probe_B, int(B1 * cores * 0.7 / 0.3)), 65536)      # at most the GPU workload's own batch
    el = run(Bs)
    orc.set_threads(1)
    # the GRU half as the reference itself computes it (gru/gru_model.py:16-24 = torch.nn.GRU + Linear + sigmoid) on the
    # host cores, fp32, for context: torch is a library, not reference code
    torch.set_num_threads(cores)
    ref_gru = torch.nn.GRU(60, H, L, batch_first=True); ref_fc = torch.nn.Linear(H, 24)
    xb = torch.rand(min(Bs, 8192), T, 60)
    with torch.no_grad():
        torch.sigmoid(ref_fc(ref_gru(xb[:64])[0][:, -1]))
        tg = time.perf_counter()
        torch.sigmoid(ref_fc(ref_gru(xb)[0][:, -1]))
        tg = time.perf_counter() - tg
    return {"value": Bs * T / el, "unit": "timesteps/s", "cores": cores, "kind": "port",
            "gru_half_torch_cpu": {"value": xb.shape[0] * T / tg, "unit": "timesteps/s", "threads": cores,
                                   "what": "torch.nn.GRU(60,%d,%d)+Linear+sigmoid fp32 on the host, GRU half only" % (H, L)},
            "sample": f"{Bs} trajectories x {T} steps (KF + GRU float64 C oracle, {cores} threads, {el:.1f} s)",
            "single_thread_value": B1 * T / el1,
            "single_thread_sample": f"{B1} trajectories x {T} steps, 1 thread, {el1:.1f} s"}


def bench_train(a, rank, local_rank, world, dist):
    """BASELINE configs[3]: data-parallel gru_train.py step, RNN(188,128,4,24), 8192 windows of 10 steps per GPU, Adam 1e-4,
    one flat 1.69 MB fp32 gradient bucket all-reduced per step.  A 'step' here is one optimisation step."""
    import torch
    from optistate_amd import RNN
    from optistate_amd.train import DataParallelTrainer
    dev = torch.device("cuda", local_rank)
    B, T, I, H, L, C = 8192, 10, 188, 128, 4, 24
    torch.manual_seed(0)
    model = RNN(I, H, L, C, dev).to(dev)
    tr = DataParallelTrainer(model, lr=1e-4)
    g = torch.Generator(device=dev); g.manual_seed(100 + rank)
    x = torch.rand(B, T, I, device=dev, generator=g); y = torch.rand(B, C // 2, device=dev, generator=g)
    for _ in range(a.warmup):
        tr.step(x, y)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ar_us = None
    if dist:
        dist.barrier()
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dist.all_reduce(tr.bucket.g)
        e1.record(); torch.cuda.synchronize()
        ar_us = e0.elapsed_time(e1) / 20 * 1e3
    if rank == 0:
        fl_fwd = gru_flops_per_step(I, H, L) * B * T
        out = {"metric": "GRU training windows/sec (gru_train.py step, data parallel)", "value": B * world * a.steps / el,
               "unit": "windows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "gru_train.py step RNN(188,128,4,24), Adam lr 1e-4, windows of 10", "batch_per_gpu": B,
                          "seq_len": T, "global_batch": B * world, "parallelism": f"dp{world}, one flat fp32 bucket all-reduce",
                          "baseline_config": "BASELINE.json configs[3]"},
               "approx_TFLOPs_fwd_bwd": 3 * fl_fwd / (el / a.steps) / 1e12, "allreduce_us": ar_us,
               "grad_bucket_bytes": int(tr.bucket.g.numel() * 4), "final_loss": float(loss.item())}
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


def bench_full(a, rank, local_rank, world, dist):
    """BASELINE configs[4]: 1024 depth frames (128 trajectories x 8 steps) -> ViT encoder latent (128-d) -> appended to the 60
    Kalman features -> GRU(188,128,4,24).  A 'step' is one pass over the 1024 frames."""
    import torch
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda", local_rank)
    B, T = 128, 8
    eng = Engine(local_rank); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    torch.manual_seed(0)
    vit = Transformer_Autoencoder().to(dev)
    model = RNN(188, 128, 4, 24, dev)
    eng.load_gru(flatten_state_dict(model.state_dict(), 4, dev), 188, 128, 4, 24)
    d = synth_torch(B, T, dev, seed=7 + rank)
    contact = eng.contact_soa_to_packed(d["contact"])
    frames = torch.rand(B * T, 1, 224, 224, device=dev)
    minmax = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).to(dev)

    def one():
        lat = vit.forward_encoder(frames).reshape(B, T, 128)
        x, P = d["x0"].clone(), d["P0"].clone()
        return eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], contact, d["accel"], minmax, x, P, latent=eng.pack(lat))
    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        dist.barrier()
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "depth frames/sec through ViT latent + KF + GRU", "value": B * T * world * a.steps / el,
                          "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "ViT encoder (3 blocks, dim 128) latent + Kalman + GRU(188,128,4,24)",
                                     "frames": B * T, "trajectories": B, "seq_len": T, "baseline_config": "BASELINE.json configs[4]",
                                     "note": "ViT parity unpinned (timm/weights absent)"}}), flush=True)


def bench_mpc(a, rank, local_rank, world, dist):
    """estimate_state_mpc over the batch -- per step the convex-MPC force QP (exact float64 active-set
    solve, one wavefront per trajectory) followed by the predict_mpc/update filter step.  A 'step' is one pass over
    B trajectories x T time steps."""
    import torch
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = torch.device("cuda", local_rank)
    B, T = a.batch, a.seq
    eng = Engine(local_rank); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(B, T, dev, seed=11 + rank)
    contact = eng.contact_soa_to_packed(d["contact"])
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    tt = torch.arange(T, device=dev)[:, None] * 0.01
    ref[:, 0] = 0.02 * torch.sin(3 * tt); ref[:, 1] = 0.02 * torch.cos(2 * tt)
    last = {}

    def one():
        x, P = d["x0"].clone(), d["P0"].clone()
        last["r"] = eng.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True)
    for _ in range(a.warmup):
        one()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = eng.profile_read()
    if dist:
        dist.barrier()
        tv = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tv, op=dist.ReduceOp.MAX)
        el = float(tv.item())
        dist.destroy_process_group()
    if rank == 0:
        it = last["r"]["iters"].float()
        print(json.dumps({"metric": "KF timesteps/sec with the convex-MPC force QP in the loop (estimate_state_mpc)",
                          "value": B * T * world * a.steps / el, "unit": "timesteps/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f64 (QP) / f32 (filter)", "data": "synthetic",
                          "config": {"workload": "estimate_state_mpc: 60-variable force QP + Kalman(12/10) predict_mpc/update",
                                     "batch_per_gpu": B, "seq_len": T, "parallelism": f"trajectory-sharded x{world}, no collective",
                                     "note": "QP parity unpinned (qpOASES absent): checked against the KKT-certified oracle"},
                          "qp_iterations_mean": float(it.mean()), "qp_iterations_max": int(it.max()),
                          "status_nonzero_trajectories": int((last["r"]["status"] != 0).sum()),
                          "kernels": {k: v for k, v in prof.items()}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="trajectories per GPU")
    ap.add_argument("--seq", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--mode", default="fused", choices=["fused", "kf", "train", "full", "mpc"],
                    help="kf = BASELINE configs[1]-style KF-only run; train = configs[3] data-parallel gru_train step")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT

    if a.mode == "train":
        return bench_train(a, rank, local_rank, world, dist)
    if a.mode == "full":
        return bench_full(a, rank, local_rank, world, dist)
    if a.mode == "mpc":
        return bench_mpc(a, rank, local_rank, world, dist)
    B, T, H, L, I = a.batch, a.seq, a.hidden, a.layers, 60
    eng = Engine(local_rank)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    dev = eng.device
    d = synth_torch(B, T, dev, seed=1000 + rank)
    contact = eng.contact_soa_to_packed(d["contact"])
    torch.manual_seed(0)
    model = RNN(I, H, L, 24, dev)                       # random-init weights of the named architecture
    eng.load_gru(flatten_state_dict(model.state_dict(), L, dev), I, H, L, 24)
    # min-max constants for the synthetic distributions (every feature lands in (0,1) like the reference's scaling)
    minmax = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).to(dev)
    x0, P0 = d["x0"], d["P0"]
    x, P = x0.clone(), P0.clone()

    def one_step():
        x.copy_(x0); P.copy_(P0)                        # device-to-device reset of the 40 MB filter state
        if a.mode == "fused":
            return eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], contact, d["accel"], minmax, x, P)
        return eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], contact, x, P)

    for _ in range(a.warmup):
        one_step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    eng.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = one_step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        dist.barrier()
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    prof = eng.profile_read()
    eng.profile(False)
    bad = int((r["status"] != 0).sum().item())

    if rank == 0:
        steps_per_pass = B * T
        total = steps_per_pass * world * a.steps
        # dominant kernel: the one with the largest summed device time in the timed region
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        dom_name, (dom_ms, dom_n) = dom
        avg_ms = dom_ms / max(dom_n, 1)
        if dom_name in ("gru_layer", "fused"):
            # MFMA-bound kernels: algorithmic flops = the GRU cell's matrix flops the launch performs (the Kalman
            # arithmetic of the fused kernel runs on the VALU and is not counted)
            launches_per_pass = L if dom_name == "gru_layer" else 1
            layer_flops = gru_flops_per_step(I, H, L) if dom_name == "gru_layer" else gru_flops_per_step(I, H, 1)
            fl = layer_flops * steps_per_pass / launches_per_pass
            ach = fl / (avg_ms * 1e-3) / 1e12
            kname = "gru_layer_kernel" if dom_name == "gru_layer" else "fused_kf_gru_kernel"
            roof = {"kernel": kname + " (v_mfma_f32_32x32x2_f32)", "bound": "mfma", "achieved": ach,
                    "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": None,
                    "avg_launch_ms": avg_ms}
            if dom_name == "fused":
                gbs = BYTES_PER_STEP_FUSED * steps_per_pass / (avg_ms * 1e-3) / 1e9
                roof["hbm_algorithmic_GBps"] = gbs
                roof["hbm_frac"] = gbs / HBM_PEAK_GBS
        else:
            bps = BYTES_PER_STEP_KF
            ach = bps * steps_per_pass / (avg_ms * 1e-3) / 1e9
            roof = {"kernel": "kf_run_sym_kernel" if dom_name == "kf" else dom_name, "bound": "hbm", "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "avg_launch_ms": avg_ms}
        # HBM bytes per launch from the PMC passes (tools/traffic_pass.sh -> profiles/traffic.json; FETCH_SIZE corrected
        # x2 per the gfx950 calibration); valid for the default bench shape only
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tj) and B == 65536 and T == 100:
            try:
                roof["traffic"] = json.load(open(tj)).get(roof["kernel"].split(" ")[0])
                roof["traffic_unit"] = "bytes/launch"
                roof["algorithmic_bytes"] = (BYTES_PER_STEP_FUSED if dom_name == "fused" else BYTES_PER_STEP_KF) * steps_per_pass
            except Exception:
                pass
        kernels = {k: {"ms_per_launch": v[0] / max(v[1], 1), "launches": v[1]} for k, v in prof.items() if v[1]}
        if "kf" in kernels:
            kf_ms = kernels["kf"]["ms_per_launch"]
            kernels["kf"]["algorithmic_GBps"] = BYTES_PER_STEP_KF * steps_per_pass / (kf_ms * 1e-3) / 1e9
            kernels["kf"]["hbm_frac"] = kernels["kf"]["algorithmic_GBps"] / HBM_PEAK_GBS
        out = {
            "metric": "KF+GRU timesteps/sec" if a.mode == "fused" else "KF timesteps/sec",
            "value": total / el, "unit": "timesteps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"fused Kalman(12-state/10-meas)+GRU(in={I},hidden={H},layers={L},out=24) inference"
                                   if a.mode == "fused" else "Kalman(12-state/10-meas) predict/update only",
                       "batch_per_gpu": B, "seq_len": T, "global_batch": B * world,
                       "parallelism": f"trajectory-sharded x{world}, no collective",
                       "baseline_config": "BASELINE.json configs[2]" if a.mode == "fused" else "configs[1]-like"},
            "roofline": roof,
            "kernels": kernels,
            "status_nonzero_trajectories": bad,
        }
        if a.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(H, L, a.cpu_seconds)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
