#!/usr/bin/env python3
"""bench.py -- KF+GRU timesteps/s (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no torchrun environment this process starts the N ranks itself (a fresh
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` child, BEFORE anything here
touches the GPU) and exits with the child's code; launched under torch.distributed.run it is one rank.

A "step" is one pass of the hot path (Kalman predict/update + feature pack + GRU + head) over one batch of
synthetic input: B = 65,536 trajectories x T = 100 timesteps per GPU (BASELINE.json configs[2]).  Inputs are
resident in HBM before the timed region.  Trajectories are independent, so N GPUs run N disjoint batches with
no data-path collective (weak scaling, the default); the only cross-rank traffic is the barrier and the max-over-ranks of
the elapsed time.  Rank 0 prints ONE JSON line.

--scaling strong (modes fused / kf): --batch is the GLOBAL batch (default 65,536 = BASELINE.json's "at batch 65536"), every rank
generates the same synthetic batch and runs its contiguous shard [r B/N, (r+1) B/N) of it (optistate_amd.train.shard_range),
`value` = B x T x K / time, `"scaling": "strong"`.  os_fused_run picks the kernel tile shape for the shard size (256 ... 16
trajectories per CU), so a shard of 8,192 still fills the chip.

Other lines (same contract): --mode kf (configs[1]: --batch 4096 --seq 1000), --mode train (configs[3]), --mode full
(configs[4]), --mode mpc (estimate_state_mpc), --mode windows (the reference's own inference mode,
gru/gru_test.py:138-191: every output re-runs a 10-step window from h0 = 0), --split-bf16 (opt-in reduced-precision gate
GEMM of the fused kernel, reported beside -- never instead of -- the exact-fp32 default).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA dense peak (= fp32 vector peak)
MFMA_BF16_PEAK_TF = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
FP64_VECTOR_PEAK_TF = 78.6   # AMD MI355X spec sheet: fp64 vector = half of MI355X_MICROARCH.md's 157.3 TF fp32 vector figure (VERDICT r4 names it)
BYTES_PER_STEP_FUSED = 244   # 196 B read + 48 B KF-state write per (trajectory, timestep)
BYTES_PER_STEP_KF = 220
KF_FLOPS_STRUCTURED = 7500   # ~7-8 kflop per step exploiting the H selection and the two-block F_d
# the reference itself (NumPy, one thread) measured in the BUILD container, 327 us/step; it cannot
# travel to the GPU box, so it rides along as a constant beside the C port timed live
REFERENCE_PYTHON_STEPS_PER_S = 3.05e3


NOISE_NOTE = {"default": "settings.py:28-31 defaults (Q diag 1e-2 / 1e-4, R = 1e-2 I), P0 = Q",
              "fitted": "Q_R.pkl fitted values with R[0:3] = 1e-4 (data_conversion_Kalman_to_Training.py:139-144), P0 = Q"}
INPUT_NOTE = {"nominal": "SURVEY 8(d) distributions: trot contacts (2 stance legs), attitude within +-0.15 rad",
              "hostile": "0-4 stance legs per step in random segments (standing and flight included), yaw unwrapping past +-pi, "
                         "roll / pitch to +-1 rad, exact-zero and k*pi/2 attitude starts (synth.py: hostile=True)"}


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


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU launch
# ------------------------------------------------------------------------------------------------------------------
def launch_command(n, argv, port):
    """The command `python bench.py --gpus N ...` turns into when it is not already a rank (contract in the task
    statement: one rank per GPU over RCCL, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n, argv):
    """Starts the N rank processes as a CHILD (never exec: nothing in this process has touched the GPU, and it stays that
    way) and relays the child's exit code; rank 0's JSON line goes straight to our stdout."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    env.update(rccl_env_from_argv(argv))                    # --rccl-proto / --rccl-algo: in place before any rank creates a communicator
    return subprocess.run(launch_command(n, argv, _free_port()), env=env).returncode


def rccl_env_from_argv(argv):
    """NCCL_PROTO / NCCL_ALGO for the ranks (RCCL reads them at communicator creation).  Parsed by hand: this runs in the
    launcher, before argparse and before anything imports torch."""
    env = {}
    argv = list(argv)
    for flag, name, allowed in (("--rccl-proto", "NCCL_PROTO", ("default", "LL", "LL128", "Simple")),
                                ("--rccl-algo", "NCCL_ALGO", ("default", "Ring", "Tree"))):
        val = None
        for i, t in enumerate(argv):
            if t == flag and i + 1 < len(argv):
                val = argv[i + 1]
            elif t.startswith(flag + "="):
                val = t.split("=", 1)[1]
        if val is not None and val != "default":
            if val not in allowed:
                raise SystemExit(f"{flag}: one of {allowed}")
            env[name] = val
    return env


class Ranks:
    """Process-group plumbing shared by every mode: rank ids, barrier + max-over-ranks timing, rank/device report."""

    def __init__(self, a):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={self.world}: launch `python bench.py --gpus N` and let it start its ranks")
        self.dist = None
        self.cpu_only = a.launch_check and torch.cuda.device_count() == 0
        # --share-gpu (development only): rank r uses device r % device_count and the process group runs over gloo with the
        # collectives staged through the host -- RCCL refuses two ranks on one device.  It exists so that the world > 1 code
        # paths run with real device tensors on a one-GPU box; its timings say nothing about xGMI and the line says so.
        self.share = bool(getattr(a, "share_gpu", False)) and not self.cpu_only
        if not self.cpu_only:
            self.device_index = self.local_rank % torch.cuda.device_count() if self.share else self.local_rank
            torch.cuda.set_device(self.device_index)
            self.dev = torch.device("cuda", self.device_index)
        else:
            self.device_index = 0
            self.dev = torch.device("cpu")
        if self.world > 1 or getattr(a, "force_dist", False):
            import torch.distributed as dist
            self.dist = dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_free_port()))
            # launched directly under torch.distributed.run (the driver's form): the knob still lands before the communicator exists
            os.environ.update(rccl_env_from_argv(["--rccl-proto", getattr(a, "rccl_proto", "default"), "--rccl-algo", getattr(a, "rccl_algo", "default")]))
            if self.cpu_only or self.share:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)   # nccl == RCCL on ROCm

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if not self.dist:
            return seconds
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if self.share else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def report(self):
        """What the process group itself says about the job: world size as RCCL sees it and every rank's device."""
        import torch
        me = {"rank": self.rank, "local_rank": self.local_rank, "pid": os.getpid(),
              "rccl_env": {k: os.environ[k] for k in ("NCCL_PROTO", "NCCL_ALGO") if k in os.environ}}
        if not self.cpu_only:
            pr = torch.cuda.get_device_properties(self.dev)
            me.update(device=torch.cuda.current_device(), name=pr.name, gcn_arch=getattr(pr, "gcnArchName", ""),
                      pci_bus_id=getattr(pr, "pci_bus_id", None), hbm_gib=round(pr.total_memory / 2 ** 30, 1))
        if not self.dist:
            return {"rccl_world_size": 1, "backend": None, "rank_devices": [me]}
        allr = [None] * self.world
        self.dist.all_gather_object(allr, me)
        out = {"rccl_world_size": self.dist.get_world_size(), "backend": self.dist.get_backend(), "rank_devices": allr}
        if self.share:
            out["share_gpu"] = True
            out["share_gpu_note"] = ("development run: every rank on the same device, process group over gloo with host-staged "
                                     "collectives; exercises the world > 1 code paths, its timings are not a scaling point")
        return out

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def _kernel_table(eng, prof, steps):
    return {k: {"ms_per_launch": v[0] / v[1], "launches_per_step": v[1] / steps, "kernel": eng.kernel_name(k)}
            for k, v in prof.items() if v[1]}


def timed_region(rk, warmup, steps, one_step, eng=None):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides, max over ranks.
    eng: record the library's per-kernel HIP events (on the launch stream, around each internal kernel) over the timed
    region itself; returns (seconds, last result, kernel table | None)."""
    import torch
    r = None
    for _ in range(warmup):
        r = one_step()
    torch.cuda.synchronize()
    rk.barrier()
    if eng is not None:
        eng.profile(True)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = one_step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rk.barrier()
    kernels = None
    if eng is not None:
        kernels = _kernel_table(eng, eng.profile_read(), steps)
        eng.profile(False)
    return rk.max_over_ranks(el), r, kernels


def events_pass(eng, steps, one_step):
    """A second, untimed pass with the library's per-kernel HIP events on (recorded on the launch stream around each
    internal kernel): the wall time above stays free of the event overhead."""
    import torch
    eng.profile(True)
    for _ in range(steps):
        one_step()
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    return _kernel_table(eng, prof, steps)


_JSON_FD = None


def protect_stdout():
    """RCCL writes a version banner to the process's STDOUT (file descriptor 1) when a communicator is created: with ranks
    involved, everything that goes to fd 1 is sent to stderr and the ONE JSON line is written to the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    if _JSON_FD is None:
        print(line, flush=True)
    else:
        sys.stdout.flush()
        os.write(_JSON_FD, (line + "\n").encode())


def base_line(a, rk, metric, unit, value, el, dtype, config):
    out = {"metric": metric, "value": value, "unit": unit, "n_gpus": rk.world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": getattr(a, "scaling", "weak"), "vs_baseline": None,
           "dtype": dtype, "data": "synthetic", "config": config}
    return out


TRAFFIC_SOURCE_OF = {"fused_kf_gru_kernel_v2": "fused_kernels.hip", "fused_kf_gru_kernel_v3": "fused_kernels.hip", "fused_kf_gru_bf16_kernel": "fused_kernels.hip",
                     "fused_kf_gru_kernel": "fused_kernels.hip", "kf_run_sym_kernel": "kf_kernels.hip",
                     "kf_run_rows2_kernel": "kf_rows_kernel.hip"}


def load_traffic(kernel_name, shape_ok):
    """HBM bytes per launch from the PMC passes (tools/traffic_pass.sh -> profiles/traffic.json; FETCH_SIZE corrected x2
    per the gfx950 calibration), WITH PROVENANCE: every entry carries the content key of the object file its kernel was
    compiled into when the counters were collected (optistate_amd/build.py object_key: source + the headers it includes +
    flags + toolchain) and the collection date.  The figure is returned only when that key equals the key of the sources
    the loaded library was built from and the shape is the one it was collected on; anything else -- a changed kernel under
    the same name, another shape, a renamed kernel -- reports null and says why.  Returns (bytes or None, source string)."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    name = kernel_name.split("<")[0].split(" ")[0]
    if not shape_ok:
        return None, "not collected for this shape / mode"
    if not os.path.exists(tj):
        return None, "profiles/traffic.json absent"
    try:
        tab = json.load(open(tj))
        if name not in tab:
            return None, f"profiles/traffic.json has no entry for {name}"
        det = tab.get(name + "_detail", {})
        from optistate_amd import build as _b
        src = TRAFFIC_SOURCE_OF.get(name)
        now = _b.object_key(src, _b.toolchain_id()) if src else None
        if not det.get("source_key") or det.get("source_key") != now:
            return None, (f"profiles/traffic.json entry for {name} is stale: collected for object key {det.get('source_key')}, "
                          f"the library's {src} has key {now}")
        return tab[name], f"profiles/traffic.json@{det['source_key']} ({src}, collected {det.get('collected', '?')})"
    except Exception as e:                               # a reporting aid must never take the bench line down
        return None, f"profiles/traffic.json unreadable: {e!r}"


def rocprof_avg_ms(kernel_name, tag="default"):
    """Average launch duration of `kernel_name` in the newest TRACKED rocprofv3 --kernel-trace --stats summary of this mode
    (profiles/rNN_<tag>_kernel_stats.md, written on the GPU box by tools/collect_profiles.sh from the same command): the figure the live
    HIP-event measurement is calibrated against.  Returns (ms or None, source)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]*_{tag}_kernel_stats.md")))
    if not files:
        return None, f"no profiles/rNN_{tag}_kernel_stats.md"
    name = kernel_name.split("<")[0].split(" ")[0]
    f = files[-1]
    try:
        for line in open(f):
            cells = [c.strip() for c in line.split("|")]
            if len(cells) > 5 and name in cells[1] and re.fullmatch(r"[0-9.]+", cells[4] or "x"):
                return float(cells[4]) / 1e3, f"{os.path.relpath(f, ROOT)} (rocprofv3 average over {cells[2]} calls, profiler on, warm-ups included)"
    except Exception as e:
        return None, f"{os.path.relpath(f, ROOT)} unreadable: {e!r}"
    return None, f"{os.path.relpath(f, ROOT)} has no row for {name}"


# ------------------------------------------------------------------------------------------------------------------
# CPU baselines and the parity block: the ONLY users of oracle/ in this file (checker / reported baseline, never the
# thing measured as `value`)
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(H, L, target_seconds, kf_only=False, all_cores=None):
    """The float64 C oracle (oracle/kf_oracle.c + gru_oracle.c: a port of the reference's algorithm) timed on the host
    cores over a bounded sample of the same workload (same distributions, T = 100): trajectories split over all cores
    (OpenMP), plus the single-thread figure of the scalar port.
    all_cores: the all-cores figure when oracle_pass() has already run the port over the timed batch itself (one run
    serves the parity block and this baseline); only the single-thread sample is timed here then."""
    import numpy as np
    import torch
    from oracle import c_oracle as orc
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from optistate_amd import RNN
    T = 100
    torch.manual_seed(0)
    w = None if kf_only else orc.flatten_state_dict(RNN(60, H, L, 24, torch.device("cpu")).state_dict(), L)

    def run(Bs):
        # timed: the two C entry points (filter, then GRU); the numpy glue between them (feature pack + normalise, single
        # threaded) is outside the clock so that the all-cores figure measures the port, not numpy
        d = synth_numpy(Bs, T, seed=77)
        a64 = {k: np.ascontiguousarray(d[k], dtype=np.float64) for k in ("p", "f", "dp", "imu", "accel", "x0")}
        P0 = np.tile(Q_DEFAULT, (Bs, 1, 1))
        t0 = time.perf_counter()
        r = orc.kf_run_batch(a64["p"], a64["f"], a64["dp"], a64["imu"], d["contact"], a64["x0"], P0, Q_DEFAULT, R_DEFAULT)
        t1 = time.perf_counter()
        if kf_only:
            return t1 - t0
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
    what = "KF float64 C oracle" if kf_only else "KF + GRU float64 C oracle"
    if all_cores is None:
        cores = orc.set_threads(usable_cores())
        Bs = min(max(cores * probe_B, int(B1 * cores * 0.7 / 0.3)), 65536)      # at most the GPU workload's own batch
        el = run(Bs)
        orc.set_threads(1)
        Tb = T
        sample = f"{Bs} trajectories x {T} steps ({what}, {cores} threads, {el:.1f} s)"
    else:
        Bs, Tb, el, cores = all_cores["trajectories"], all_cores["timesteps_each"], all_cores["seconds"], all_cores["cores"]
        sample = (f"{Bs} trajectories x {Tb} steps of the TIMED batch itself ({what}, {cores} threads, {el:.1f} s in the C entry "
                  "points; the same run is the parity block's reference)")
    out = {"value": Bs * Tb / el, "unit": "timesteps/s", "cores": cores, "kind": "port",
           "sample": sample,
           "single_thread_value": B1 * T / el1,
           "single_thread_sample": f"{B1} trajectories x {T} steps, 1 thread, {el1:.1f} s",
           "reference_python_steps_per_s": REFERENCE_PYTHON_STEPS_PER_S,
           "reference_python_note": "the reference's own NumPy path (get_odom+set_measurements+predict+update), one thread, "
                                    "measured in the build container (SURVEY.md section 6); it cannot travel to the GPU box"}
    if not kf_only:
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
        out["gru_half_torch_cpu"] = {"value": xb.shape[0] * T / tg, "unit": "timesteps/s", "threads": cores,
                                     "what": "torch.nn.GRU(60,%d,%d)+Linear+sigmoid fp32 on the host, GRU half only" % (H, L)}
    return out


def cpu_baseline_train(target_seconds, batch=8192):
    """gru/gru_train.py:232-249 as the reference runs it (torch.nn.GRU + Linear + sigmoid, the self-referential target,
    MSELoss, Adam lr 1e-4) on the host cores, fp32, on a bounded number of windows of the same shape (at most the bench's own
    batch: --batch 64 is the reference's, gru/gru_train.py:36)."""
    import torch
    cores = usable_cores()
    torch.set_num_threads(cores)
    I, H, L, C, T = 188, 128, 4, 24, 10
    torch.manual_seed(0)
    gru = torch.nn.GRU(I, H, L, batch_first=True); fc = torch.nn.Linear(H, C)
    opt = torch.optim.Adam(list(gru.parameters()) + list(fc.parameters()), lr=1e-4)

    def step(x, y):
        out = torch.sigmoid(fc(gru(x)[0][:, -1]))
        tgt = torch.cat([y, (out[:, :12].detach() - y).abs()], dim=1)
        loss = torch.nn.functional.mse_loss(out, tgt)
        opt.zero_grad(); loss.backward(); opt.step()

    Bs = min(256, batch)
    x, y = torch.rand(Bs, T, I), torch.rand(Bs, 12)
    step(x, y)
    t0 = time.perf_counter(); step(x, y); t1 = time.perf_counter() - t0
    Bs = int(min(batch, max(Bs, Bs * 0.5 * target_seconds / max(t1, 1e-6))))
    x, y = torch.rand(Bs, T, I), torch.rand(Bs, 12)
    t0 = time.perf_counter(); step(x, y); el = time.perf_counter() - t0
    return {"value": Bs / el, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"one optimisation step on {Bs} windows x {T} steps, RNN(188,128,4,24), torch CPU fp32, {cores} threads, {el:.1f} s",
            "what": "torch.nn.GRU/Linear/MSELoss/Adam exactly as gru_train.py:217-249 composes them (torch is a library)"}


def cpu_baseline_full(target_seconds):
    """ViT-encoder latent (oracle/vit_oracle.py, float64 numpy) + KF + GRU(188,128,4) C oracle on a bounded number of frames."""
    import numpy as np
    import torch
    from oracle import c_oracle as orc, vit_oracle
    from optistate_amd.synth import synth_numpy, Q_DEFAULT, R_DEFAULT
    from optistate_amd import RNN
    from optistate_amd.transformer_model import Transformer_Autoencoder
    cores = orc.set_threads(usable_cores())
    torch.manual_seed(0)
    sd = {k: v.detach().numpy() for k, v in Transformer_Autoencoder().state_dict().items()}
    w = orc.flatten_state_dict(RNN(188, 128, 4, 24, torch.device("cpu")).state_dict(), 4)
    Bt, T = 2, 8
    frames = np.random.default_rng(0).random((Bt * T, 224, 224))
    d = synth_numpy(Bt, T, seed=5)
    t0 = time.perf_counter()
    lat = vit_oracle.encode(frames, sd).reshape(Bt, T, 128)
    r = orc.kf_run_batch(d["p"], d["f"], d["dp"], d["imu"], d["contact"], d["x0"], np.tile(Q_DEFAULT, (Bt, 1, 1)), Q_DEFAULT, R_DEFAULT)
    rows = np.concatenate([r["x"], d["accel"], d["f"], r["p_rot"], d["dp"], d["imu"]], axis=2)
    orc.gru_forward(np.concatenate([(rows + 30.0) / 60.0, lat], axis=2), w, 188, 128, 4, 24)
    el = time.perf_counter() - t0
    orc.set_threads(1)
    return {"value": Bt * T / el, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{Bt * T} frames (numpy float64 ViT restatement on BLAS threads + C oracle KF/GRU, {el:.1f} s); the ViT blocks are pinned to transformers' ViTLayer under the reference's glue (G14), not to timm 0.3.2 itself"}


def cpu_baseline_mpc(target_seconds):
    """estimate_state_mpc on the host: oracle/mpc_oracle.py (numpy active-set QP with KKT certificate) + C oracle filter step."""
    import numpy as np
    from oracle import mpc_oracle as mo
    from optistate_amd.synth import synth_numpy
    d = synth_numpy(4, 8, seed=3)
    ref = np.array([0, 0, 0, 0, 0, 0.28, 0, 0, 0, 0.1, 0, 0.])
    n = 0
    t0 = time.perf_counter()
    for b in range(4):
        for t in range(8):
            mo.mpc_forces(d["x0"][b].astype(np.float64), ref, d["p"][b, t].astype(np.float64), d["contact"][b, t])
            n += 1
            if time.perf_counter() - t0 > target_seconds:
                break
    el = time.perf_counter() - t0
    return {"value": n / el, "unit": "timesteps/s", "cores": 1, "kind": "port",
            "sample": f"{n} force QPs (numpy float64 active-set restatement, 1 thread, {el:.1f} s); the filter step adds <1 %; "
                      "the QP is the reference's own assembly (G12: H, g and every constraint row extracted from force_controller.py); "
                      "qpOASES itself is absent, the minimiser of the strictly convex QP is KKT-certified"}


def oracle_pass(d, x_out, out, model, H, L, Q, R, P0, want, cap_seconds, kf_only=False, latent=None):
    """The float64 C oracle over the TIMED batch itself (rank 0's shard), outside the timed region, split over all usable
    host cores: state_linf / gru_linf over EVERY (trajectory, timestep) it covers (BASELINE.json's metric is 'timesteps/s
    ...; state l-inf vs CPU ref', SURVEY 8(d): over all B x T), and the seconds the C entry points took -- the same run is
    the all-cores cpu_baseline.  d: the device input streams [T][F][B]; x_out [T][12][B], out [B][C] from the last timed
    pass; want: trajectories to cover (B = all).  Chunks of 8192 trajectories in a seeded random order, so a run that hits
    cap_seconds (a box with few cores) still reports an unbiased sample and says how many trajectories it covered."""
    import numpy as np
    import torch
    from oracle import c_oracle as orc
    B, T = int(d["p"].shape[2]), int(d["p"].shape[0])
    cores = orc.set_threads(usable_cores())
    order = torch.randperm(B, generator=torch.Generator().manual_seed(3))[:max(1, min(want, B))]
    w = None if kf_only else orc.flatten_state_dict(model.state_dict(), L)
    CH = 8192
    done, secs, s_linf, g_linf, bad, nonfinite = 0, 0.0, 0.0, 0.0, 0, 0
    worst = None
    for c0 in range(0, int(order.numel()), CH):
        idx = order[c0:c0 + CH].sort().values.to(d["p"].device)
        n = int(idx.numel())
        g = lambda k: d[k][:, :, idx].permute(2, 0, 1).double().cpu().numpy()
        a = {k: g(k) for k in (("p", "f", "dp", "imu") if kf_only else ("p", "f", "dp", "imu", "accel"))}
        contact = d["contact"][:, :, idx].permute(2, 0, 1).contiguous().cpu().numpy()
        x0 = d["x0"][:, idx].t().double().cpu().numpy()
        P0n = np.tile(np.asarray(P0, dtype=np.float64), (n, 1, 1))
        t0 = time.perf_counter()
        ref = orc.kf_run_batch(a["p"], a["f"], a["dp"], a["imu"], contact, x0, P0n, Q, R, aux=not kf_only)
        secs += time.perf_counter() - t0
        bad += int((ref["status"] != 0).sum())
        err = np.abs(x_out[:, :, idx].permute(2, 0, 1).cpu().numpy() - ref["x"])
        nf = ~np.isfinite(err)                       # a NaN anywhere must fail the gate: Python's max(a, nan) is a
        nonfinite += int(nf.sum())
        err[nf] = np.inf
        if err.max() > s_linf or worst is None:
            bi, ti, ci = np.unravel_index(int(err.argmax()), err.shape)
            worst = {"trajectory": int(idx[bi].item()), "timestep": int(ti), "component": int(ci)}
        s_linf = max(s_linf, float(err.max()))
        if not kf_only:
            rows = np.concatenate([ref["x"], a["accel"], a["f"], ref["p_rot"], a["dp"], a["imu"]], axis=2)
            rows = (rows + 30.0) / 60.0
            if latent is not None:          # the latent stream is appended un-normalised (it is in (0, 1) already: gru_test.py:135-136)
                rows = np.concatenate([rows, latent[:, :, idx].permute(2, 0, 1).double().cpu().numpy()], axis=2)
            t0 = time.perf_counter()
            ro, _, _ = orc.gru_forward(rows, w, rows.shape[2], H, L, 24)
            secs += time.perf_counter() - t0
            gerr = np.abs(out[idx].cpu().numpy() - ro)
            nonfinite += int((~np.isfinite(gerr)).sum())
            gerr[~np.isfinite(gerr)] = np.inf
            g_linf = max(g_linf, float(gerr.max()))
        done += n
        del a, ref, err
        if secs > cap_seconds:
            break
    orc.set_threads(1)
    res = {"trajectories": done, "timesteps_each": T, "of_batch": B, "whole_tensor": bool(done == B),
           "state_linf": s_linf, "state_linf_at": worst, "state_bar": 1e-4, "oracle_status_nonzero": bad,
           "nonfinite": nonfinite,
           "reference": "oracle/kf_oracle.c + gru_oracle.c (float64; pinned to reference-generated goldens)"}
    if done < min(want, B):
        res["note"] = f"stopped after {secs:.0f} s of oracle time on {cores} cores (cap {cap_seconds:.0f} s): seeded random sample"
    if not kf_only:
        res["gru_linf"] = g_linf
        res["gru_bar"] = 1e-5
        res["fused_chain_bar"] = 1e-4
    res["ok"] = bool(nonfinite == 0 and s_linf < 1e-4 and g_linf < 1e-4)
    for k in ("state_linf", "gru_linf"):            # strict JSON has no Infinity: null + the non-finite count says it
        if k in res and not np.isfinite(res[k]):
            res[k] = None
    return res, {"trajectories": done, "timesteps_each": T, "seconds": secs, "cores": cores}


# ------------------------------------------------------------------------------------------------------------------
# modes
# ------------------------------------------------------------------------------------------------------------------
def bench_train(a, rk):
    """BASELINE configs[3]: data-parallel gru_train.py step, RNN(188,128,4,24), 8192 windows of 10 steps per GPU, Adam 1e-4,
    one flat 1.69 MB fp32 gradient bucket all-reduced per step.  A 'step' here is one optimisation step."""
    import torch
    from optistate_amd import RNN
    from optistate_amd.train import DataParallelTrainer
    dev = rk.dev
    B, T, I, H, L, C = a.batch, 10, 188, 128, 4, 24                # default 8192 (main); --batch 64: the reference's own batch size (gru/gru_train.py:36)
    torch.manual_seed(0)
    model = RNN(I, H, L, C, dev).to(dev)
    tr = DataParallelTrainer(model, lr=1e-4, split_allreduce=not a.no_split_allreduce, force_distributed=a.force_dist,
                             proto=a.rccl_proto if rk.dist else None, algo=a.rccl_algo if rk.dist else None)
    if a.split_bf16:
        # opt-in (second line, never the default): the weight-gradient products on the bf16 matrix instruction with split operands
        tr.eng.set_gru_split_bf16(a.split_bf16, train=True)
    g = torch.Generator(device=dev); g.manual_seed(100 + rk.rank)
    x = torch.rand(B, T, I, device=dev, generator=g); y = torch.rand(B, C // 2, device=dev, generator=g)
    el, loss, _ = timed_region(rk, a.warmup, a.steps, lambda: tr.step(x, y))
    kernels = events_pass(tr.eng, max(2, min(a.steps, 5)), lambda: tr.step(x, y))
    ar_us = None
    if rk.dist:
        from optistate_amd.train import all_reduce_
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        all_reduce_(tr.bucket.g)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            all_reduce_(tr.bucket.g)
        e1.record(); torch.cuda.synchronize()
        ar_us = e0.elapsed_time(e1) / 20 * 1e3
    # replica equality (every rank calls it: one broadcast + one MAX all-reduce): identical replicas + one averaged gradient +
    # the same fused Adam must leave bit-identical weights on every rank after the timed steps
    divergence = tr.replica_divergence()
    info = rk.report()
    if rk.rank == 0:
        rows = B * T
        # algorithmic flops per optimisation step and phase (fp32 MFMA GEMMs only): forward gates, backward-sweep dx/dh
        # products, weight-gradient reductions
        fl_fwd = gru_flops_per_step(I, H, L) * rows
        fl_sweep = sum(2 * 3 * H * ((H if l > 0 else 0) + H) for l in range(L)) * rows          # dx only where a layer below exists
        fl_dw = sum(2 * 3 * H * (I if l == 0 else H) * rows + 2 * 3 * H * H * (rows - B) for l in range(L))
        phase_flops = {"gru_layer": fl_fwd, "train_sweep": fl_sweep, "train_dw": fl_dw}
        dom = max((k for k in kernels if k in phase_flops), key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
        dk = kernels[dom]
        fl_launch = phase_flops[dom] / dk["launches_per_step"]
        ach = fl_launch / (dk["ms_per_launch"] * 1e-3) / 1e12
        out = base_line(a, rk, "GRU training windows/sec (gru_train.py step, data parallel)", "windows/s", B * rk.world * a.steps / el,
                        el, f"f32; weight-gradient GEMMs on bf16x{a.split_bf16}-split operands, f32 accumulate (opt-in)" if a.split_bf16 else "f32", {"workload": "gru_train.py step RNN(188,128,4,24), Adam lr 1e-4, windows of 10", "batch_per_gpu": B,
                                    "seq_len": T, "global_batch": B * rk.world,
                                    "parallelism": f"dp{rk.world}, one flat fp32 bucket all-reduce", "baseline_config": "BASELINE.json configs[3]"})
        tot = fl_fwd + fl_sweep + fl_dw
        out["roofline"] = {"kernel": dk["kernel"] + " (v_mfma_f32_32x32x2_f32)", "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF,
                           "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF, "traffic": None, "traffic_source": "not collected for this mode", "avg_launch_ms": dk["ms_per_launch"],
                           "algorithmic_flops_per_launch": fl_launch,
                           "whole_step": {"algorithmic_TFLOP": tot / 1e12, "achieved_TFLOPs": tot / (el / a.steps) / 1e12,
                                          "frac": tot / (el / a.steps) / 1e12 / MFMA_F32_PEAK_TF}}
        out["kernels"] = kernels
        out["kernel_events"] = "HIP events in a second, untimed pass of the same step"
        out["parity"] = {"replica_max_abs_diff": divergence, "ranks": rk.world, "ok": bool(divergence == 0.0),
                         "what": "max_r |w_r - w_0| over the flat parameter vector after the timed steps (0 = replicas identical); "
                                 "single-step numerics vs the reference's loop body: tests/test_gpu_train.py (golden G6)"}
        out["rccl"] = dict(tr.rccl)
        out.update(allreduce_us=ar_us, grad_bucket_bytes=int(tr.bucket.g.numel() * 4), final_loss=float(loss.item()),
                   allreduce="two halves: layers L/2..L-1 + head on a side stream behind their dW kernel, the rest on the main stream"
                   if tr.split is not None else "one bucket behind the backward", **info)
        out["cpu_baseline"] = cpu_baseline_train(a.cpu_seconds, B) if (a.cpu_seconds > 0 and rk.world == 1) else None
        emit(json.dumps(out))


def bench_full(a, rk):
    """BASELINE configs[4]: 1024 depth frames (128 trajectories x 8 steps) -> ViT encoder latent (128-d) -> appended to the 60
    Kalman features -> GRU(188,128,4,24).  A 'step' is one pass over the 1024 frames."""
    import torch
    from optistate_amd import default_engine, RNN, flatten_state_dict
    from optistate_amd.transformer_model import Transformer_Autoencoder
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = rk.dev
    B, T = 128, 8
    eng = default_engine(rk.device_index); eng.set_noise(Q_DEFAULT, R_DEFAULT)     # the context the ViT module uses too
    torch.manual_seed(0)
    vit = Transformer_Autoencoder().to(dev)
    model = RNN(188, 128, 4, 24, dev)
    eng.load_gru(flatten_state_dict(model.state_dict(), 4, dev), 188, 128, 4, 24)
    d = synth_torch(B, T, dev, seed=7 + rk.rank)
    contact = eng.contact_soa_to_packed(d["contact"])
    frames = torch.rand(B * T, 1, 224, 224, device=dev)
    minmax = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).to(dev)

    def one():
        lat = vit.forward_encoder(frames).reshape(B, T, 128)
        x, P = d["x0"].clone(), d["P0"].clone()
        return eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], contact, d["accel"], minmax, x, P, latent=eng.pack(lat))
    el, _, _ = timed_region(rk, a.warmup, a.steps, one)
    kernels = events_pass(eng, max(2, min(a.steps, 5)), one)
    info = rk.report()
    if rk.rank == 0:
        N = B * T
        # ViT dense projections per frame (197 tokens, D = 128): qkv + proj + fc1 + fc2 per block x 3, plus the patch embedding
        fl_vit_gemm = N * (3 * 197 * 2 * (128 * 384 + 128 * 128 + 2 * 128 * 512) + 196 * 2 * 256 * 128)
        fl_attn = N * 3 * 4 * 2 * 2 * 197 * 197 * 32
        fl_gru = gru_flops_per_step(188, 128, 4) * N
        phase_flops = {"vit_gemm": fl_vit_gemm, "vit_attn": fl_attn, "gru_layer": fl_gru}
        cand = [k for k in kernels if k in phase_flops]
        out = base_line(a, rk, "depth frames/sec through ViT latent + KF + GRU", "frames/s", N * rk.world * a.steps / el, el, "f32",
                        {"workload": "ViT encoder (3 blocks, dim 128) latent + Kalman + GRU(188,128,4,24)", "frames": N,
                         "trajectories": B, "seq_len": T, "baseline_config": "BASELINE.json configs[4]",
                         "note": "ViT blocks checked against transformers' ViTLayer under the reference's glue (G11 / G14); timm 0.3.2 and the trained weights are absent"})
        if cand:
            dom = max(cand, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
            dk = kernels[dom]
            fl_launch = phase_flops[dom] / dk["launches_per_step"]
            ach = fl_launch / (dk["ms_per_launch"] * 1e-3) / 1e12
            kname = dk["kernel"] if dom != "vit_gemm" else "ViT projections: vit_gemm_kernel<...> (patch, first qkv) + vit_mlp_kernel[_bm64] (proj + LN + fc1 + GELU + fc2 + next LN + next qkv)"
            out["roofline"] = {"kernel": kname, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                               "frac": ach / MFMA_F32_PEAK_TF, "traffic": None, "traffic_source": "not collected for this mode", "avg_launch_ms": dk["ms_per_launch"],
                               "algorithmic_flops_per_launch": fl_launch, "phase": dom,
                               "note": "average over the launches of the phase (the projections differ in shape)"}
        out["kernels"] = kernels
        out["kernel_events"] = "HIP events in a second, untimed pass of the same step"
        out.update(info)
        out["cpu_baseline"] = cpu_baseline_full(a.cpu_seconds) if (a.cpu_seconds > 0 and rk.world == 1) else None
        emit(json.dumps(out))


def bench_mpc(a, rk):
    """estimate_state_mpc over the batch -- per step the convex-MPC force QP (exact float64 active-set
    solve, one wavefront per trajectory) followed by the predict_mpc/update filter step.  A 'step' is one pass over
    B trajectories x T time steps."""
    import torch
    from optistate_amd import Engine
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    dev = rk.dev
    B, T = a.batch, a.seq
    eng = Engine(rk.device_index); eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(B, T, dev, seed=11 + rk.rank)
    contact = eng.contact_soa_to_packed(d["contact"])
    ref = torch.zeros((T, 12, B), device=dev); ref[:, 5] = 0.28; ref[:, 9] = 0.1
    tt = torch.arange(T, device=dev)[:, None] * 0.01
    ref[:, 0] = 0.02 * torch.sin(3 * tt); ref[:, 1] = 0.02 * torch.cos(2 * tt)

    def one():
        x, P = d["x0"].clone(), d["P0"].clone()
        return eng.kf_mpc_run(d["p"], d["dp"], d["imu"], contact, ref, x, P, want_iters=True)
    el, last, _ = timed_region(rk, a.warmup, a.steps, one)
    kernels = events_pass(eng, 1, one)
    info = rk.report()
    if rk.rank == 0:
        it = last["iters"].float()
        out = base_line(a, rk, "KF timesteps/sec with the convex-MPC force QP in the loop (estimate_state_mpc)", "timesteps/s",
                        B * T * rk.world * a.steps / el, el, "f64 (QP) / f32 (filter)",
                        {"workload": "estimate_state_mpc: 60-variable force QP + Kalman(12/10) predict_mpc/update",
                         "batch_per_gpu": B, "seq_len": T, "parallelism": f"trajectory-sharded x{rk.world}, no collective",
                         "note": "QP formulation pinned to the reference's own assembly (G12); solver = KKT-certified minimiser (qpOASES absent)"})
        # The QP is float64 vector work (no matrix core has an fp64 path this small): roofline = algorithmic fp64 flops of the
        # active-set iterations actually taken against the fp64 vector peak.  Per iteration on n = 15 x (legs on the ground)
        # variables (mpc_kernels.hip: solve_face + the multiplier check): face-restricted rows n^2 x 11 + n x 200 (six
        # multiply-adds per entry from the 6-vector generators, the (alpha, beta) weights per horizon block), Gaussian elimination
        # 2 n^3 / 3 + n^2, back substitution 2 n^2, ratio test + gradient through the per-step generator sums ~60 n.
        cw = contact.view(torch.int32)
        nst = ((cw & 0xff) != 0).int() + (((cw >> 8) & 0xff) != 0).int() + (((cw >> 16) & 0xff) != 0).int() + (((cw >> 24) & 0xff) != 0).int()
        nvar = (15 * nst).double()
        fl_iter = 2.0 * nvar ** 3 / 3.0 + 14.0 * nvar ** 2 + 260.0 * nvar
        qp_flops = float((fl_iter * last["iters"].double()).sum())
        mk = kernels.get("mpc", {})
        mpc_ms = mk.get("ms_per_launch", 0.0) * mk.get("launches_per_step", 0.0)
        # (two parts of the batch on two streams: their launches overlap and the sum of their durations exceeds the pass -- the pass's
        # own time is the denominator then, filter steps included)
        overlapped = mpc_ms > el / a.steps * 1e3
        if overlapped:
            mpc_ms = el / a.steps * 1e3
        ach = qp_flops / (mpc_ms * 1e-3) / 1e12 if mpc_ms else 0.0
        kname = mk.get("kernel", "")
        persistent = kname.startswith("kf_mpc_persistent") or kname.startswith("kf_mpc_rows")
        quad = "quad" in kname or kname.startswith("kf_mpc_rows")
        out["roofline"] = {"kernel": mk.get("kernel", "mpc_solve_kernel") + " (float64 vector pipe: v_fma_f64)", "bound": "mfma", "pipe": "fp64 VECTOR pipe (compute-bound class of the contract's two; nothing here runs on a matrix core)", "achieved": ach,
                           "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": ach / FP64_VECTOR_PEAK_TF, "traffic": None,
                           "traffic_source": "not collected for this mode", "algorithmic_flops_per_pass": qp_flops,
                           "flops_per_iteration": "2 n^3 / 3 + 14 n^2 + 260 n, n = 15 x stance legs (30 at trot: 32 kflop)",
                           "device_ms_of_the_phase": mpc_ms,
                           "note": ("one kernel, a 16-lane row per trajectory for all T steps: the phase time includes the float64 filter step of every time step" if kname.startswith("kf_mpc_rows") else
                                    "one persistent kernel: the phase time includes the float64 filter step of every time step" if persistent else
                                    "the whole pass (QP launches of two parts of the batch overlap; the filter step runs inside them)" if overlapped else
                                    "QP launches, the filter step of every trajectory inside them" if "filter step inside" in kname else
                                    "QP launches only (the filter steps are the kf phase)"),
                           "limiter": ("sixteen lanes per QP, four QPs per wavefront, two wavefronts per SIMD at 256 registers: the vector pipe is busy ~50 % of "
                                       "a launch (profiles/r06_pmc_mpc.md: SQ_ACTIVE_INST_VALU against the launch time), the rest is LDS round trips "
                                       "between the phases of an active-set iteration and the long tail of the iteration count (mean 4.5, maximum ~45: "
                                       "the last quarter of a launch runs a few stragglers)" if quad else
                                       "latency: one QP per wavefront, a dependent chain of n pivots per elimination whose pivot rows cross lanes by "
                                       "v_readlane (profiles/r05_pmc_mpc.md: SQ_INSTS_VALU against SQ_WAIT_INST_ANY); 64 lanes execute, n - k do useful "
                                       "work at pivot k"),
                           "hbm_algorithmic_GBps": 268 * B * T / (el / a.steps) / 1e9}
        out.update(qp_iterations_mean=float(it.mean()), qp_iterations_max=int(it.max()),
                   status_nonzero_trajectories=int(eng.failed(last["status"]).sum()),
                   trunc_edge_trajectories=int(eng.trunc_edge(last["status"]).sum()), kernels=kernels, **info)
        out["cpu_baseline"] = cpu_baseline_mpc(min(a.cpu_seconds, 20.0)) if (a.cpu_seconds > 0 and rk.world == 1) else None
        emit(json.dumps(out))


def bench_windows(a, rk):
    """The reference's OWN inference mode (gru/gru_test.py:138-140,174-191): every output timestep re-runs a window of 10 steps from
    h0 = 0 through RNN(188,128,4,24).  Here all windows of one time-ordered ROW STREAM at once, without building the window tensor,
    with the first layer's input projection computed once per row (os_gru_forward_windows, pipeline.predict_rows); --materialise
    runs the round-4 form (unfold().contiguous() + os_gru_forward) for comparison."""
    import torch
    from optistate_amd import RNN
    from optistate_amd.pipeline import predict_windows, predict_rows
    dev = rk.dev
    N, T, I, H, L = a.batch, 10, 188, 128, 4
    torch.manual_seed(0)
    model = RNN(I, H, L, 24, dev).to(dev).eval()
    g = torch.Generator(device=dev); g.manual_seed(5 + rk.rank)
    rows = torch.rand(N + T - 1, I, device=dev, generator=g)          # N windows = N + T - 1 rows
    mn, mx = torch.zeros(12, device=dev), torch.ones(12, device=dev)
    if a.materialise:
        w = rows.unfold(0, T, 1).permute(0, 2, 1).contiguous()
        one = lambda: predict_windows(model, w, mn, mx)
    else:
        one = lambda: predict_rows(model, rows, T, mn, mx)
    el, _, _ = timed_region(rk, a.warmup, a.steps, one)
    kernels = events_pass(model._engine, max(2, min(a.steps, 5)), one)
    info = rk.report()
    if rk.rank == 0:
        out = base_line(a, rk, "sliding-window GRU outputs/sec (gru_test.py inference mode, window 10 from h0=0)", "windows/s",
                        N * rk.world * a.steps / el, el, "f32",
                        {"workload": "RNN(188,128,4,24) on the sliding windows of 10 of one row stream, one output per window (gru/gru_test.py:138-191)",
                         "windows_per_gpu": N, "rows_per_gpu": N + T - 1, "seq_len": T, "gru_timesteps_per_output": T,
                         "form": "materialised windows (B, 10, 188) + os_gru_forward" if a.materialise else
                                 "row stream (N + 9, 188) + os_gru_forward_windows: layer 0's x W_ih^T once per row"})
        dk = kernels.get("gru_layer")
        if dk:
            # algorithmic flops of THIS form: the stream form does layer 0's input half once per row instead of once per (window, step)
            fl_full = gru_flops_per_step(I, H, L) * N * T
            fl_in0 = 2 * 3 * H * I
            fl = fl_full if a.materialise else fl_full - fl_in0 * N * T + fl_in0 * (N + T - 1)
            ms_all = dk["ms_per_launch"] * dk["launches_per_step"]
            ach = fl / (ms_all * 1e-3) / 1e12
            out["roofline"] = {"kernel": dk["kernel"] + " (v_mfma_f32_32x32x2_f32; all GRU launches of a pass: " + str(dk["launches_per_step"]) + ")",
                               "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TF,
                               "traffic": None, "traffic_source": "not collected for this mode", "avg_launch_ms": dk["ms_per_launch"],
                               "algorithmic_flops_per_pass": fl, "flops_of_the_materialised_form": fl_full,
                               "note": "achieved = the form's own (reduced) flop count over the summed device time of its GRU launches"}
        out["kernels"] = kernels
        out.update(info)
        if a.cpu_seconds > 0 and rk.world == 1:
            cores = usable_cores(); torch.set_num_threads(cores)
            ref = torch.nn.GRU(I, H, L, batch_first=True); fc = torch.nn.Linear(H, 24)
            xb = torch.rand(2048, T, I)
            with torch.no_grad():
                torch.sigmoid(fc(ref(xb[:32])[0][:, -1]))
                t0 = time.perf_counter(); torch.sigmoid(fc(ref(xb)[0][:, -1])); tg = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 2048 / tg, "unit": "windows/s", "cores": cores, "kind": "port",
                                   "sample": f"2048 windows x 10, torch.nn.GRU(188,128,4)+Linear+sigmoid fp32 batched on the host ({tg:.2f} s); "
                                             "the reference script itself runs batch 1: 2.09 ms/window = 478 windows/s (SURVEY section 6)"}
        else:
            out["cpu_baseline"] = None
        emit(json.dumps(out))


def bench_hot_path(a, rk):
    """--mode fused (default, BASELINE configs[2]) and --mode kf (configs[1]-style)."""
    import numpy as np
    import torch
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, NOISE_SETS
    B, T, H, L, I = a.batch, a.seq, a.hidden, a.layers, 60 + a.latent
    fused = a.mode == "fused"
    eng = Engine(rk.device_index)
    dev = eng.device
    strong = a.scaling == "strong"
    B_global = B * rk.world
    if strong:
        # fixed global batch: every rank generates the SAME batch (seed 1000) and keeps its contiguous shard
        from optistate_amd.train import shard_range
        B_global = B
        lo, hi = shard_range(B_global, rk.rank, rk.world)
        d = synth_torch(B_global, T, dev, seed=1000, hostile=a.hostile)
        d = {k: (v[..., lo:hi].contiguous() if rk.world > 1 else v) for k, v in d.items()}
        B = hi - lo
        if B <= 0:
            raise SystemExit(f"--scaling strong: --batch {B_global} leaves rank {rk.rank} of {rk.world} without trajectories")
    else:
        d = synth_torch(B, T, dev, seed=1000 + rk.rank, hostile=a.hostile)
    contact = eng.contact_soa_to_packed(d["contact"])
    torch.manual_seed(0)
    model = RNN(I, H, L, 24, dev)                       # random-init weights of the named architecture
    eng.load_gru(flatten_state_dict(model.state_dict(), L, dev), I, H, L, 24)
    # --split-bf16: the single-kernel shape (GRU(60,64,1)) has its own fused bf16 kernel (os_fused_run flag); the H = 128 model
    # shapes run their layer launches on gru_layer_bf16_kernel (os_gru_set_split_bf16); other shapes have no bf16 path
    fused_bf16 = a.split_bf16 if (H == 64 and L == 1 and not a.latent) else 0
    if a.split_bf16 and not fused_bf16:
        if H != 128:
            raise SystemExit("--split-bf16: GRU(60,64,1) (fused kernel) or hidden 128 (layer kernels)")
        eng.set_gru_split_bf16(a.split_bf16)
    # min-max constants for the synthetic distributions (every feature lands in (0,1) like the reference's scaling)
    minmax = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).to(dev)
    x0 = d["x0"]
    x, P = x0.clone(), d["P0"].clone()
    # the ViT latent stream of the reference's real model (188 = 60 + 128 inputs, gru/gru_train.py:30-34): already in (0, 1)
    # (generated straight into rows 60.. of the GRU input buffer [T][60 + NL][B]: the Kalman kernel fills rows 0..59 in place,
    # Engine.gru_input_with_latent does the same from the encoder's (B, T, NL) output)
    gru_in = latent = None
    if a.latent:
        gru_in = torch.empty((T, 60 + a.latent, B), device=dev)
        gru_in[:, 60:] = torch.rand((T, a.latent, B), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
        latent = gru_in[:, 60:]

    def run_set(noise, warmup, steps):
        """One timed run under a noise set: Q / R into the context, P0 = Q as the reference's callers start
        (settings.py:31 `P = Q`; data_conversion_Kalman_to_Training.py:144 `KF2.P = copy.deepcopy(Q)`)."""
        Q, R = NOISE_SETS[noise]
        eng.set_noise(Q, R)
        P0 = torch.tensor(np.asarray(Q, dtype=np.float32).reshape(144, 1), device=dev).repeat(1, B).contiguous()

        def one_step():
            x.copy_(x0); P.copy_(P0)                    # device-to-device reset of the 40 MB filter state
            if fused:
                return eng.fused_run(d["p"], d["f"], d["dp"], d["imu"], contact, d["accel"], minmax, x, P, gru_input=gru_in,
                                     split_bf16=fused_bf16)
            if a.wave_per_trajectory:       # the north_star's literal layout, measured beside the default kernels (never the default)
                return eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], contact, x, P, sequential=False, symmetric=False, wave_per_trajectory=True)
            return eng.kf_run(d["p"], d["f"], d["dp"], d["imu"], contact, x, P)
        # one to three launches per pass here: the per-kernel HIP events stay on over the timed region itself (two event
        # records per launch against milliseconds of kernel; the modes with dozens of short launches per step use a second
        # pass instead)
        el, r, kernels = timed_region(rk, warmup, steps, one_step, eng=eng)
        return el, r, kernels, Q, R

    el, r, kernels, Q, R = run_set(a.noise, a.warmup, a.steps)
    bad = int(eng.failed(r["status"]).sum().item())           # bits 0-3; bit 4 is informational and counted on its own
    edge = int(eng.trunc_edge(r["status"]).sum().item())
    info = rk.report()
    checksum = None
    if strong:
        # the SAME samples whatever N: float64 sums of the final filter states and of the head outputs over the whole global batch
        # (1-GPU vs N-GPU runs of this line agree to fp32 noise: the filter bit for bit, the gate sums to their tile shape's k order)
        cs = torch.stack([r["x_out"][-1].double().sum(), (r["out"].double().sum() if fused else torch.zeros((), dtype=torch.float64, device=dev)),
                          torch.tensor(float(B), dtype=torch.float64, device=dev)])
        if rk.dist:
            cs = cs.cpu() if rk.share else cs
            rk.dist.all_reduce(cs)
        checksum = {"sum_x_final": float(cs[0]), "sum_out": float(cs[1]), "trajectories": int(cs[2])}
    if rk.rank != 0:
        return
    steps_per_pass = B * T
    total = (B_global * T if strong else steps_per_pass * rk.world) * a.steps
    # dominant kernel: the phase with the largest device time per pass; its NAME is the variant the library actually
    # launched (os_profile_kernel_name), so a small batch reports kf_run_rows_kernel, not the fast-path kernel
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_launch"] * kernels[k]["launches_per_step"])
    dk = kernels[dom]
    avg_ms = dk["ms_per_launch"]
    default_shape = B == 65536 and T == 100 and H == 64 and L == 1 and a.latent == 0
    if dom in ("gru_layer", "fused"):
        # MFMA-bound kernels: algorithmic flops = the GRU cell's matrix flops the launch performs (the Kalman
        # arithmetic of the fused kernel runs on the VALU and is not counted)
        fl = (gru_flops_per_step(I, H, L) if dom == "gru_layer" else gru_flops_per_step(I, H, 1)) * steps_per_pass / dk["launches_per_step"]
        if dom == "gru_layer" and "fused" in kernels:
            # the single fused kernel ran layer 0 (H = 64 stacks): the layer launches are layers 1 .. L-1
            fl = (gru_flops_per_step(I, H, L) - gru_flops_per_step(I, H, 1)) * steps_per_pass / dk["launches_per_step"]
        ach = fl / (avg_ms * 1e-3) / 1e12
        on_bf16 = bool(a.split_bf16) and "bf16" in dk["kernel"]
        peak = MFMA_BF16_PEAK_TF if on_bf16 else MFMA_F32_PEAK_TF
        insn = ("v_mfma_f32_32x32x16_bf16 x6 (hi/mid/lo split)" if a.split_bf16 == 3 else "v_mfma_f32_32x32x16_bf16 x3 (hi/lo split)") \
            if on_bf16 else ("v_mfma_f32_16x16x4_f32" if "kernel_v3" in dk["kernel"] else "v_mfma_f32_32x32x2_f32")
        roof = {"kernel": f"{dk['kernel']} ({insn})", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None, "traffic_source": "not collected for this mode", "avg_launch_ms": avg_ms, "algorithmic_flops_per_launch": fl}
        if on_bf16 and dom == "fused":
            roof["note"] = ("algorithmic (fp32-equivalent) GRU flops against the bf16 dense peak; the kernel is VALU-bound "
                            "(Kalman step + operand splitting), not matrix-bound")
        elif on_bf16:
            nprod = 6 if a.split_bf16 == 3 else 3
            roof["note"] = (f"algorithmic (fp32-equivalent) GRU flops against the bf16 dense peak; the kernel issues {nprod} bf16 products per "
                            f"fp32 product ({nprod} x these flops on the matrix pipe = {ach * nprod / peak:.2f} of the peak); power-bound: the chip "
                            "drops to ~1.7 GHz under this kernel's 68 % matrix-pipe occupancy (DESIGN 4.2f, profiles/r05_pmc_bf16_layer.txt)")
            roof["executed_bf16_TFLOPs"] = ach * nprod
        if dom == "fused":
            gbs = BYTES_PER_STEP_FUSED * steps_per_pass / (avg_ms * 1e-3) / 1e9
            roof["hbm_algorithmic_GBps"] = gbs
            roof["hbm_frac"] = gbs / HBM_PEAK_GBS
            roof["algorithmic_bytes"] = BYTES_PER_STEP_FUSED * steps_per_pass
    else:
        ach = BYTES_PER_STEP_KF * steps_per_pass / (avg_ms * 1e-3) / 1e9
        roof = {"kernel": dk["kernel"], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "traffic_source": "not collected for this mode", "avg_launch_ms": avg_ms,
                "algorithmic_bytes": BYTES_PER_STEP_KF * steps_per_pass}
        # the filter is ~35 flop/byte in its structured form: beside the HBM figure report the fp32 vector-pipe figure
        tf = KF_FLOPS_STRUCTURED * steps_per_pass / (avg_ms * 1e-3) / 1e12
        roof["valu_structured_TFLOPs"] = tf
        roof["valu_frac_of_fp32_vector_peak"] = tf / MFMA_F32_PEAK_TF
        if dk["kernel"].startswith("kf_run_rows"):
            per = 4
            waves = (B + per - 1) // per
            roof["limiter"] = (f"instruction issue of a serial chain: {waves} wavefronts on 1024 SIMDs, four trajectories (16 lanes each) "
                               f"per wavefront, T = {T} dependent steps of ~410 VALU instructions at ~5 cycles each "
                               f"({avg_ms * 1e3 / T:.2f} us per step: profiles/r03_rows2_timestamps.md); HBM is not the bound at this batch")
        else:
            roof["limiter"] = "VALU issue at one wavefront per SIMD (~5 cycles per instruction)"
    roof["frac_source"] = "HIP events on the launch stream over the timed region, this box"
    if default_shape and a.mode == "fused" and not a.split_bf16:
        # beside the live figure: the same fraction from the tracked rocprofv3 summary of the same command
        rp_ms, rp_src = rocprof_avg_ms(dk["kernel"])
        roof["frac_rocprof"] = (roof["achieved"] * avg_ms / rp_ms) / roof["peak"] if rp_ms else None
        roof["rocprof_avg_launch_ms"] = rp_ms
        roof["rocprof_source"] = rp_src
    rows_shape = a.mode == "kf" and B == 4096 and T == 1000 and dk["kernel"].startswith("kf_run_rows2")      # the shape traffic.json holds for it
    roof["traffic"], roof["traffic_source"] = load_traffic(dk["kernel"], (default_shape or rows_shape) and not a.split_bf16)
    if roof["traffic"] is not None:
        roof["traffic_unit"] = "bytes/launch"
    for k in kernels:
        if k == "kf":
            kernels[k]["algorithmic_GBps"] = BYTES_PER_STEP_KF * steps_per_pass / (kernels[k]["ms_per_launch"] * 1e-3) / 1e9
            kernels[k]["hbm_frac"] = kernels[k]["algorithmic_GBps"] / HBM_PEAK_GBS
    dtype = f"bf16x{a.split_bf16}-split operands, f32 accumulate (opt-in; KF in f32)" if a.split_bf16 else "f32"
    out = base_line(a, rk, "KF+GRU timesteps/sec" if fused else "KF timesteps/sec", "timesteps/s", total / el, el, dtype,
                    {"workload": f"fused Kalman(12-state/10-meas)+GRU(in={I},hidden={H},layers={L},out=24) inference"
                                 if fused else "Kalman(12-state/10-meas) predict/update only",
                     "batch_per_gpu": B, "seq_len": T, "global_batch": B_global,
                     "parallelism": f"trajectory-sharded x{rk.world}, no collective" + (" (contiguous shards of ONE fixed batch)" if strong else ""),
                     "noise": NOISE_NOTE[a.noise], "inputs": INPUT_NOTE["hostile" if a.hostile else "nominal"],
                     "baseline_config": "BASELINE.json configs[2]" if fused else "BASELINE.json configs[1] (true dims 12/10)"})
    out["roofline"] = roof
    if strong:
        out["global_checksum"] = checksum
        out["strong_note"] = ("fixed global batch: rank r runs trajectories shard_range(B, r, N) of the same synthetic batch; roofline / kernels / "
                              "parity are rank 0's shard; os_fused_run chose the kernel tile shape for the shard size")
    out["kernels"] = kernels
    out["kernel_events"] = "HIP events on the launch stream, recorded over the timed region (the wall time includes them)"
    out["status_nonzero_trajectories"] = bad
    out["trunc_edge_trajectories"] = edge
    out["status_note"] = ("status_nonzero_trajectories counts FAILURES (status bits 0-3); trunc_edge_trajectories counts the "
                          "informational bit 4 (int64-truncation knife edge, include/optistate_hip.h)")
    out.update(info)
    # parity of the timed batch itself (rank 0's shard) over ALL of its (trajectory, timestep) pairs, outside the timed
    # region; the oracle's seconds double as the all-cores cpu_baseline (one run serves both)
    all_cores = None
    cap = max(4.0 * a.cpu_seconds, 60.0)
    if a.parity_samples != 0:
        want = B if a.parity_samples < 0 else min(a.parity_samples, B)
        if rk.world > 1 and a.parity_samples < 0:
            want = min(B, 8192)            # N ranks share the host cores (torchrun pins OMP_NUM_THREADS): a sample of rank 0's shard
        out["parity"], timing = oracle_pass(d, r["x_out"], r.get("out"), model, H, L, Q, R, Q, want, cap, kf_only=not fused, latent=latent)
        if timing["trajectories"] >= 2048:
            all_cores = timing
        if a.split_bf16:
            out["parity"]["gru_bar"] = 1e-5
            out["parity"]["note"] = "opt-in split-bf16 gate GEMM: the GRU bar stays 1e-5, the fused chain < 1e-4"
    if a.second_noise and a.noise == "default" and rk.world == 1 and not a.split_bf16:
        # "second run with Q_R.pkl values" -- the reference's fitted Q / R (cond(S) ~ 1e6) on the same inputs
        el2, r2, k2, Q2, R2 = run_set("fitted", 1, 3)
        sec = {"noise": NOISE_NOTE["fitted"], "steps": 3, "ms_per_step": el2 / 3 * 1e3, "value": steps_per_pass * rk.world * 3 / el2,
               "status_nonzero_trajectories": int(eng.failed(r2["status"]).sum().item()),
               "trunc_edge_trajectories": int(eng.trunc_edge(r2["status"]).sum().item())}
        if a.parity_samples != 0:
            want2 = B if a.parity_samples < 0 else min(a.parity_samples, B)
            sec["parity"], _ = oracle_pass(d, r2["x_out"], r2.get("out"), model, H, L, Q2, R2, Q2, want2, cap, kf_only=not fused, latent=latent)
        out["second_noise_set"] = sec
    if a.cpu_seconds > 0 and rk.world == 1:
        out["cpu_baseline"] = cpu_baseline(H, L, a.cpu_seconds, kf_only=not fused, all_cores=all_cores if a.latent == 0 else None)
        if a.latent:
            out["cpu_baseline"]["note"] = "sampled on the 60-wide input (no latent stream)"
    else:
        out["cpu_baseline"] = None
    emit(json.dumps(out))


def launch_check(a, rk):
    """Rendezvous self-test of the launcher (CPU contract test): every rank joins, one all-reduce, rank 0 reports.  With
    --scaling strong every rank also reports the shard of the global batch it would run."""
    import torch
    v = torch.ones(1, device=rk.dev)
    if rk.dist:
        rk.dist.all_reduce(v)
    info = rk.report()
    extra = {}
    if a.scaling == "strong":
        from optistate_amd.train import shard_range
        mine = list(shard_range(a.batch, rk.rank, rk.world))
        shards = [None] * rk.world
        if rk.dist:
            rk.dist.all_gather_object(shards, mine)
        else:
            shards = [mine]
        extra = {"scaling": "strong", "global_batch": a.batch, "shards": shards}
    if rk.rank == 0:
        emit(json.dumps({"launch_check": True, "n_gpus": a.gpus, "sum_of_ones": float(v.item()), **extra, **info}))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed passes; default 10 (windows: 100 -- a pass is 0.6 ms, ten of them time the clock ramp)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed passes; default 2 (windows: 20)")
    ap.add_argument("--batch", type=int, default=None, help="trajectories (windows) per GPU; default 65536 (windows, train: 8192)")
    ap.add_argument("--seq", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--latent", type=int, default=0, help="width of a latent stream appended to the 60 Kalman features (128: the reference's 188-wide GRU input)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget (0 = skip)")
    ap.add_argument("--parity-samples", type=int, default=-1,
                    help="trajectories of the timed batch checked against the oracle over all their timesteps (-1 = the whole batch, 0 = skip)")
    ap.add_argument("--noise", default="default", choices=["default", "fitted"],
                    help="Q / R set: settings.py defaults, or the reference's fitted Q_R.pkl values with R[0:3] = 1e-4")
    ap.add_argument("--no-second-noise", dest="second_noise", action="store_false",
                    help="skip the short second run under the fitted noise set that the default line carries as second_noise_set")
    ap.add_argument("--hostile", action="store_true",
                    help="inputs that leave the fast branches: 0-4 stance legs per step, yaw unwrapping past +-pi, wide roll / pitch")
    ap.add_argument("--split-bf16", type=int, nargs="?", const=3, default=0, choices=[0, 2, 3],
                    help="opt-in gate GEMM on the bf16 MFMA with 3 (default) or 2 bf16 terms per fp32 operand (second line; never the headline)")
    ap.add_argument("--wave-per-trajectory", action="store_true",
                    help="--mode kf: one wavefront per trajectory with x and P in LDS (the north_star's literal layout; measurement only)")
    ap.add_argument("--materialise", action="store_true",
                    help="--mode windows: build the (B, 10, 188) window tensor and call os_gru_forward (the round-4 form) instead of the row-stream entry")
    ap.add_argument("--force-dist", action="store_true",
                    help="--mode train on ONE GPU with a one-rank RCCL process group: exercises (and prices) the split all-reduce path")
    ap.add_argument("--no-split-allreduce", action="store_true", help="--mode train: one all-reduce behind the whole backward")
    ap.add_argument("--rccl-proto", default="default", choices=["default", "LL", "LL128", "Simple"],
                    help="NCCL_PROTO for the ranks (set before any communicator exists); the 1.69 MB gradient bucket is latency-bound")
    ap.add_argument("--rccl-algo", default="default", choices=["default", "Ring", "Tree"], help="NCCL_ALGO for the ranks")
    ap.add_argument("--share-gpu", action="store_true",
                    help="development: with --gpus N on a box with fewer devices, rank r runs on device r %% device_count over gloo "
                         "(host-staged collectives); runs the world > 1 code paths, not a scaling measurement")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --batch trajectories PER GPU; strong (modes fused / kf): --batch is the global batch, rank r runs its "
                         "contiguous shard of the same synthetic batch (SURVEY 8(e))")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, rendezvous, and report the process group")
    ap.add_argument("--mode", default="fused", choices=["fused", "kf", "train", "full", "mpc", "windows"],
                    help="kf = BASELINE configs[1]-style KF-only run; train = configs[3] data-parallel gru_train step")
    a = ap.parse_args(argv)
    if a.batch is None:
        a.batch = 8192 if a.mode in ("windows", "train") else 65536
    if a.steps is None:
        a.steps = 100 if a.mode == "windows" else 10
    if a.warmup is None:
        a.warmup = 20 if a.mode == "windows" else 2
    if a.scaling == "strong" and a.mode not in ("fused", "kf") and not a.launch_check:
        raise SystemExit("--scaling strong: modes fused and kf (the trajectory-sharded inference paths)")

    # N > 1 and not yet a rank: start the ranks as a child process BEFORE anything here touches the GPU
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a.gpus, argv))

    if "WORLD_SIZE" in os.environ or a.force_dist:
        protect_stdout()
    rk = Ranks(a)
    try:
        if a.launch_check:
            launch_check(a, rk)
        elif a.mode == "train":
            bench_train(a, rk)
        elif a.mode == "full":
            bench_full(a, rk)
        elif a.mode == "mpc":
            bench_mpc(a, rk)
        elif a.mode == "windows":
            bench_windows(a, rk)
        else:
            bench_hot_path(a, rk)
    finally:
        rk.close()


if __name__ == "__main__":
    main()
