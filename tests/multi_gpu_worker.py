"""Worker of tests/test_gpu_multi.py: run as `python -m torch.distributed.run --nproc-per-node 2 tests/multi_gpu_worker.py`.

SURVEY.md 8(e) determinism check on real GPUs: the SAME samples processed by 1 GPU x B and by 2 GPUs x B/2.
  * inference (no collective): every rank runs its contiguous shard of the trajectory axis through the fused kernel; the
    gathered shard outputs must equal rank 0's full-batch run;
  * training (one flat-bucket all-reduce over RCCL): DataParallelTrainer on shards vs a single-process full-batch step --
    gradients equal to fp32 reduction-order noise, identical replicas after the Adam update.
With OS_SHARE_GPU=1 every rank uses device (local_rank % device_count) and the process group runs over gloo with the
collectives staged through the host (optistate_amd.train.all_reduce_): the SAME code paths -- shards, the weighted flat
bucket, the split all-reduce behind os_gru_backward_mark, the replica check -- on the one GPU a development box has (RCCL
refuses two ranks on one device).  OS_WORKER_TRAIN_BATCH sets the global training batch (a value that does not divide by the
world size makes the last shards ragged).  Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    share = os.environ.get("OS_SHARE_GPU", "0") == "1"
    di = lr % torch.cuda.device_count() if share else lr         # device_count() does not initialise the GPU
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    from optistate_amd.train import DataParallelTrainer, shard_range, all_gather_cat
    res = {"world": dist.get_world_size(), "backend": dist.get_backend()}
    devs = [None] * world
    dist.all_gather_object(devs, {"rank": rank, "pid": os.getpid(), "device": di,
                                  "pci_bus_id": getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None)})
    res["rank_devices"] = devs

    # ---------------- inference: trajectory sharding, no collective on the data path ----------------
    B, T = 8192 * world, 12
    eng = Engine(di)
    eng.set_noise(Q_DEFAULT, R_DEFAULT)
    d = synth_torch(B, T, dev, seed=4242)                 # same seed on every rank: the same samples (philox)
    cp = eng.contact_soa_to_packed(d["contact"])
    torch.manual_seed(0)
    model = RNN(60, 64, 1, 24, torch.device("cpu"))
    eng.load_gru(flatten_state_dict(model.state_dict(), 1, dev), 60, 64, 1, 24)
    mm = torch.stack([torch.full((60,), -30.0), torch.full((60,), 30.0)]).to(dev)

    def run(lo, hi):
        sl = lambda k: d[k][:, :, lo:hi].contiguous()
        x, P = d["x0"][:, lo:hi].contiguous(), d["P0"][:, lo:hi].contiguous()
        r = eng.fused_run(sl("p"), sl("f"), sl("dp"), sl("imu"), cp[:, lo:hi].contiguous(), sl("accel"), mm, x, P, two_kernel=False)
        return r["x_out"], r["out"], r["status"]
    lo, hi = shard_range(B, rank, world)
    xs, os_, st = run(lo, hi)
    gx, go = all_gather_cat(xs, dim=2), all_gather_cat(os_, dim=0)      # host-side gather of outputs, NOT part of the data path
    if rank == 0:
        xf, of, sf = run(0, B)
        res["infer_state_max_abs_diff"] = float((gx - xf).abs().max())
        res["infer_out_max_abs_diff"] = float((go - of).abs().max())
        res["infer_status_nonzero"] = int(eng.failed(sf).sum())

    # ---------------- training: one flat gradient bucket all-reduced over RCCL ----------------
    Bt, Tt, dims = int(os.environ.get("OS_WORKER_TRAIN_BATCH", 2048 * world)), 10, (188, 128, 4, 24)
    g = torch.Generator(device=dev); g.manual_seed(7)
    x = torch.rand(Bt, Tt, dims[0], device=dev, generator=g); y = torch.rand(Bt, 12, device=dev, generator=g)
    torch.manual_seed(3)
    m = RNN(*dims, dev).to(dev)
    tr = DataParallelTrainer(m, lr=1e-4)
    lo, hi = shard_range(Bt, rank, world)
    res["train_shard_sizes"] = [shard_range(Bt, r, world)[1] - shard_range(Bt, r, world)[0] for r in range(world)]
    res["train_split_allreduce"] = tr.split is not None
    tr.step(x[lo:hi], y[lo:hi])
    g_dp, w_dp = tr.bucket.g.clone(), tr.bucket.w.clone()
    ws = all_gather_cat(w_dp.reshape(1, -1), dim=0)
    res_w = float((ws - ws[0:1]).abs().max())
    res_div = tr.replica_divergence()                    # the product's own check: broadcast + MAX all-reduce
    dist.barrier()
    if rank == 0:
        # single-process full-batch step: same init, no process group involved (world-1 semantics by construction)
        from optistate_amd.train import FlatBucket, _flat_order_params
        torch.manual_seed(3)
        m1 = RNN(*dims, dev).to(dev)
        bucket = FlatBucket(_flat_order_params(m1), dev)
        e = tr.eng
        e.load_gru(bucket.w, *dims, m1.use_sigmoid)
        out = e.gru_forward_train(x)
        _, dout, _ = e.gru_loss(out, y)
        e.gru_backward(x, out, dout, grad_flat=bucket.g)
        scale = float(bucket.g.abs().max())
        res["train_grad_max_abs_diff"] = float((g_dp - bucket.g).abs().max())
        res["train_grad_scale"] = scale
        res["train_replica_weight_max_abs_diff"] = res_w
        res["train_replica_divergence"] = res_div
        # the weights after the data-parallel step against the single-process full-batch Adam step
        m_, v_ = torch.zeros_like(bucket.w), torch.zeros_like(bucket.w)
        e.adam_step(bucket.w, bucket.g, m_, v_, 1e-4, 0.9, 0.999, 1e-8, 1)
        res["train_weight_vs_single_process_max_abs_diff"] = float((w_dp - bucket.w).abs().max())
        res["grad_bucket_bytes"] = int(bucket.g.numel() * 4)
        print(json.dumps(res), flush=True)
    dist.barrier()

    # ---------------- a lost producer on ONE rank (OS_WORKER_DROP_RANK): the trainer's whole-step redo ----------------
    # The reference's own batch (64 windows per rank) runs its layers as progress-counter launches.  A fresh context on the chosen rank
    # is created with OS_STACK_DBG_DROP (layer 1 stops publishing at step 3) and a short bounded wait: that rank's forward and backward
    # lose a producer in EVERY step.  DataParallelTrainer verifies once per step before the all-reduce and redoes the step with a
    # launch per layer, so no poisoned gradient reaches the collective: replicas stay identical and equal the clean run's weights.
    drop_rank = int(os.environ.get("OS_WORKER_DROP_RANK", "-1"))
    if drop_rank >= 0:
        from optistate_amd import engine as _engine
        if rank == drop_rank:
            os.environ["OS_STACK_DBG_DROP"] = "1,3"; os.environ["OS_STACK_DBG_POLLS"] = "3000"
        _engine._default_engines.__dict__.pop("engines", None)      # a new per-thread default context: os_create reads the knobs
        solo = [dist.new_group([r]) for r in range(world)]      # (collective: every rank creates every one-rank group)
        Bs = 64 * world
        xs_, ys_ = torch.rand(Bs, Tt, dims[0], device=dev, generator=g), torch.rand(Bs, 12, device=dev, generator=g)
        torch.manual_seed(5)
        m2 = RNN(*dims, dev).to(dev)
        tr2 = DataParallelTrainer(m2, lr=1e-4)
        lo, hi = shard_range(Bs, rank, world)
        for _ in range(3):
            loss = tr2.step(xs_[lo:hi], ys_[lo:hi])
        torch.cuda.synchronize()
        lost = [None] * world
        dist.all_gather_object(lost, {"rank": rank, "lost_steps": tr2.lost_steps, "loss_finite": bool(torch.isfinite(loss).all()),
                                      "stack_mode_after": tr2.eng._stack_mode})
        w2 = all_gather_cat(tr2.bucket.w.reshape(1, -1), dim=0)
        if rank == 0:
            # the clean run: one process, the full batch, no drop (rank 0's context has none), three steps
            torch.manual_seed(5)
            m3 = RNN(*dims, dev).to(dev)
            tr3 = DataParallelTrainer(m3, lr=1e-4, group=solo[0])
            for _ in range(3):
                tr3.step(xs_, ys_)
            torch.cuda.synchronize()
            res2 = {"lost": lost, "weights_finite": bool(torch.isfinite(w2).all()), "replica_weight_max_abs_diff": float((w2 - w2[0:1]).abs().max()),
                    "weight_vs_clean_run_max_abs_diff": float((w2[0] - tr3.bucket.w).abs().max())}
            print("LOST " + json.dumps(res2), flush=True)
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
