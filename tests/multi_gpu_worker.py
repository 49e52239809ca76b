"""Worker of tests/test_gpu_multi.py: run as `python -m torch.distributed.run --nproc-per-node 2 tests/multi_gpu_worker.py`.

SURVEY.md 8(e) determinism check on real GPUs: the SAME samples processed by 1 GPU x B and by 2 GPUs x B/2.
  * inference (no collective): every rank runs its contiguous shard of the trajectory axis through the fused kernel; the
    gathered shard outputs must equal rank 0's full-batch run;
  * training (one flat-bucket all-reduce over RCCL): DataParallelTrainer on shards vs a single-process full-batch step --
    gradients equal to fp32 reduction-order noise, identical replicas after the Adam update.
Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, lr = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(lr)
    dev = torch.device("cuda", lr)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from optistate_amd import Engine, RNN, flatten_state_dict
    from optistate_amd.synth import synth_torch, Q_DEFAULT, R_DEFAULT
    from optistate_amd.train import DataParallelTrainer, shard_range
    res = {"world": dist.get_world_size(), "backend": dist.get_backend()}

    # ---------------- inference: trajectory sharding, no collective on the data path ----------------
    B, T = 8192 * world, 12
    eng = Engine(lr)
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
    gx = [torch.empty_like(xs) for _ in range(world)]; go = [torch.empty_like(os_) for _ in range(world)]
    dist.all_gather(gx, xs); dist.all_gather(go, os_)      # host-side style gather of outputs, NOT part of the data path
    if rank == 0:
        xf, of, sf = run(0, B)
        res["infer_state_max_abs_diff"] = float((torch.cat(gx, dim=2) - xf).abs().max())
        res["infer_out_max_abs_diff"] = float((torch.cat(go, dim=0) - of).abs().max())
        res["infer_status_nonzero"] = int((sf != 0).sum())

    # ---------------- training: one flat gradient bucket all-reduced over RCCL ----------------
    Bt, Tt, dims = 2048 * world, 10, (188, 128, 4, 24)
    g = torch.Generator(device=dev); g.manual_seed(7)
    x = torch.rand(Bt, Tt, dims[0], device=dev, generator=g); y = torch.rand(Bt, 12, device=dev, generator=g)
    torch.manual_seed(3)
    m = RNN(*dims, dev).to(dev)
    tr = DataParallelTrainer(m, lr=1e-4)
    lo, hi = shard_range(Bt, rank, world)
    tr.step(x[lo:hi], y[lo:hi])
    g_dp, w_dp = tr.bucket.g.clone(), tr.bucket.w.clone()
    ws = [torch.empty_like(w_dp) for _ in range(world)]
    dist.all_gather(ws, w_dp)
    res_w = max(float((w - ws[0]).abs().max()) for w in ws)
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
        res["grad_bucket_bytes"] = int(bucket.g.numel() * 4)
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
