"""CPU, world_size 2, gloo: the host-side sharding and flat-bucket logic of the multi-GPU paths.

Inference shards the trajectory axis with no collective; training all-reduces ONE flat gradient bucket.
The HIP kernels cannot run here, so the per-rank gradients come from torch's own GRU (test-only stand-in); what
is under test is optistate_amd.train.FlatBucket / shard_range and their equivalence to a single-process run."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optistate_amd.train import FlatBucket, shard_range
        torch.manual_seed(0)                               # identical replicas
        gru = torch.nn.GRU(12, 16, 2, batch_first=True); fc = torch.nn.Linear(16, 8)
        params = list(gru.parameters()) + list(fc.parameters())
        B = 37                                             # ragged on purpose
        g = torch.Generator().manual_seed(1)
        x = torch.rand(B, 5, 12, generator=g); y = torch.rand(B, 4, generator=g)

        def loss_of(xs, ys, scale):
            o, _ = gru(xs)
            out = torch.sigmoid(fc(o[:, -1, :]))
            tgt = torch.cat([ys, (out[:, :4].detach() - ys).abs()], dim=1)
            return ((out - tgt) ** 2).sum() / scale

        bucket = FlatBucket(params)
        lo, hi = shard_range(B, rank, world)
        # each rank's loss is normalised by the GLOBAL element count times 1/world so that the bucket MEAN equals
        # the full-batch gradient even with unequal shards
        bucket.g.zero_()
        loss = loss_of(x[lo:hi], y[lo:hi], B * 8 / world)
        grads = torch.autograd.grad(loss, params)
        off = 0
        for p, gr in zip(params, grads):
            bucket.g[off:off + p.numel()].copy_(gr.reshape(-1)); off += p.numel()
        bucket.allreduce_mean_()
        # single-process reference on the full batch
        full = torch.autograd.grad(loss_of(x, y, B * 8), params)
        ref = torch.cat([t.reshape(-1) for t in full])
        err = (bucket.g - ref).abs().max().item()
        # parameters are views of the flat tensor: an in-place update of the bucket is visible through state_dict
        bucket.w.add_(1.0)
        view_ok = all(torch.equal(p.data.reshape(-1), bucket.w[o:o + p.numel()]) for p, o in
                      zip(params, [sum(q.numel() for q in params[:i]) for i in range(len(params))]))
        ret[rank] = (err, lo, hi, view_ok)
    finally:
        dist.destroy_process_group()


def test_flat_bucket_allreduce_equals_full_batch_gradient():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    spans = sorted((ret[r][1], ret[r][2]) for r in range(world))
    assert spans[0][0] == 0 and spans[-1][1] == 37 and spans[0][1] == spans[1][0]      # disjoint cover
    for r in range(world):
        assert ret[r][0] < 1e-6, ret[r]
        assert ret[r][3]


def _worker_weighted(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optistate_amd.train import FlatBucket, shard_range
        torch.manual_seed(0)
        gru = torch.nn.GRU(12, 16, 2, batch_first=True); fc = torch.nn.Linear(16, 8)
        params = list(gru.parameters()) + list(fc.parameters())
        B = 41                                             # 14 + 14 + 13 over three ranks
        g = torch.Generator().manual_seed(2)
        x = torch.rand(B, 5, 12, generator=g); y = torch.rand(B, 4, generator=g)

        def local_mean_loss(xs, ys):                       # what DataParallelTrainer's loss kernel computes: MSE over the rank's own samples
            o, _ = gru(xs)
            out = torch.sigmoid(fc(o[:, -1, :]))
            tgt = torch.cat([ys, (out[:, :4].detach() - ys).abs()], dim=1)
            return torch.nn.functional.mse_loss(out, tgt)

        bucket = FlatBucket(params)
        lo, hi = shard_range(B, rank, world)
        grads = torch.autograd.grad(local_mean_loss(x[lo:hi], y[lo:hi]), params)
        off = 0
        for p, gr in zip(params, grads):
            bucket.g[off:off + p.numel()].copy_(gr.reshape(-1)); off += p.numel()
        plain = bucket.g.clone()
        bucket.allreduce_weighted_(hi - lo)
        ref = torch.cat([t.reshape(-1) for t in torch.autograd.grad(local_mean_loss(x, y), params)])
        err_w = (bucket.g - ref).abs().max().item()
        # the unweighted mean of the local-mean gradients is NOT the full-batch gradient when shards differ
        dist.all_reduce(plain); plain /= world
        err_plain = (plain - ref).abs().max().item()
        ret[rank] = (err_w, err_plain, hi - lo, float(ref.abs().max()))
    finally:
        dist.destroy_process_group()


def test_weighted_bucket_allreduce_with_ragged_shards_world_3():
    """Three ranks, 41 samples (14 / 14 / 13): the size-weighted single all-reduce reproduces the full-batch mean gradient
    from the ranks' LOCAL-mean gradients (what the loss kernel produces); the plain mean over ranks does not."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + os.getpid() % 1000
    mp.spawn(_worker_weighted, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret[r][2] for r in range(world)) == [13, 14, 14]
    for r in range(world):
        err_w, err_plain, _, scale = ret[r]
        assert err_w < 1e-6 * max(1.0, scale), ret[r]
        assert err_plain > 10 * err_w                      # the test would notice a trainer that averaged unweighted


def test_shard_range_covers_every_unit_once():
    from optistate_amd.train import shard_range
    for n in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_replica(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from optistate_amd.train import replica_divergence, configure_rccl
        w = torch.arange(1000, dtype=torch.float32) * 1e-3
        same = replica_divergence(w)
        if rank == 1:
            w[617] += 2.5e-4                                   # one replica drifts in one parameter
        drift = replica_divergence(w)
        try:
            configure_rccl("LL")                               # too late: the group exists
            late = "accepted"
        except RuntimeError:
            late = "refused"
        ret[rank] = (same, drift, late)
    finally:
        dist.destroy_process_group()


def test_replica_divergence_world_2():
    """bench.py --mode train reports max_r |w_r - w_0| after the timed steps: every rank sees the same
    figure, 0.0 for identical replicas, the drift otherwise; the RCCL protocol cannot be changed once the group exists."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_worker_replica, args=(world, 29700 + os.getpid() % 1000, ret), nprocs=world, join=True)
    for r in range(world):
        same, drift, late = ret[r]
        assert same == 0.0 and abs(drift - 2.5e-4) < 1e-7 and late == "refused"


def test_rccl_env_knob():
    from optistate_amd.train import rccl_env, configure_rccl, replica_divergence
    assert rccl_env() == {} and rccl_env("LL128", "Tree") == {"NCCL_PROTO": "LL128", "NCCL_ALGO": "Tree"}
    with pytest.raises(ValueError):
        rccl_env("fast")
    assert replica_divergence(torch.zeros(4)) == 0.0          # no process group: a single replica
    old = {k: os.environ.get(k) for k in ("NCCL_PROTO", "NCCL_ALGO")}
    try:
        assert configure_rccl("LL") == {"NCCL_PROTO": "LL"} and os.environ["NCCL_PROTO"] == "LL"
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
