"""Training side of the GRU head (reference gru/gru_train.py:217-251) on the HIP kernels.

* `gru_forward_autograd`: what `RNN.forward` calls when gradients are enabled, so the reference's own loop
  (`outputs = model(inputs); loss = criterion(...); loss.backward(); optimizer.step()`) runs unchanged with
  torch.optim.Adam: forward and backward are the HIP kernels, torch only does the autograd bookkeeping.
* `FlatBucket` / `DataParallelTrainer`: the data-parallel step of BASELINE config 4 -- one process per GPU, identical
  replicas, ONE flat fp32 gradient bucket (422,424 floats at the reference config) all-reduced over RCCL
  (`torch.distributed`, backend "nccl"), then identical fused-Adam updates.  The bucket is latency-bound at 1.69 MB, so
  there is no bucketing/overlap machinery.  The target of gru_train.py:237-244 is built on the
  device (no host round trip, no per-sample Python loop).
"""
import os

import torch
import torch.distributed as dist

# RCCL reads these when a communicator is created: the 1.69 MB gradient bucket of BASELINE config 4 is latency-bound on
# xGMI, so the protocol (LL: 8-byte flagged stores, lowest latency; LL128: 128-byte lines over
# xGMI; Simple: bandwidth) and the algorithm (Ring / Tree) are a run-time choice, priced per node by tools/scale_sweep.sh.
RCCL_PROTOS = ("default", "LL", "LL128", "Simple")
RCCL_ALGOS = ("default", "Ring", "Tree")


def rccl_env(proto="default", algo="default"):
    """Environment entries that select RCCL's protocol / algorithm ({} for the library's own size-based tuning)."""
    if proto not in RCCL_PROTOS or algo not in RCCL_ALGOS:
        raise ValueError(f"rccl_env: proto in {RCCL_PROTOS}, algo in {RCCL_ALGOS}")
    env = {}
    if proto != "default":
        env["NCCL_PROTO"] = proto
    if algo != "default":
        env["NCCL_ALGO"] = algo
    return env


def configure_rccl(proto="default", algo="default"):
    """Apply rccl_env() to THIS process.  Must run before the process group exists: RCCL reads the variables when the
    communicator is created, a later change is silently ignored -- hence the error instead."""
    env = rccl_env(proto, algo)
    if env and dist.is_available() and dist.is_initialized():
        raise RuntimeError("configure_rccl: the process group already exists; set the protocol before init_process_group "
                           "(bench.py --rccl-proto does it in the ranks' environment)")
    os.environ.update(env)
    return env


def _host_staged(t, group=None):
    """True when the collective must go through host memory: a device tensor under the gloo backend.  That is the
    `--share-gpu` development shape -- several ranks on ONE MI355X, where RCCL refuses the duplicate device -- which runs the
    world > 1 code paths (shards, weighted bucket, split all-reduce behind os_gru_backward_mark, replica check) with real
    device tensors on a one-GPU box.  RCCL ("nccl") always takes the device tensor directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_reduce_(t, op=None, group=None):
    """dist.all_reduce on t in place, stream-ordered for device tensors on either backend: RCCL enqueues on the current
    stream; under gloo the tensor is copied to the host on the CURRENT stream (which the copy waits for), reduced there and
    copied back on the same stream."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t, group):
        h = t.detach().to("cpu")
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast_(t, src=0, group=None):
    if _host_staged(t, group):
        h = t.detach().to("cpu")
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


def all_gather_cat(t, dim=0, group=None):
    """Concatenation over the ranks of (possibly ragged along `dim`) shard tensors: the host-side gather of outputs after a
    sharded inference run (NOT part of the data path).  Goes through the object collective, so it works on every backend."""
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, t.detach().cpu(), group=group)
    return torch.cat(parts, dim=dim).to(t.device)


def replica_divergence(flat_w, group=None):
    """max_r |w_r - w_0| over the ranks of a data-parallel job (identical replicas must stay bit-identical: same averaged
    gradient, same fused Adam).  One broadcast + one MAX all-reduce; 0.0 without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0.0
    ref = flat_w.detach().clone()
    broadcast_(ref, src=0, group=group)
    d = (flat_w.detach() - ref).abs().max().reshape(1)
    all_reduce_(d, op=dist.ReduceOp.MAX, group=group)
    return float(d.item())


def _flat_order_params(module):
    """Parameters in the flat layout's order: per layer weight_ih, weight_hh, bias_ih, bias_hh; then fc.weight, fc.bias."""
    ps = []
    for l in range(module.num_layers):
        for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            ps.append(getattr(module.gru, f"{k}_l{l}"))
    ps += [module.fc.weight, module.fc.bias]
    return ps


class _GruFn(torch.autograd.Function):
    # Every forward owns its saved activations (a workspace tensor kept in the autograd ctx) and pins the flat weights
    # and dims it ran with, so any number of graphs may be alive at once: loss(model(a)) + loss(model(b)), gradient
    # accumulation, a second model or an evaluation forward in between (the engine's own scratch and its loaded
    # weights are shared per device and would be overwritten).
    @staticmethod
    def forward(ctx, x, module, *params):
        module._bind_engine(x.device)
        eng = module._engine
        xd = x.detach().contiguous()
        with eng.lock:                                     # load + forward as one unit: the context holds ONE model
            module._sync_weights(x.device)
            out, ws, flat, dims = eng.gru_forward_train_ws(xd)
        ctx.module, ctx.x_req, ctx.dims = module, x.requires_grad, dims
        ctx.save_for_backward(xd, out, ws, flat)
        return out

    @staticmethod
    def backward(ctx, dout):
        xd, out, ws, flat = ctx.saved_tensors
        module = ctx.module
        eng = module._engine
        res = eng.gru_backward_ws(ctx.dims, flat, ws, xd, out, dout.contiguous(), want_dx=ctx.x_req)
        flat, dx = res if ctx.x_req else (res, None)
        grads, off = [], 0
        for p in _flat_order_params(module):
            n = p.numel()
            grads.append(flat[off:off + n].view_as(p))
            off += n
        return (dx, None, *grads)


def gru_forward_autograd(module, x):
    return _GruFn.apply(x, module, *_flat_order_params(module))


def shard_range(n, rank, world):
    """Contiguous split of n independent units (trajectories / samples): rank r gets [lo, hi)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatBucket:
    """All parameters (and their gradients) of a module as views of ONE flat fp32 tensor each."""

    def __init__(self, params, device=None):
        params = list(params)
        device = device or params[0].device
        n = sum(p.numel() for p in params)
        self.n = n
        self.w = torch.empty(n, dtype=torch.float32, device=device)
        # one extra slot behind the gradients: the rank's sample count rides in the same all-reduce (allreduce_weighted_)
        self._gbuf = torch.zeros(n + 1, dtype=torch.float32, device=device)
        self.g = self._gbuf[:n]
        off = 0
        for p in params:
            k = p.numel()
            self.w[off:off + k].copy_(p.detach().reshape(-1))
            p.data = self.w[off:off + k].view_as(p)           # state_dict()/checkpoints keep working
            p.grad = self.g[off:off + k].view_as(p)
            off += k
        self.params = params

    def allreduce_mean_(self, group=None):
        """SUM all-reduce of the single gradient bucket, then divide by the world size."""
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
            if world > 1:
                all_reduce_(self.g, group=group)
                self.g.div_(world)
        return self.g

    def allreduce_weighted_(self, n_local, group=None, force=False):
        """g holds the gradient of this rank's LOCAL mean loss over n_local samples (what the loss kernel produces).  The
        gradient of the GLOBAL mean is sum_r n_r g_r / sum_r n_r: equal to the plain mean over ranks only when the shards are
        equal, so ragged last shards (a dataset that does not divide by the world size, gru_train.py:216 random_split) are
        weighted by their size.  ONE collective: the count travels in the bucket's extra slot."""
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(group)
            if world > 1 or force:
                self.g.mul_(float(n_local))
                self._gbuf[self.n] = float(n_local)
                all_reduce_(self._gbuf, group=group)
                self.g.div_(self._gbuf[self.n])
        return self.g


class DataParallelTrainer:
    """One optimisation step = forward, device-side target + MSE, backward, bucket all-reduce, fused Adam."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, group=None, split_allreduce=True, force_distributed=False,
                 proto=None, algo=None):
        from .engine import default_engine
        # proto / algo: the RCCL protocol / algorithm this trainer is meant to run under (rccl_env).  They take effect only
        # when set before the process group was created, so a mismatch with the environment is an error, not a silent no-op.
        for name, want in (("NCCL_PROTO", proto), ("NCCL_ALGO", algo)):
            if want not in (None, "default") and os.environ.get(name) != want:
                raise RuntimeError(f"DataParallelTrainer: {name}={want} requested but the process runs with "
                                   f"{os.environ.get(name)!r}; call configure_rccl() before init_process_group")
        self.rccl = {k: os.environ.get(k, "default") for k in ("NCCL_PROTO", "NCCL_ALGO")}
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("DataParallelTrainer needs the model on the MI355X (no CPU fallback)")
        self.model, self.group = model, group
        self.eng = default_engine(dev.index or 0)
        self.bucket = FlatBucket(_flat_order_params(model), dev)
        self.m = torch.zeros_like(self.bucket.w)
        self.v = torch.zeros_like(self.bucket.w)
        self.lr, self.betas, self.eps, self.t = lr, betas, eps, 0
        self.lost_steps = 0                                  # steps redone with a launch per layer after a lost producer (step())
        # force_distributed: run the collectives on a one-rank group too (tests and bench.py --force-dist price the machinery)
        self.force = bool(force_distributed) or split_allreduce == "force"
        self.distributed = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.force)
        if self.distributed and dist.get_world_size(group) > 1:
            broadcast_(self.bucket.w, src=0, group=group)       # identical replicas
        # Two halves of the one bucket: the backward runs top layer first, so the gradients of layers L/2 .. L-1 and of the
        # head -- the TAIL of the flat vector, the count slot included -- are final while layers L/2-1 .. 0 are still being
        # swept.  Their all-reduce starts on a side stream at that point (os_gru_backward_mark) and travels underneath the
        # rest of the backward; the head of the vector follows on the main stream.  Latency-bound at 1.69 MB either way: what
        # this buys is the first half's launch + ring latency.
        self.split = None
        L = model.num_layers
        if self.distributed and split_allreduce and L > 1:
            per_layer = [sum(getattr(model.gru, f"{k}_l{l}").numel() for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for l in range(L)]
            self.split = dict(layer=L // 2, off=sum(per_layer[:L // 2]), side=torch.cuda.Stream(device=dev),
                              ev_top=torch.cuda.Event(), ev_done=torch.cuda.Event())
            self.split["ev_top"].record(); self.split["ev_done"].record()       # materialise the hipEvent handles

    def _allreduce_split(self, n_local):
        """allreduce_weighted_ in two collectives: the tail [off:] (top layers + head + count) on the side stream as soon as
        its gradients exist, the head [:off] on the main stream behind the whole backward."""
        sp, b = self.split, self.bucket
        main = torch.cuda.current_stream(b.g.device)
        top, low = b._gbuf[sp["off"]:], b._gbuf[:sp["off"]]
        sp["side"].wait_event(sp["ev_top"])                      # recorded by the library behind layer L/2's dW kernel
        with torch.cuda.stream(sp["side"]):
            top[:-1].mul_(n_local); top[-1:].fill_(n_local)
            all_reduce_(top, group=self.group)
            sp["ev_done"].record(sp["side"])
        low.mul_(n_local)
        all_reduce_(low, group=self.group)
        main.wait_event(sp["ev_done"])
        b.g.div_(b._gbuf[b.n])

    def step(self, x, y):
        e = self.eng
        with e.lock:                                       # the whole step: the context's loaded weights and scratch are this trainer's
            # Small batches run their layers as progress-counter launches.  The trainer keeps them asynchronous inside the step
            # (os_gru_set_stack 2: no stream synchronise per launch) and verifies ONCE, between the backward and the all-reduce
            # (stack_check: waits only if a stacked launch went out at all).  A lost producer means this rank's forward / backward
            # wrote NaN: the step's forward, loss and backward are redone with a launch per layer BEFORE anything reaches the
            # collective or the optimiser, so no rank ever contributes a poisoned gradient and the replicas cannot diverge
            # (the per-call retry of mode 1 is wrong here, and a local-only Adam skip is wrong with world > 1).
            prev = e._stack_mode                           # the mode in force (os_create's OS_GRU_STACK or the last set_stack_mode)
            if prev == 1:
                e.set_stack_mode(2)
            try:
                return self._step(x, y)
            finally:
                if e._stack_mode != prev:
                    e.set_stack_mode(prev)                 # exactly what was there: a user's OS_GRU_STACK=0 / 2 stays

    def _fwd_bwd(self, x, y):
        m, e = self.model, self.eng
        out = e.gru_forward_train(x)
        loss, dout, _ = e.gru_loss(out, y)
        if self.split is not None:
            e.gru_backward_mark(self.split["layer"], self.split["ev_top"])      # per call: the context is shared with other users
        e.gru_backward(x, out, dout, grad_flat=self.bucket.g)
        if self.split is not None:
            e.gru_backward_mark(0, None)
        return loss

    def _step(self, x, y):
        m, e = self.model, self.eng
        e.load_gru(self.bucket.w, m.input_size, m.hidden_size, m.num_layers, m.num_classes, m.use_sigmoid, owner=self)
        loss = self._fwd_bwd(x, y)
        if e.stack_check():
            # this step's stacked launches lost a producer: nothing has been applied or sent yet -- redo it with a launch per layer
            mode = e._stack_mode
            e.set_stack_mode(0)
            try:
                loss = self._fwd_bwd(x, y)
            finally:
                e.set_stack_mode(mode)
            self.lost_steps += 1
            e.stack_fallbacks += 1
        if self.split is None:
            self.bucket.allreduce_weighted_(x.shape[0], self.group, force=self.force)  # equal shards: the plain mean; ragged: weighted by size
        else:
            self._allreduce_split(float(x.shape[0]))
        self.t += 1
        e.adam_step(self.bucket.w, self.bucket.g, self.m, self.v, self.lr, self.betas[0], self.betas[1], self.eps, self.t)
        # the fused Adam wrote the flat bucket in place: the packed copy in the engine is stale now, and torch's version
        # counters did not move -- make the next model(x) (evaluation between steps, gru_train.py:253-261) load again
        e.invalidate_gru()
        return loss

    def replica_divergence(self):
        """max over ranks of |w_r - w_0| (0.0 = the replicas are still identical)."""
        return replica_divergence(self.bucket.w, self.group)
