"""GPU parity of the training step (gru/gru_train.py:232-249): HIP forward/backward vs torch autograd on the
reference's nn.GRU math (CPU, float32 -> compared in float64 tolerance terms), and the G6 golden Adam step."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def torch_reference_grads(sd, dims, x, y):
    """The reference's loop body on CPU with torch's own GRU (this is the checker, not the product)."""
    I, H, L, C = dims
    gru = torch.nn.GRU(I, H, L, batch_first=True).double()
    fc = torch.nn.Linear(H, C).double()
    gru.load_state_dict({k[4:]: v.double() for k, v in sd.items() if k.startswith("gru.")})
    fc.load_state_dict({k[3:]: v.double() for k, v in sd.items() if k.startswith("fc.")})
    xx = x.double().requires_grad_(True)
    o, _ = gru(xx)
    out = torch.sigmoid(fc(o[:, -1, :]))
    tgt = torch.cat([y.double(), (out[:, :C // 2].detach() - y.double()).abs()], dim=1)
    loss = torch.nn.functional.mse_loss(out, tgt)
    loss.backward()
    grads = {f"gru.{k}": p.grad for k, p in gru.named_parameters()}
    grads.update({f"fc.{k}": p.grad for k, p in fc.named_parameters()})
    return out.detach(), tgt, loss.item(), grads, xx.grad


@pytest.mark.parametrize("dims,B,T", [((60, 64, 1, 24), 50, 10), ((60, 64, 2, 24), 130, 7), ((188, 128, 4, 24), 70, 10),
                                      # the reference's own training batch (gru/gru_train.py:32-36): forward = one gru_stack_kernel launch
                                      ((188, 128, 4, 24), 64, 10),
                                      ((61, 32, 2, 6), 33, 5),
                                      # whole 32-row tiles: the fused W_ih / W_hh gradient kernel (dw3_kernel<2,4>, <4,4>, <2,2>),
                                      # its batch_first layer-0 input, and T = 1 (falls back to the two-launch form)
                                      ((60, 128, 2, 24), 256, 6), ((60, 64, 2, 24), 96, 5), ((60, 128, 1, 24), 64, 1),
                                      # the 188-wide first layer in whole tiles: the ten-accumulator form dw3_kernel<6,4,16>
                                      ((188, 128, 2, 24), 64, 3),
                                      # large ragged batches: 64-row layer kernel with activation saves, many dW slices
                                      ((60, 128, 2, 24), 20013, 4), ((60, 64, 2, 24), 40001, 3)])
def test_backward_matches_torch_autograd(dims, B, T):
    from optistate_amd import RNN
    I, H, L, C = dims
    torch.manual_seed(2)
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda")
    x = torch.rand(B, T, I); y = torch.rand(B, C // 2)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref_out, ref_tgt, ref_loss, ref_g, ref_dx = torch_reference_grads(sd, dims, x, y)

    xg = x.cuda().requires_grad_(True)
    out = m(xg)                                           # HIP forward (training path: grad enabled)
    assert np.abs(out.detach().cpu().numpy() - ref_out.numpy()).max() < 1e-5
    if 2 <= L <= 8 and ((B + 31) // 32) * L <= 256:       # small batch: the layers of the training forward run as one pipelined launch
        assert m._engine.kernel_name("gru_layer") in ("gru_stack_kernel", "gru_wide_kernel"), m._engine.kernel_name("gru_layer")
    tgt = torch.cat([y.cuda(), (out[:, :C // 2].detach() - y.cuda()).abs()], dim=1)
    loss = torch.nn.functional.mse_loss(out, tgt)         # the reference's criterion; torch only does bookkeeping
    loss.backward()                                       # HIP backward
    assert abs(loss.item() - ref_loss) < 1e-6
    for k, p in m.named_parameters():
        g, r = p.grad.cpu().double(), ref_g[k]
        scale = max(r.abs().max().item(), 1e-8)
        assert (g - r).abs().max().item() < 2e-4 * scale + 1e-9, (k, (g - r).abs().max().item(), scale)
    scale = ref_dx.abs().max().item()
    assert (xg.grad.cpu().double() - ref_dx).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("terms,bar", [(3, 2e-4), (2, 2e-3)])
@pytest.mark.parametrize("dims,B,T", [((60, 128, 2, 24), 256, 6), ((60, 64, 2, 24), 96, 5), ((188, 128, 2, 24), 64, 3), ((188, 128, 4, 24), 2048, 10)])
def test_opt_in_split_bf16_weight_gradients_match_float64_autograd(dims, B, T, terms, bar):
    """OS_GRU_SPLIT_TRAIN (opt-in, never the default): the weight-gradient products on dw3_bf16_kernel -- every fp32 operand split into
    three (two) bf16 terms, fp32 accumulation.  Three terms keep the fp32 path's bar against float64 autograd (2e-4 of the largest entry);
    two terms (16 mantissa bits per operand) are held to 2e-3.  The default path is untouched (the engine is switched back)."""
    from optistate_amd import RNN
    I, H, L, C = dims
    torch.manual_seed(2)
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda")
    x = torch.rand(B, T, I); y = torch.rand(B, C // 2)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref_out, ref_tgt, ref_loss, ref_g, ref_dx = torch_reference_grads(sd, dims, x, y)
    xg = x.cuda().requires_grad_(True)
    out = m(xg)
    eng = m._engine
    eng.set_gru_split_bf16(terms, train=True)
    try:
        tgt = torch.cat([y.cuda(), (out[:, :C // 2].detach() - y.cuda()).abs()], dim=1)
        torch.nn.functional.mse_loss(out, tgt).backward()
        assert eng.kernel_name("train_dw") == f"dw3_bf16_kernel<{terms}>", eng.kernel_name("train_dw")
    finally:
        eng.set_gru_split_bf16(0)
    for k, p in m.named_parameters():
        g, r = p.grad.cpu().double(), ref_g[k]
        scale = max(r.abs().max().item(), 1e-8)
        assert (g - r).abs().max().item() < bar * scale + 1e-9, (k, (g - r).abs().max().item(), scale)


def test_g6_adam_step_matches_reference_loop():
    """One optimisation step with the reference's loop (gru_train.py:232-249) and torch.optim.Adam on the drop-in RNN."""
    from optistate_amd import RNN
    g = load_golden("gru_g6_train.npz")
    I, H, L, C = [int(v) for v in g["dims"]]
    m = RNN(I, H, L, C, torch.device("cuda"))
    m.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w0:")})
    m = m.to("cuda")
    opt = torch.optim.Adam(m.parameters(), lr=0.0001)
    inputs, labels = torch.as_tensor(g["inputs"]).cuda(), torch.as_tensor(g["labels"]).cuda()
    outputs = m(inputs)
    assert np.abs(outputs.detach().cpu().numpy() - g["outputs"]).max() < 1e-5
    err = (outputs[:, :12].detach() - labels).abs()
    target = torch.cat([labels, err], dim=1)
    assert np.abs(target.cpu().numpy() - g["target"]).max() < 1e-5
    loss = torch.nn.MSELoss()(outputs, target)
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    opt.zero_grad(); loss.backward(); opt.step()
    for k, p in m.named_parameters():
        gr = g["g:" + k]
        assert np.abs(p.grad.cpu().numpy() - gr).max() < 2e-4 * max(np.abs(gr).max(), 1e-8) + 1e-9, k
        # Adam's first step moves every weight by ~lr * sign(g): compare the post-step weights
        assert np.abs(p.detach().cpu().numpy() - g["w1:" + k]).max() < 2e-5, k


def test_device_side_loss_and_fused_adam_trainer():
    """DataParallelTrainer (world size 1): device-side target/MSE + flat bucket + fused Adam == the reference loop."""
    from optistate_amd import RNN
    from optistate_amd.train import DataParallelTrainer
    g = load_golden("gru_g6_train.npz")
    I, H, L, C = [int(v) for v in g["dims"]]
    m = RNN(I, H, L, C, torch.device("cuda"))
    m.load_state_dict({k[3:]: torch.as_tensor(g[k]) for k in g.files if k.startswith("w0:")})
    m = m.to("cuda")
    tr = DataParallelTrainer(m, lr=0.0001)
    loss = tr.step(torch.as_tensor(g["inputs"]).cuda(), torch.as_tensor(g["labels"]).cuda())
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) < 1e-6
    sd = m.state_dict()                                   # parameters are views of the flat bucket
    for k in sd:
        assert np.abs(sd[k].cpu().numpy() - g["w1:" + k]).max() < 2e-5, k
    # a second step must keep working (activations re-saved, weights re-packed)
    loss2 = tr.step(torch.as_tensor(g["inputs"]).cuda(), torch.as_tensor(g["labels"]).cuda())
    assert loss2.item() < loss.item() + 1e-3


def test_loss_is_order_independent_and_backward_clears_its_gradient_vector():
    """os_gru_loss: the last workgroup adds the per-workgroup partial sums in index order and WRITES the loss (no pre-zeroed
    accumulator, no atomics on the result): repeated calls give bit-identical values, whatever the loss tensor held before.
    os_gru_backward: the flat gradient vector is cleared by the launch itself (its first kernel), so stale contents do not leak."""
    from optistate_amd import RNN, flatten_state_dict
    from optistate_amd.engine import default_engine
    torch.manual_seed(5)
    I, H, L, C, B, T = 60, 64, 2, 24, 4096, 4
    m = RNN(I, H, L, C, torch.device("cuda")).to("cuda")
    eng = default_engine(0)
    eng.load_gru(flatten_state_dict(m.state_dict(), L, "cuda"), I, H, L, C)
    x = torch.rand(B, T, I, device="cuda"); y = torch.rand(B, C // 2, device="cuda")
    out = eng.gru_forward_train(x)
    losses = []
    for _ in range(5):
        loss, dout, tgt = eng.gru_loss(out, y, want_target=True)
        losses.append(loss.cpu().numpy().view(np.uint32)[0])
    assert len(set(losses)) == 1, losses
    ref = torch.nn.functional.mse_loss(out.double(), tgt.double()).item()
    assert abs(float(np.array(losses[0], dtype=np.uint32).view(np.float32)) - ref) < 1e-6
    n = sum(p.numel() for p in m.parameters())
    g_clean = torch.zeros(n, device="cuda")
    eng.gru_backward(x, out, dout, grad_flat=g_clean)
    g_dirty = torch.full((n,), 1.0e6, device="cuda")
    out2 = eng.gru_forward_train(x)
    eng.gru_backward(x, out2, dout, grad_flat=g_dirty)
    scale = g_clean.abs().max().item()
    assert scale > 0 and (g_clean - g_dirty).abs().max().item() < 1e-4 * scale      # float atomics: summation order only


def test_split_allreduce_on_a_side_stream_gives_the_same_step():
    """DataParallelTrainer with the gradient bucket reduced in two halves (top layers + head on a side stream behind their dW
    kernel: os_gru_backward_mark; the rest behind the backward) against the plain single-process step: same loss, same
    reduced gradient (to the reduction-order noise of the dW kernels' atomics: Adam would turn a sign flip of a noise-level
    gradient into a visible weight difference, so the gradients are compared, after one step each).  One-rank RCCL group: the
    collectives really run, on the streams the N-rank job uses."""
    import socket
    import torch.distributed as dist
    from optistate_amd import RNN
    from optistate_amd.train import DataParallelTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dims, B, T = (188, 128, 4, 24), 512, 10
    x, y = torch.rand(B, T, dims[0]).cuda(), torch.rand(B, 12).cuda()

    def run(**kw):
        torch.manual_seed(7)
        m = RNN(*dims, torch.device("cuda")).to("cuda")
        tr = DataParallelTrainer(m, lr=1e-3, **kw)
        loss = float(tr.step(x, y).item())
        torch.cuda.synchronize()
        return loss, tr.bucket.g.clone(), tr
    l0, g0, _ = run(split_allreduce=False)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        l1, g1, tr = run(split_allreduce=True, force_distributed=True)
        assert tr.split is not None and tr.split["layer"] == 2 and tr.split["off"] == 3 * 128 * (188 + 128 + 2) + 3 * 128 * (128 + 128 + 2)
        l2, g2, tr2 = run(split_allreduce=False, force_distributed=True)
        assert tr2.split is None
    finally:
        dist.destroy_process_group()
    assert l0 == l1 == l2
    scale = g0.abs().max().item()
    assert (g1 - g0).abs().max().item() < 1e-5 * scale and (g2 - g0).abs().max().item() < 1e-5 * scale
    assert g0.abs().max().item() > 0


@pytest.mark.parametrize("wide", ["0", "1"], ids=["one-cu-per-tile", "four-cus-per-tile"])
@pytest.mark.parametrize("dims,B,T", [((188, 128, 4, 24), 64, 10), ((188, 128, 4, 24), 70, 10), ((60, 128, 2, 24), 1024, 6), ((128, 128, 8, 24), 33, 3),
                                      ((188, 128, 4, 24), 2048, 2), ((60, 128, 3, 24), 5, 1), ((188, 128, 4, 24), 512, 4), ((1, 128, 2, 24), 40, 12)])
def test_stacked_backward_sweep_matches_per_layer_sweeps(monkeypatch, dims, B, T, wide):
    """Small batches at H = 128 run every layer's backward sweep in ONE launch (bwd_sweep_stack_kernel: blockIdx.y = 0 is the top layer,
    layer l takes dx_{l+1}[t] through a progress counter as soon as it is published).  The workgroups run bwd_sweep_kernel<1,8,32>'s
    body, so the gate derivatives are the same numbers and the flat gradient may differ from a launch per layer only by the order of
    the dW kernels' atomic additions -- on every one of 10 repeats (a consumer that read a step early would show here); torch
    autograd (float64) as truth on the small shapes.  Shapes: the reference's training batch, a partial tile, 32 tiles x 2 layers, eight layers, the largest
    eligible batch (64 tiles x 4), T = 1, the largest batch of the four-CUs-per-tile form, input width 1.
    wide = "1" (the default): where the shape allows it (up to 512 windows at four layers) forward and sweep run as gru_wide_kernel /
    bwd_sweep_wide_kernel -- four workgroups per (layer, tile) that exchange their slices of h_t / dh_t every step; another order of the
    partial sums, so 1e-5 relative against the per-layer launches there."""
    from optistate_amd import Engine, RNN, flatten_state_dict
    I, H, L, C = dims
    torch.manual_seed(47)
    m = RNN(I, H, L, C, torch.device("cpu"))
    x = torch.rand(B, T, I); y = torch.rand(B, C // 2)
    flat = flatten_state_dict(m.state_dict(), L, "cuda")
    grads = {}
    monkeypatch.setenv("OS_GRU_WIDE", wide)
    tiles = (B + 31) // 32
    is_wide = wide == "1" and (tiles + 7) // 8 * 4 * L <= 32
    for stack in ("0", "1"):
        monkeypatch.setenv("OS_GRU_STACK", stack)
        e = Engine(0)
        e.load_gru(flat, I, H, L, C)
        for rep in range(10 if stack == "1" else 1):
            out = e.gru_forward_train(x.cuda())
            _, dout, _ = e.gru_loss(out, y.cuda())
            g = e.gru_backward(x.cuda(), out, dout).clone()
            torch.cuda.synchronize()
            want = "bwd_sweep_wide_kernel" if is_wide else "bwd_sweep_stack_kernel"
            assert (e.kernel_name("train_sweep") == want) == (stack == "1"), e.kernel_name("train_sweep")
            if stack == "1":
                scale = grads["0"].abs().max().item()
                bar = (1e-5 if is_wide else 2e-6) * scale + 1e-9
                assert torch.isfinite(g).all() and (g - grads["0"]).abs().max().item() < bar, (rep, (g - grads["0"]).abs().max().item(), scale)
        grads[stack] = g
    # against autograd, parameter by parameter in the flat order (the smallest shapes only: the float64 CPU reference takes most of a
    # minute per case on the GPU box's host, and the per-layer launches are held to it in test_backward_matches_torch_autograd)
    if B * T > 128:
        return
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    _, _, _, ref_g, _ = torch_reference_grads(sd, dims, x, y)
    order = [f"gru.{k}_l{l}" for l in range(L) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")] + ["fc.weight", "fc.bias"]
    off = 0
    gflat = grads["1"].cpu().double()
    for k in order:
        r = ref_g[k].reshape(-1)
        gk = gflat[off:off + r.numel()]
        scale = max(r.abs().max().item(), 1e-8)
        assert (gk - r).abs().max().item() < 2e-4 * scale + 1e-9, k
        off += r.numel()


@pytest.mark.parametrize("I,H,B,T", [(64, 128, 33, 25), (32, 128, 64, 3), (20, 64, 40, 5)])
def test_one_layer_model_with_fewer_input_chunks_than_hidden_chunks(I, H, B, T):
    """Round 5, found by tools/fuzz_shapes.py: the backward's transposed weight pack sized its grid by the INPUT chunks of the widest
    layer; a one-layer model with input_size <= 96 at hidden 128 (<= 32 at hidden 64) left W_hh^T's upper chunks unpacked and every
    GRU gradient was wrong by ~10 % (the reference's shapes -- four layers, or 60 inputs at hidden 64 -- were never affected).
    Gradients against fp64 autograd."""
    from optistate_amd import RNN
    torch.manual_seed(25)
    C = 24
    m = RNN(I, H, 1, C, torch.device("cuda")).to("cuda").eval()
    x = torch.rand(B, T, I) * 2 - 1
    with torch.no_grad():
        m(x[:1].cuda())
    eng = m._engine
    xg = x.cuda()
    y = torch.rand(B, C // 2, device="cuda")
    o = eng.gru_forward_train(xg)
    _, dout, _ = eng.gru_loss(o, y, want_target=True)
    g = eng.gru_backward(xg, o, dout).double().cpu()
    md = torch.nn.GRU(I, H, 1, batch_first=True).double()
    fc = torch.nn.Linear(H, C).double()
    sd = {k: v.double().cpu() for k, v in m.state_dict().items()}
    md.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("gru.")})
    fc.load_state_dict({k[3:]: v for k, v in sd.items() if k.startswith("fc.")})
    hseq, _ = md(x.double())
    od = torch.sigmoid(fc(hseq[:, -1]))
    od.backward(dout.double().cpu())
    ref = torch.cat([p.grad.reshape(-1) for p in list(md.parameters()) + list(fc.parameters())])
    assert (o.double().cpu() - od.detach()).abs().max().item() < 1e-5
    assert (g - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
