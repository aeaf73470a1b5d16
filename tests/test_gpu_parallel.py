"""GPU tests of the data-parallel step on the flat adapter buffers (moka_amd/parallel.py):
``moka_adamw_flat`` against ``torch.optim.AdamW`` (fp32, same hyper-parameters), its gradient averaging /
zeroing / bf16 working copy, and the argument validation of the entry point.

Tolerance: the kernel evaluates the torch formula in fp32 with fused multiply-adds, torch's fused kernel in a
different operation order -> <= 2e-6 relative on the parameters after several steps (written here on purpose).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [4 * 1000 + 3, 1 << 20])
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adamw_matches_torch_adamw(n, wd):
    from moka_amd.parallel import FlatAdamW
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(7)
    p0 = torch.randn(n, generator=g)
    master = p0.to(dev).clone()
    grad = torch.zeros(n, device=dev)
    work = torch.empty(n, device=dev, dtype=torch.bfloat16)
    opt = FlatAdamW(master, grad, work, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=wd)
    ref_p = torch.nn.Parameter(p0.to(dev).clone())
    ref = torch.optim.AdamW([ref_p], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=wd)
    world = 4
    for step in range(5):
        gsum = torch.randn(n, generator=g).to(dev) * (0.1 + step)        # what an all-reduce (sum) over `world` ranks leaves
        grad.copy_(gsum)
        ref_p.grad = gsum / world
        opt.step(grad_scale=1.0 / world, zero_grad=True)
        ref.step()
        assert float(grad.abs().max()) == 0.0, "the gradient buffer must come back zeroed"
        err = ((master - ref_p.data).norm() / ref_p.data.norm()).item()
        assert err <= 2e-6, (step, err)
        assert torch.equal(work, master.to(torch.bfloat16)), "bf16 working copy = RNE(master)"
    st = ref.state[ref_p]
    assert ((opt.exp_avg - st["exp_avg"]).norm() / st["exp_avg"].norm()).item() <= 2e-6
    assert ((opt.exp_avg_sq - st["exp_avg_sq"]).norm() / st["exp_avg_sq"].norm()).item() <= 2e-6


def test_flat_adamw_without_working_copy_and_without_zeroing():
    from moka_amd.parallel import FlatAdamW
    dev = _dev()
    master = torch.ones(1024, device=dev)
    grad = torch.full((1024,), 0.5, device=dev)
    opt = FlatAdamW(master, grad, None, lr=1e-2, weight_decay=0.0)
    opt.step(zero_grad=False)
    assert float(grad.min()) == 0.5
    # first Adam step with bias correction: p -= lr * g / (|g| + eps)
    assert torch.allclose(master, torch.full_like(master, 1.0 - 1e-2), atol=1e-6)


def test_flat_adamw_rejects_bad_arguments():
    from moka_amd import _lib
    from moka_amd.parallel import FlatAdamW
    dev = _dev()
    lib = _lib.load()
    a = torch.zeros(64, device=dev)
    sp = torch.cuda.current_stream(dev).cuda_stream
    assert lib.moka_adamw_flat(a.data_ptr(), None, a.data_ptr(), a.data_ptr(), a.data_ptr(), 64, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, 1.0, 0, sp) == -1
    assert b"step" in lib.moka_last_error()
    assert lib.moka_adamw_flat(None, None, a.data_ptr(), a.data_ptr(), a.data_ptr(), 64, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, 0, sp) == -1
    assert lib.moka_adamw_flat(a.data_ptr(), None, a.data_ptr(), a.data_ptr(), a.data_ptr(), 64, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, 1.0, 0, sp) == -1
    with pytest.raises(TypeError):
        FlatAdamW(a, a.to(torch.bfloat16))
    with pytest.raises(_lib.MokaError):
        FlatAdamW(torch.zeros(8), torch.zeros(8)).step()


def test_flat_adamw_in_slices_with_device_coefficients_equals_the_one_launch_step():
    """``begin_step(); step_range(lo, hi)`` per gradient bucket (moka_adamw_flat_dev: the step-dependent coefficients are read from device
    memory, written there by the one-thread launch of begin_step from its launch arguments) = ``step()`` bit for bit -- also when the
    slice launches are captured ONCE in a hipGraph and replayed behind a live begin_step, and (to the last bit of the device's pow) when
    begin_step itself is captured and counts the steps on the device."""
    from moka_amd.parallel import FlatAdamW
    dev = _dev()
    n = 4096 * 5 + 64
    g = torch.Generator(device="cpu").manual_seed(11)
    p0 = torch.randn(n, generator=g)
    grads = [(torch.randn(n, generator=g) * (0.1 + k)).to(dev) for k in range(4)]

    def fresh():
        master, grad = p0.to(dev).clone(), torch.zeros(n, device=dev)
        work = torch.empty(n, device=dev, dtype=torch.bfloat16)
        return master, grad, work, FlatAdamW(master, grad, work, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01)

    m0, g0, w0, ref = fresh()
    for gk in grads:
        g0.copy_(gk)
        ref.step(grad_scale=0.25, zero_grad=True)
    cuts = [0, 4096, 4096 * 3, n]
    # live slices
    m1, g1, w1, opt = fresh()
    for gk in grads:
        g1.copy_(gk)
        opt.begin_step()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            opt.step_range(lo, hi, grad_scale=0.25, zero_grad=True)
    assert torch.equal(m1, m0) and torch.equal(w1, w0) and torch.equal(opt.exp_avg_sq, ref.exp_avg_sq) and float(g1.abs().max()) == 0.0
    # the same launches captured once
    m2, g2, w2, opt2 = fresh()
    opt2.set_device_step(0)                                       # (allocates the coefficient state without counting a step)
    graph, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
    with torch.cuda.graph(graph, stream=side):
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            opt2.step_range(lo, hi, grad_scale=0.25, zero_grad=True)
    for gk in grads:
        g2.copy_(gk)
        opt2.begin_step()
        graph.replay()
    torch.cuda.synchronize()
    assert opt2.t == ref.t and torch.equal(m2, m0) and torch.equal(w2, w0) and torch.equal(opt2.exp_avg, ref.exp_avg)
    # begin_step captured too: the device counts the steps, the host enqueues NOTHING but replays (and may run any number of steps ahead)
    m3, g3s, w3, opt3 = fresh()
    opt3.set_device_step(0)
    gsrc = torch.zeros_like(g3s)
    graph3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph3, stream=side):
        g3s.copy_(gsrc)
        opt3.begin_step(device_counter=True)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            opt3.step_range(lo, hi, grad_scale=0.25, zero_grad=True)
    opt3.t -= 1
    for gk in grads:
        gsrc.copy_(gk)
        graph3.replay()
        opt3.t += 1
    torch.cuda.synchronize()
    assert int(opt3._state[3:4].view(torch.int32).item()) == ref.t == opt3.t
    assert (m3 - m0).abs().max().item() <= 1e-6 * m0.abs().max().item() and (opt3.exp_avg - ref.exp_avg).abs().max().item() == 0.0
    with pytest.raises(ValueError):
        opt.step_range(2, 64)


def test_no_decay_ranges_follow_the_hf_rule_and_equal_torch_adamw_with_two_groups():
    """attach(no_decay="hf"): biases and norm weights take no weight decay (what HF Trainer's default optimizer does and the reference
    therefore trains with); the fused step in segments == torch.optim.AdamW with a decay and a no-decay group."""
    from moka_amd.parallel import FlatAdamW
    dev = _dev()
    n = 1024 * 3
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    master, grad = p0.to(dev).clone(), torch.zeros(n, device=dev)
    opt = FlatAdamW(master, grad, None, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.1)
    opt.no_decay_ranges = [(1024, 1536), (2048, 2052)]
    ref = torch.nn.Parameter(p0.clone().to(dev))
    mask = torch.zeros(n, dtype=torch.bool)
    mask[1024:1536] = True
    mask[2048:2052] = True
    # torch.optim.AdamW on one flat parameter cannot mix decays: emulate with two parameters
    pa, pb = torch.nn.Parameter(p0[~mask].clone().to(dev)), torch.nn.Parameter(p0[mask].clone().to(dev))
    topt = torch.optim.AdamW([{"params": [pa], "weight_decay": 0.1}, {"params": [pb], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    for k, gk in enumerate(grads):
        grad.copy_(gk.to(dev))
        if k == 1:
            opt.begin_step()
            opt.step_range(0, 2048, zero_grad=True)          # (slices cut at arbitrary multiples of 4: the segments follow)
            opt.step_range(2048, n, zero_grad=True)
        else:
            opt.step(zero_grad=True)
        pa.grad, pb.grad = gk[~mask].to(dev), gk[mask].to(dev)
        topt.step()
    torch.cuda.synchronize()
    got = master.cpu()
    assert (got[~mask] - pa.detach().cpu()).abs().max().item() <= 2e-6
    assert (got[mask] - pb.detach().cpu()).abs().max().item() <= 2e-6
    del ref
