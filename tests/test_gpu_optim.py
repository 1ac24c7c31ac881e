"""Optimizer tail: multi-tensor grad norm + clip + AdamW against torch's own
clip_grad_norm_ / torch.optim.AdamW on the same device tensors, and the multi-tensor weight packing
against the single-tensor entry point (bit-exact).  Everything goes through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZES = [(1,), (7,), (3, 5), (16384,), (16385,), (64, 33, 3, 3), (100003,), (2, 2), (40000,)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def make_params(dev, n_extra=0, seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = SIZES + [(17,)] * n_extra
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in shapes]


def set_grads(ps, scale, seed):
    g = torch.Generator().manual_seed(seed)
    for p in ps:
        p.grad = (torch.randn(p.shape, generator=g) * scale).to(p.device)


@pytest.mark.parametrize("n_extra", [0, 45])           # 45 extra tensors: more than one 40-tensor launch
@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adamw_matches_torch(dev, n_extra, wd):
    from unet_amd.optim import FusedAdamW
    a, b = make_params(dev, n_extra), make_params(dev, n_extra)
    oa = FusedAdamW(a, lr=3e-3, weight_decay=wd, betas=(0.9, 0.99))
    ob = torch.optim.AdamW(b, lr=3e-3, weight_decay=wd, betas=(0.9, 0.99), foreach=False)
    for it in range(4):
        set_grads(a, 0.3, 10 + it); set_grads(b, 0.3, 10 + it)
        if it == 2:
            a[1].grad = None; b[1].grad = None          # a parameter without a gradient is skipped
        v0 = a[0]._version
        oa.step(); ob.step()
        assert a[0]._version > v0                        # packed-weight caches key on the version
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa["step"]) == float(sb["step"])
        torch.testing.assert_close(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=1e-8)
        torch.testing.assert_close(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-10)
    # state_dict layout is torch's: a torch AdamW loads it and continues identically
    oc = torch.optim.AdamW(a, lr=3e-3, weight_decay=wd, betas=(0.9, 0.99))
    oc.load_state_dict(oa.state_dict())
    assert float(oc.state[a[0]]["step"]) == 4.0


@pytest.mark.parametrize("scale", [5.0, 1e-4])        # norm well above / well below max_norm = 1
def test_clip_and_step_matches_torch(dev, scale):
    from unet_amd.optim import FusedAdamW
    a, b = make_params(dev), make_params(dev)
    extra_a = torch.nn.Parameter(torch.ones(33, device=dev)); extra_b = torch.nn.Parameter(torch.ones(33, device=dev))
    oa = FusedAdamW(a, lr=1e-3, weight_decay=1e-4)
    ob = torch.optim.AdamW(b, lr=1e-3, weight_decay=1e-4, foreach=False)
    for it in range(2):
        set_grads(a + [extra_a], scale, 20 + it); set_grads(b + [extra_b], scale, 20 + it)
        na = oa.clip_and_step(a + [extra_a], 1.0)      # extra: clipped with the rest, not stepped
        nb = torch.nn.utils.clip_grad_norm_(b + [extra_b], 1.0)
        ob.step()
        torch.testing.assert_close(na, nb, rtol=2e-6, atol=0)
        for pa, pb in zip(a + [extra_a], b + [extra_b]):
            torch.testing.assert_close(pa.grad, pb.grad, rtol=3e-6, atol=1e-12)     # clipped in place, as torch does
    for pa, pb in zip(a, b):
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)
    assert torch.equal(extra_a, extra_b)


def test_fallback_paths_are_torch(dev):
    """amsgrad (not covered by the kernel) must take torch's own step, not silently differ."""
    from unet_amd.optim import FusedAdamW
    a, b = make_params(dev), make_params(dev)
    oa = FusedAdamW(a, lr=1e-3, amsgrad=True)
    ob = torch.optim.AdamW(b, lr=1e-3, amsgrad=True)
    set_grads(a, 1.0, 5); set_grads(b, 1.0, 5)
    oa.clip_and_step(a, 1.0)
    torch.nn.utils.clip_grad_norm_(b, 1.0); ob.step()
    for pa, pb in zip(a, b):
        assert torch.equal(pa, pb)
    cpu = [torch.nn.Parameter(torch.ones(4))]
    oc = FusedAdamW(cpu, lr=1e-3)
    cpu[0].grad = torch.ones(4)
    oc.step()                                            # CPU tensors: torch's path
    assert float(cpu[0][0]) < 1.0


def test_pack_weights_multi_bit_exact(dev):
    from unet_amd import ops
    g = torch.Generator().manual_seed(1)
    items = []
    for dt in (ops.BF16, ops.F32):
        ws = [(torch.randn(24, 16, 3, 3, generator=g).to(dev), 0), (torch.randn(24, 16, 3, 3, generator=g).to(dev), 1),
              (torch.randn(64, 40, 3, 3, generator=g).to(dev), 1), (torch.randn(16, 8, 2, 2, generator=g).to(dev), 2),
              (torch.randn(16, 8, 2, 2, generator=g).to(dev), 3), (torch.randn(8, 8, 1, 3, generator=g).to(dev), 0)]
        ws += [(torch.randn(8, 8, 3, 3, generator=g).to(dev), i & 1) for i in range(44)]
        # forward + dgrad image of ONE source (merged into one tile pass), ragged tiles, Cin=1, 5x5 / 1x5 / 7x7 taps
        for shape in [(72, 40, 3, 3), (64, 1, 3, 3), (33, 65, 3, 3), (24, 16, 5, 5), (16, 24, 1, 5), (8, 16, 7, 7), (128, 256, 3, 3)]:
            w = torch.randn(*shape, generator=g).to(dev)
            ws += [(w, 0), (w, 1)]
        outs = [torch.empty(w.numel(), dtype=ops.torch_dtype(dt), device=dev) for w, _ in ws]
        ops.pack_weights_multi([(w, mode, o) for (w, mode), o in zip(ws, outs)], dt)
        for (w, mode), o in zip(ws, outs):
            assert torch.equal(o, ops.pack_weights(w, mode, dt))


def test_engine_prepack_tracks_parameter_updates(dev):
    """After an optimizer step the packed images must be refreshed (one multi-tensor launch) and the next
    forward must see the new weights."""
    from unet_amd.models import REGISTRY
    from unet_amd.optim import FusedAdamW
    torch.manual_seed(0)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16]).to(dev).train()
    x = torch.rand(2, 1, 32, 32, device=dev)
    opt = FusedAdamW(m.parameters(), lr=1e-2)
    m(x).float().square().mean().backward()
    opt.step()
    y1 = m(x).detach().clone()
    fresh = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16]).to(dev).train()
    fresh.load_state_dict(m.state_dict())
    # same parameters, cold caches: identical forward (BN running stats do not enter a train-mode forward)
    y2 = fresh(x).detach()
    assert torch.equal(y1, y2)
