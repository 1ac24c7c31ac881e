"""Inputs whose size is not divisible by 2^levels: nn.MaxPool2d's floor rule at the odd levels and the reference's
F.interpolate(..., bilinear, align_corners=True) of the upsampled tensor onto its skip connection
(/root/reference/models/unet.py:56-57, models/reparm_unet.py:209-210).  Whole models against goldens captured from the real
reference on 2x1x36x44 (tests/golden/make_odd_size.py); the two kernels against torch on the CPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _model(key, kw, g, dev):
    from unet_amd.models import REGISTRY
    torch.manual_seed(42)
    m = REGISTRY[key](**kw)
    init = {k[5:]: v for k, v in g.items() if k.startswith("init/")}
    assert list(m.state_dict()) == list(init)
    for k, v in m.state_dict().items():
        assert torch.equal(v, init[k]), k
    return m.to(dev).train()


@pytest.mark.parametrize("name,key,kw,tol", [
    ("unet_odd_36x44", "unet", dict(in_channels=1, n_classes=2, features=[8, 16, 32]), 1e-4),
    ("repunet_odd_36x44", "reparam_unet", dict(in_channels=1, n_classes=2, features=[8, 16, 32], kernel_sizes=[3]), 3e-4),
])
def test_odd_size_fp32_matches_reference(dev, name, key, kw, tol):
    from unet_amd.losses import get_loss
    g = load(name)
    m = _model(key, kw, g, dev)
    logits = m(g["x"].to(dev))
    loss = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(logits, g["y"].to(dev))
    loss.backward()
    assert tuple(logits.shape) == tuple(g["logits"].shape)
    assert relmax(logits.detach().cpu(), g["logits"]) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert torch.equal(logits.argmax(1).cpu(), g["logits"].argmax(1))
    gmax = max(float(v.abs().max()) for k, v in g.items() if k.startswith("grad/"))
    for k, p in m.named_parameters():
        ref = g["grad/" + k]
        assert p.grad is not None, k
        if key == "unet" and (k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")):
            assert float((p.grad.cpu() - ref).abs().max()) < 1e-5 * gmax, k      # noise-only gradient in front of train-mode BN
        else:
            assert relmax(p.grad.cpu(), ref) < tol, (k, relmax(p.grad.cpu(), ref))
    for k, b in m.named_buffers():
        ref = g["buf1/" + k]
        if ref.dtype == torch.int64:
            assert torch.equal(b.cpu(), ref), k
        else:
            assert relmax(b.cpu(), ref) < 1e-5, k


def test_odd_size_eval_and_bf16_run(dev):
    """Inference path (BatchNorm folded, no grad) and the bf16 mode on the same odd-sized input: eval logits against the
    oracle with the running statistics; bf16 logits within the stated bf16 bound (rel-L2 3e-2) of the fp32 ones."""
    from oracle import unet_ref
    g = load("unet_odd_36x44")
    m = _model("unet", dict(in_channels=1, n_classes=2, features=[8, 16, 32]), g, dev)
    x = g["x"].to(dev)
    with torch.no_grad():
        m(x)                                      # one train-mode forward moves the running statistics off their init
    m.eval()
    st = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        got = m(x)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got16 = m(x)
    ref = unet_ref.unet_forward(g["x"], st, train=False)
    assert relmax(got.cpu(), ref) < 1e-4
    assert float((got16.float().cpu() - ref).norm() / ref.norm()) < 3e-2


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("shape", [(2, 8, 10, 9, 11, 16), (1, 4, 4, 5, 5, 8), (2, 18, 22, 18, 23, 32), (1, 1, 6, 3, 7, 8)])
def test_resize_bilinear_fwd_bwd_vs_torch(dev, dt, shape):
    from unet_amd import ops
    N, h, w, H, W, C = shape
    tdt = torch.bfloat16 if dt == 1 else torch.float32
    g = torch.Generator().manual_seed(h * 100 + w)
    x = torch.randn(N, C, h, w, generator=g).to(tdt).float().requires_grad_(True)
    ref = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)
    dy = torch.randn(N, C, H, W, generator=g).to(tdt).float()
    ref.backward(dy)
    xa = ops.Act(x.detach().permute(0, 2, 3, 1).contiguous().to(tdt).to(dev), 0, C)
    buf = torch.zeros(N, H, W, C + 8, dtype=tdt, device=dev)
    ops.resize_bilinear(xa, ops.Act(buf, 8, C))
    got = buf[..., 8:].float().cpu().permute(0, 3, 1, 2)
    tol = 1e-2 if dt == 1 else 1e-5
    assert float((got - ref.detach()).abs().max()) <= tol * float(ref.detach().abs().max())
    assert float(buf[..., :8].float().abs().max()) == 0.0
    dya = ops.Act(dy.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev), 0, C)
    dxa = ops.dense(N, h, w, C, dt, dev)
    ops.resize_bilinear_bwd(dya, dxa)
    gdx = dxa.buf.float().cpu().permute(0, 3, 1, 2)
    assert float((gdx - x.grad).abs().max()) <= tol * float(x.grad.abs().max())


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("hw", [(9, 11), (8, 7), (5, 4), (3, 3)])
def test_maxpool_odd_sizes_floor_rule(dev, dt, hw):
    from unet_amd import ops
    H, W = hw
    N, C = 2, 16
    tdt = torch.bfloat16 if dt == 1 else torch.float32
    g = torch.Generator().manual_seed(H * 10 + W)
    x = torch.randn(N, C, H, W, generator=g).to(tdt).float().requires_grad_(True)
    ref, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    dy = torch.randn(ref.shape, generator=g).to(tdt).float()
    dsk = torch.randn(N, C, H, W, generator=g).to(tdt).float()
    ref.backward(dy)
    xa = ops.Act(x.detach().permute(0, 2, 3, 1).contiguous().to(tdt).to(dev), 0, C)
    y, code = ops.maxpool_fwd(xa)
    assert (y.H, y.W) == (H // 2, W // 2)
    assert torch.equal(y.buf.float().cpu().permute(0, 3, 1, 2), ref.detach())
    assert torch.equal(ops.pool_indices_int64(code, W).cpu(), idx)
    dya = ops.Act(dy.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev), 0, C)
    dska = ops.Act(dsk.permute(0, 2, 3, 1).contiguous().to(tdt).to(dev), 0, C)
    dx = ops.maxpool_bwd(dya, code, dska)
    want = (x.grad + dsk).to(tdt).float()
    assert torch.equal(dx.buf.float().cpu().permute(0, 3, 1, 2), want)
    dx0 = ops.maxpool_bwd(dya, code, None, out_hw=(H, W))
    assert torch.equal(dx0.buf.float().cpu().permute(0, 3, 1, 2), x.grad.to(tdt).float())
