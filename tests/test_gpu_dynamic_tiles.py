"""Dynamic tile queues of the persistent bf16 conv / weight-gradient kernels (csrc/tile_queue.h, unet_set_dynamic_tiles)
against the static partition they replace: which workgroup computes a tile must not change the tile.  Activations and the
per-tile statistics rows are bit-identical; the weight gradient sums its split-K partials in a timing-dependent grouping, so
it is compared to fp32 summation noise and against torch like tests/test_gpu_realwidth.py does.  Level 2 also switches the
queues on for launches with few tiles per workgroup, where most workgroups find their own queue dry and steal."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_kernels import close, from_act, q, rnd, to_act

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def dyn():
    from unet_amd import ops
    yield ops.set_dynamic_tiles
    ops.set_dynamic_tiles(0)


# (N, H, W, Cin, Cout): >= 256 tiles of 8x32 pixels x (128 | 64) channels; the last one has four rounds (level 1 regime)
DYN_CONV = [
    (2, 192, 192, 64, 128),      # 288 tiles, 128-channel tile kernel: 36 per queue, the prologue draws 96 -> stealing
    (1, 128, 256, 128, 256),     # 256 tiles, two channel blocks
    (2, 192, 192, 64, 64),       # request-wave kernel (64-channel tiles)
    (3, 200, 168, 64, 64),       # ragged tiles at both edges
    (4, 256, 256, 64, 128),      # 1024 tiles
]


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("shape", DYN_CONV)
def test_conv_dynamic_tiles_bit_identical(dev, dyn, level, shape):
    from unet_amd import ops
    N, H, W, Cin, Cout = shape
    dt = ops.BF16
    g = torch.Generator(device="cpu").manual_seed(7)
    xa = ops.Act((torch.randn((N, H, W, Cin), generator=g)).to(torch.bfloat16).to(dev), 0, Cin)
    w = (torch.randn((Cout, Cin, 3, 3), generator=g) * (Cin * 9) ** -0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    wp = ops.pack_weights(w, 0, dt)
    outs = []
    for lv in (0, level, level):
        dyn(lv)
        ya = ops.dense(N, H, W, Cout, dt, dev)
        ya.buf.fill_(float("nan"))
        # per-tile statistics rows (per_image_stats) do not depend on the workgroup; 128-channel tiles have them by default
        rows = ops.conv_fwd(xa, wp, b, ya, 3, 3, Cin, Cout, stats=True, per_image_stats=(Cout == 64))
        plain = ops.dense(N, H, W, Cout, dt, dev)
        plain.buf.fill_(float("nan"))
        ops.conv_fwd(xa, wp, None, plain, 3, 3, Cin, Cout)
        outs.append((ya.buf, rows, plain.buf))
    torch.cuda.synchronize()
    assert not bool(torch.isnan(outs[0][0].float()).any())
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]), "conv output"
        assert torch.equal(o[1], outs[0][1]), "statistics rows"
        assert torch.equal(o[2], outs[0][2]), "conv output, no statistics"
    # and the static result is the conv (one shape against torch; the others are covered by test_gpu_realwidth)
    if shape == DYN_CONV[0]:
        ref = F.conv2d(xa.buf.float().permute(0, 3, 1, 2).cpu(), w.to(torch.bfloat16).float().cpu(), b.cpu(), padding=1)
        close(from_act(ops.Act(outs[1][0], 0, Cout)), ref, dt, "conv fwd, dynamic tiles")


@pytest.mark.parametrize("level", [1, 2])
@pytest.mark.parametrize("shape", [(2, 192, 192, 64, 128), (2, 128, 128, 128, 128), (4, 256, 256, 64, 64), (1, 64, 96, 256, 512),
                                   (16, 32, 32, 512, 512)])     # the last one: too many channel blocks for the queues, static
def test_wgrad_dynamic_tiles(dev, dyn, level, shape):
    from unet_amd import ops
    N, H, W, Cin, Cout = shape
    dt = ops.BF16
    x = q(rnd((N, Cin, H, W), 201), dt)
    dyv = q(rnd((N, Cout, H, W), 202), dt)
    xa, dza = to_act(x, dt, dev), to_act(dyv, dt, dev)
    ws = ops.Workspace(dev)
    dyn(0)
    dw0, _, _ = ops.conv_wgrad(dza, xa, 3, 3, ws)
    dw0 = dw0.clone()
    for _ in range(2):
        dyn(level)
        dw1, _, _ = ops.conv_wgrad(dza, xa, 3, 3, ws)
        torch.cuda.synchronize()
        assert float((dw1 - dw0).abs().max()) <= 2e-6 * float(dw0.abs().max()), "dynamic vs static partition"
    if shape[0] * shape[1] * shape[2] <= 2 * 192 * 192:
        w = torch.zeros((Cout, Cin, 3, 3), requires_grad=True)
        F.conv2d(x, w, None, padding=1).backward(dyv)
        close(dw1.view(Cout, Cin, 3, 3).cpu(), w.grad, dt, "conv wgrad, dynamic tiles", f32_out=True)


def test_engine_step_dynamic_tiles(dev, dyn):
    """Whole model at a width and size where the level-1 thresholds switch the queues on: same logits bit for bit, gradients
    to summation noise."""
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    torch.manual_seed(5)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128]).to(dev)
    m.compute_dtype = "bf16"
    g = torch.Generator().manual_seed(6)
    x = torch.rand(4, 1, 256, 256, generator=g).to(dev)
    y = (torch.rand(4, 256, 256, generator=g) > 0.9).long().to(dev)
    loss_fn = get_loss("ce_dice", ignore_index=255)
    outs = []
    for lv in (0, 1, 2):
        dyn(lv)
        m.zero_grad(set_to_none=True)
        lg = m(x)
        loss_fn(lg, y).backward()
        torch.cuda.synchronize()
        outs.append((lg.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0])
        for k, a in o[1].items():
            b = outs[0][1][k]
            if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                continue                                     # noise-only gradients in train mode
            assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max()) + 1e-9, k
