"""The two ends of the U-Net fused into their BatchNorm passes (csrc/edge_fused.hip) against torch-CPU autograd through
the reference ops (conv2d -> batch_norm -> relu [-> 1x1 conv], /root/reference/models/unet.py:9-16,30,39,62), and against
the separate kernels they replace.  Tolerances as tests/test_gpu_kernels.py (fp32 2e-5 of max, bf16: storage rounding)."""
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


@pytest.mark.parametrize("dt", [0, 1])
# the last shape has more tiles (640) than the LDS-ring backward kernel has workgroups (512): its request stream crosses tile boundaries
# (2, 18, 40, 8): ONE channel group in bf16 -- a wave's 64 pixels + 2 do not fit its lanes, so the register-prefetch kernels run
@pytest.mark.parametrize("shape", [(2, 20, 36, 16), (1, 33, 31, 64), (3, 16, 32, 24), (2, 64, 64, 64), (20, 128, 512, 64), (2, 18, 40, 8)])
def test_first_layer_fused_vs_torch(dev, dt, shape):
    from unet_amd import ops
    N, H, W, Cout = shape
    x = torch.rand(N, 1, H, W, generator=torch.Generator().manual_seed(51))
    w = rnd((Cout, 1, 3, 3), 52, 0.3)
    b = rnd((Cout,), 53, 0.1)
    gamma, beta = rnd((Cout,), 54).abs() + 0.5, rnd((Cout,), 55) * 0.3
    w.requires_grad_(True); b.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True)
    z = F.conv2d(x, w, b, padding=1)
    zq = z + (q(z.detach(), dt) - z.detach())              # the HIP path normalises the stored (rounded) conv output
    y = F.relu(F.batch_norm(zq, None, None, gamma, beta, True, 0.1, 1e-5))
    dyv = q(rnd(tuple(y.shape), 56), dt)
    y.backward(dyv)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, bd = w.detach().to(dev), b.detach().to(dev)
    ws = ops.Workspace(dev)
    rows = ops.conv_first_stats(xh, wd, bd, dt)
    s = rows.double().sum(0).cpu()
    close(s[:, 0].float(), z.detach().double().sum((0, 2, 3)).float(), dt, "stats-only sum", f32_out=True)
    # the same statistics from the image's 3x3 patch moments (what the engine uses: no conv output is computed at all):
    # mean and variance against an fp64 evaluation of the reference's formula, to 1e-6 of a standard deviation / 1e-5 relative
    rows_m, mom = ops.first_stats_moments(xh, wd, bd)
    assert tuple(rows_m.shape) == (2, Cout, 2) and float(mom[54]) == N * H * W
    sm = rows_m.double().sum(0).cpu()
    z64 = F.conv2d(x.double(), w.detach().double(), b.detach().double(), padding=1)
    n = N * H * W
    mean64, var64 = z64.mean((0, 2, 3)), z64.var((0, 2, 3), unbiased=False)
    mean_m = sm[:, 0] / n
    var_m = sm[:, 1] / n - mean_m * mean_m
    assert float(((mean_m - mean64).abs() / var64.sqrt()).max()) <= 1e-6
    assert float(((var_m - var64).abs() / var64).max()) <= 1e-5
    st = ops.bn_finalize(rows, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), None, None, None, 0.1, 1e-5, ws)
    ya = ops.Act(torch.zeros((N, H, W, Cout + 8), dtype=ops.torch_dtype(dt), device=dev), 8, Cout)
    ops.first_bn_act(xh, wd, bd, st, ya)
    # batch statistics of the rounded tensor differ from those of the fp32 one by O(2^-9 / sqrt(n)): loosen bf16 slightly
    ref_y = y.detach()
    if dt == 0:
        close(from_act(ya), ref_y, dt, "first layer y")
    else:
        assert float((from_act(ya) - ref_y).abs().max()) <= 2 ** -7 * float(ref_y.abs().max())
    assert float(ya.buf[..., :8].float().abs().max()) == 0.0
    dg, db_, dbias, dw = ops.first_bn_relu_bwd(to_act(dyv, dt, dev), xh, wd, bd, st, ws)
    # handing the forward's moments over gives the same numbers as letting the backward compute them (fp64 atomics: to rounding)
    dgm, dbm, _, dwm = ops.first_bn_relu_bwd(to_act(dyv, dt, dev), xh, wd, bd, st, ws, moments=mom)
    assert torch.equal(dgm, dg) and torch.equal(dbm, db_) and float((dwm - dw).abs().max()) <= 1e-6 * float(dw.abs().max())
    if dt == 0 and N * H * W < 500_000:
        # (the 1.3 M-pixel shape: 84 M pre-activations, of which a handful sit within an fp32 rounding of zero and take the
        # other side in torch's conv, whose sum runs in another order; one such pixel moves a channel's dgamma by |dy * xhat|,
        # far above 2e-5 of the sum.  That shape is pinned by the comparison with the separate kernels below, which make the
        # same decisions bit for bit.)
        # bf16: the batch statistics come from the fp32 conv output while the normalised tensor is its bf16 rounding, so a
        # few ReLU decisions differ from torch's and each moves a per-channel sum by ~1/sqrt(n); the bf16 backward is
        # pinned by the comparison with the separate kernels below (and by the whole-model tests)
        close(dg.cpu(), gamma.grad, dt, "dgamma", f32_out=True)
        close(db_.cpu(), beta.grad, dt, "dbeta", f32_out=True)
        close(dw.cpu(), w.grad, dt, "first conv dW", f32_out=True)
    # the conv bias in front of a train-mode BN has a noise-only gradient: absolute bound
    assert float(dbias.cpu().abs().max()) <= 2e-2 * float(dyv.abs().max()) * (N * H * W) ** 0.5
    # against the separate kernels (conv -> z, affine, bn backward, first-conv wgrad): same numbers
    zt = ops.dense(N, H, W, Cout, dt, dev)
    rows2 = ops.conv_first_fwd(xh, wd, bd, zt, 3, 3)
    assert torch.equal(rows2, rows)
    y2 = ops.dense(N, H, W, Cout, dt, dev)
    ops.affine_act(zt, y2, st)
    assert torch.equal(y2.buf, ya.buf[..., 8:])
    dz = ops.dense(N, H, W, Cout, dt, dev)
    dg2, db2, dbias2 = ops.bn_relu_bwd(to_act(dyv, dt, dev), zt, dz, st, ws)
    dw2 = ops.conv_first_wgrad(dz, xh, 3, 3, ws)
    # the two paths add the same terms in different orders (column strips vs strided pixels): fp32 summation noise only.
    # dW: the fused path derives it in ONE pass from sum dy'*patch and the patch moments of x (dz is affine in dy' and z), so
    # in bf16 it skips two roundings the separate kernels make (z and dz to the storage type before the sum): the two differ
    # by that rounding noise, 2^-9 per element averaged over the pixels (fp32: no such roundings, same 1e-4 bound)
    for a_, b_, what in ((dg, dg2, "dgamma"), (db_, db2, "dbeta")):
        assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max()), what
    e_dw = float((dw - dw2).abs().max()) / float(dw2.abs().max())
    if dt == 1 and N * H * W >= 500_000:
        # At 1.3 M pixels the separate kernels' bf16 dz is the worse of the two: c1 = S1/n and xhat*c2 are ~1e-3 against
        # |dy'| ~ 1 with dy' ON the bf16 grid, so rounding dz = scale*(dy' - c1 - xhat*c2) back to bf16 returns scale*dy' and
        # the mean-subtraction is lost (sum of the stored dz: 1.2e3 instead of ~0), which moves that path's dW by 10 %.  The
        # fused pass never rounds dz: it is the one compared with torch here (measured 6e-4 / 9e-5).
        assert e_dw > 2 ** -8, "the separate kernels were expected to be the less accurate path at this size"
        for got, ref, what in ((dw, w.grad, "dW"), (dg, gamma.grad, "dgamma"), (db_, beta.grad, "dbeta")):
            e = float((got.cpu() - ref).abs().max()) / float(ref.abs().max())
            assert e <= 5e-3, f"{what}: fused one-pass vs torch {e:.3e}"
    else:
        assert e_dw <= (1e-4 if dt == 0 else 2 ** -8), f"dW: fused one-pass vs separate kernels {e_dw:.3e}"


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("geom", [(2, 12, 20, 64, 2), (1, 9, 7, 24, 3), (2, 16, 32, 8, 1), (1, 32, 32, 64, 4)])
def test_last_layer_fused_vs_torch(dev, dt, geom):
    from unet_amd import ops
    N, H, W, C, K = geom
    z = q(rnd((N, C, H, W), 61) * 1.5 + 0.2, dt)
    gamma, beta = rnd((C,), 62).abs() + 0.5, rnd((C,), 63) * 0.3
    wf, bf = rnd((K, C, 1, 1), 64, C ** -0.5), rnd((K,), 65)
    z.requires_grad_(True); gamma.requires_grad_(True); beta.requires_grad_(True); wf.requires_grad_(True); bf.requires_grad_(True)
    y = F.relu(F.batch_norm(z, None, None, gamma, beta, True, 0.1, 1e-5))
    yq = y + (q(y.detach(), dt) - y.detach())              # the 1x1 conv reads the stored (rounded) activation
    logits = F.conv2d(yq, wf, bf)
    dl = rnd(tuple(logits.shape), 66)
    logits.backward(dl)
    ws = ops.Workspace(dev)
    za = to_act(z.detach(), dt, dev, ld=C + 8, off=8)
    zz = z.detach().double()
    rows = torch.stack([zz.sum((2, 3)), (zz * zz).sum((2, 3))], -1).float().to(dev).contiguous()
    st = ops.bn_finalize(rows, N * H * W, gamma.detach().to(dev), beta.detach().to(dev), None, None, None, 0.1, 1e-5, ws)
    w2 = wf.detach().view(K, C).contiguous().to(dev)
    got = ops.affine_act_out(za, st, w2, bf.detach().to(dev))
    ref_l = logits.detach()
    tol_l = 2e-5 if dt == 0 else 2 ** -7
    assert float((got.cpu() - ref_l).abs().max()) <= tol_l * float(ref_l.abs().max())
    # against the separate kernels: bit-identical logits
    ya = ops.dense(N, H, W, C, dt, dev)
    ops.affine_act(za, ya, st)
    assert torch.equal(got, ops.conv_out_fwd(ya, w2, bf.detach().to(dev)))
    yb = ops.dense(N, H, W, C, dt, dev)
    got2 = ops.affine_act_out(za, st, w2, bf.detach().to(dev), y=yb)
    assert torch.equal(got2, got) and torch.equal(yb.buf, ya.buf)
    # backward
    dz = ops.dense(N, H, W, C, dt, dev)
    dg, db_, dbias, dwf, dbf = ops.bn_relu_bwd_out(dl.to(dev), w2, za, dz, st, ws)
    if dt == 0:
        close(from_act(dz), z.grad, dt, "dz")
        close(dg.cpu(), gamma.grad, dt, "dgamma")
        close(db_.cpu(), beta.grad, dt, "dbeta")
        close(dwf.cpu(), wf.grad.view(K, C), dt, "final dW")
        close(dbf.cpu(), bf.grad, dt, "final db")
    # against the separate kernels (final-conv backward -> stored dy -> BN backward): same numbers in both dtypes
    dx, dw2, db2 = ops.conv_out_bwd(dl.to(dev), ya, w2, ws)
    dz2 = ops.dense(N, H, W, C, dt, dev)
    dg2, dbeta2, dbias2 = ops.bn_relu_bwd(dx, za, dz2, st, ws)
    for a_, b_, what in ((dg, dg2, "dgamma"), (db_, dbeta2, "dbeta"), (dwf, dw2.view(K, C), "final dW"), (dbf, db2, "final db")):
        assert float((a_ - b_).abs().max()) <= 1e-4 * float(b_.abs().max()) + 1e-7, what
    d = (dz.buf.float() - dz2.buf.float()).abs()
    assert float(d.max()) <= (1e-6 if dt == 0 else 2 ** -7) * float(dz2.buf.float().abs().max()) + 1e-7


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_engine_with_and_without_fused_edges(dev, mode):
    """Whole model: the fused ends give the same logits bit for bit and the same gradients up to the order of fp64 atomic
    adds as the separate kernels (train mode and eval-with-gradients).  Bit equality needs the same statistics on both sides, so
    this comparison runs the fused first layer with the statistics-only conv pass (first_stats_moments = False); the
    patch-moment statistics the engine uses by default are checked against fp64 in test_first_layer_fused_vs_torch and
    against the oracle in every whole-model test."""
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    torch.manual_seed(3)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32]).to(dev)
    m.compute_dtype = mode
    g = torch.Generator().manual_seed(4)
    x = torch.rand(3, 1, 32, 64, generator=g).to(dev)
    y = (torch.rand(3, 32, 64, generator=g) > 0.9).long().to(dev)
    loss_fn = get_loss("ce_dice", ignore_index=255)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    for training in (True, False):
        outs = []
        for fused in (True, False):
            m.load_state_dict(state)
            m.train(training)
            m._engine.fuse_edges = fused
            m._engine.first_stats_moments = False
            m.zero_grad(set_to_none=True)
            lg = m(x)
            loss_fn(lg, y).backward()
            outs.append((lg.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                         {k: b.clone() for k, b in m.named_buffers()}))
        m._engine.fuse_edges = True
        m._engine.first_stats_moments = True
        assert torch.equal(outs[0][0], outs[1][0])
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
        for k in outs[0][1]:
            a, b = outs[0][1][k], outs[1][1][k]
            if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"):
                continue                                     # noise-only gradients in train mode
            if mode == "bf16" and k == "downs.0.double_conv.0.weight":
                # the fused first layer takes dW from sum dy'*patch and the patch moments of x in one pass: it skips the
                # storage-type roundings of z and dz that the separate kernels put into this sum (see the test above)
                assert float((a - b).abs().max()) <= 2 ** -8 * float(b.abs().max()), k
                continue
            torch.testing.assert_close(a, b, rtol=2e-4 if mode == "bf16" else 2e-5, atol=1e-6 * float(b.abs().max()) + 1e-9, msg=k)
