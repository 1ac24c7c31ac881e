"""unet_bn_relu_bwd_noalloc (include/unet_hip.h): the BN+ReLU backward whose workspaces live in a device array inside
the library, for callers that cannot allocate.  Called through the C ABI and checked against unet_bn_relu_bwd (caller
workspaces) on two calls in a row, the second with a smaller input -- the shared array keeps the first call's
contents, so anything that must start at zero has to be cleared by the call itself.

Bar: the two entry points run the same kernels; the only freedom is the order of the fp64 atomic adds, so
dgamma / dbeta / dbias agree to fp32 rounding and dz to one rounding of the storage type.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _case(dev, dt, N, H, W, C, seed, poison):
    from unet_amd import ops
    from unet_amd._lib import check, lib
    g = torch.Generator(device=dev).manual_seed(seed)
    tdt = ops.torch_dtype(dt)
    z = (torch.randn((N, H, W, C), generator=g, device=dev) * 1.5 + 0.3).to(tdt)
    dy = torch.randn((N, H, W, C), generator=g, device=dev).to(tdt)
    gamma = torch.rand(C, generator=g, device=dev) + 0.5
    beta = torch.randn(C, generator=g, device=dev) * 0.2
    zf = z.float()
    mean = zf.mean((0, 1, 2)).contiguous()
    invstd = (1.0 / torch.sqrt(zf.var((0, 1, 2), unbiased=False) + 1e-5)).contiguous()
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    M = N * H * W
    s = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()

    def run(noalloc):
        dz = torch.full((N, H, W, C), poison, dtype=tdt, device=dev)
        out = torch.full((3, C), poison, dtype=torch.float32, device=dev)
        if noalloc:
            check(lib().unet_bn_relu_bwd_noalloc(dt, 1, p(dy), C, 0, p(z), C, 0, p(dz), C, 0, p(scale), p(shift), p(mean), p(invstd),
                                                 p(out[0]), p(out[1]), p(out[2]), M, C, s), "bn_relu_bwd_noalloc")
        else:
            ws = ops.Workspace(dev)
            check(lib().unet_bn_relu_bwd(dt, 1, p(dy), C, 0, p(z), C, 0, p(dz), C, 0, p(scale), p(shift), p(mean), p(invstd),
                                         p(out[0]), p(out[1]), p(out[2]), p(ws.c1c2), p(ws.acc), M, C, s), "bn_relu_bwd")
        torch.cuda.synchronize()
        return dz, out

    dz_a, out_a = run(True)
    dz_b, out_b = run(False)
    torch.testing.assert_close(out_a, out_b, rtol=2e-6, atol=1e-6)
    if dt == 0:
        torch.testing.assert_close(dz_a, dz_b, rtol=2e-6, atol=1e-6)
    else:
        d = (dz_a.float() - dz_b.float()).abs()
        assert float((d > dz_b.float().abs() * 2 ** -7 + 1e-6).sum()) == 0
        assert float((d > 0).float().mean()) < 1e-3            # an ulp of a coefficient moves almost no rounded output
    # and against torch autograd, so that "equal" is also "right"
    zr = zf.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = torch.relu(torch.nn.functional.batch_norm(zr.permute(0, 3, 1, 2), None, None, gr, br, True, 0.1, 1e-5))
    y.backward(dy.float().permute(0, 3, 1, 2))
    tol = 2e-5 if dt == 0 else 2e-2
    assert float((out_a[0] - gr.grad).abs().max()) <= tol * float(gr.grad.abs().max())
    assert float((out_a[1] - br.grad).abs().max()) <= tol * float(br.grad.abs().max())
    assert float((dz_a.float() - zr.grad).abs().max()) <= tol * float(zr.grad.abs().max())


@pytest.mark.parametrize("dt", [0, 1])
def test_noalloc_matches_workspace_entry_point_twice(dev, dt):
    _case(dev, dt, 2, 16, 24, 1024, seed=1, poison=7.0)         # fills 1024 channels' worth of the shared array
    _case(dev, dt, 1, 6, 10, 64, seed=2, poison=-3.0)           # smaller: stale sums of the first call must not leak in
    _case(dev, dt, 3, 8, 8, 24, seed=3, poison=5.0)             # non-power-of-two channel-group count


def test_noalloc_rejects_sizes_beyond_the_scratch(dev):
    from unet_amd._lib import lib
    t = torch.zeros(8, device=dev)
    rc = lib().unet_bn_relu_bwd_noalloc(1, 1, t.data_ptr(), 4096, 0, t.data_ptr(), 4096, 0, t.data_ptr(), 4096, 0,
                                        t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None,
                                        1, 4096, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    assert b"exceeds the built-in scratch" in lib().unet_last_error()
