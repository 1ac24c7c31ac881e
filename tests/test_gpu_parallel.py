"""Data parallel on the GPU with the real kernels (the reference's DDP wrap, /root/reference/train.py:139-144).

The multi-rank logic is covered on the CPU by tests/test_parallel_gloo.py (world 2, gloo).  Here the wrapper drives the
HIP engine over RCCL: one rank, collectives forced on, in a fresh process so that the process group exists before the
first GPU call (tools/dp_selfcheck.py); and the 1/world factor folded into the fused clip coefficient is checked
against averaging first."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def test_wrapper_over_rccl_equals_unwrapped_bit_for_bit(dev):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29733")
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "dp_selfcheck.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=170)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "DP SELFCHECK OK" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("world", [2, 8])
def test_world_factor_folded_into_clip_coefficient(dev, world):
    """p.grad holds the all-reduced SUM; clip_and_step(grad_scale=1/world) must act like averaging first and then
    clip_grad_norm_ + AdamW: same norm, same parameters (powers of two: the fold only moves an exact scaling)."""
    from unet_amd.optim import FusedAdamW
    g = torch.Generator(device=dev).manual_seed(world)
    shapes = [(64, 32, 3, 3), (64,), (128, 64, 2, 2), (2, 64, 1, 1), (1000,)]
    avg = [torch.randn(s, generator=g, device=dev) for s in shapes]

    def run(scale):
        ps = [torch.nn.Parameter(torch.full(s, 0.5, device=dev)) for s in shapes]
        for p, a in zip(ps, avg):
            p.grad = (a / scale).clone()
        opt = FusedAdamW(ps, lr=1e-2, weight_decay=1e-2)
        norm = opt.clip_and_step(ps, 1.0, grad_scale=scale) if scale != 1.0 else opt.clip_and_step(ps, 1.0)
        return float(norm), [p.detach().clone() for p in ps], [p.grad.clone() for p in ps]

    n1, p1, g1 = run(1.0)
    n2, p2, g2 = run(1.0 / world)
    assert abs(n1 - n2) <= 1e-6 * n1
    for a, b in zip(p1, p2):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(g1, g2):                     # the clipped AVERAGED gradients are what is left in p.grad
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
