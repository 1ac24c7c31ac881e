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


def test_two_ranks_one_gpu_rehearsal(dev):
    """World size 2 with the real kernels (the 8-GPU run is the driver's): two ranks share this GPU, launched the way the
    driver launches bench.py (torchrun child processes; the launcher itself never touches the GPU), process group on gloo
    with the collectives staged through host copies (RCCL refuses two ranks on one device).  tools/dist_rehearsal.py checks
    rank-0 broadcast at wrap, p.grad == average of the per-rank gradients, several arrival-order buckets, parameters
    identical on both ranks after full step_batch calls, and that an illegal label on one rank vetoes the update on both."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-u", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29741", os.path.join(ROOT, "tools", "dist_rehearsal.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=280)
    print(r.stdout[-4000:])
    assert r.returncode == 0 and "DIST REHEARSAL OK (world 2)" in r.stdout, r.stdout[-4000:]


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


def test_c_abi_rccl_entry_points_one_rank(dev):
    """The RCCL entry points of the C ABI (unet_comm_* / unet_allreduce_bucket / unet_broadcast: what a host without
    torch.distributed uses for train.py:139-144), one rank, in a fresh process (a communicator of its own next to the other
    tests' process group is not something a real host would do): the sum over one rank and the broadcast from rank 0 leave the
    buffers as they were, for fp32 and bf16, including a count that is not a multiple of anything."""
    code = r'''
import ctypes, sys, torch
sys.path.insert(0, %r)
from unet_amd._lib import lib, check
L = lib()
torch.cuda.set_device(0)
ident = ctypes.create_string_buffer(128)
check(L.unet_comm_unique_id(ident), "unique_id")
comm = ctypes.c_void_p()
check(L.unet_comm_init(ctypes.byref(comm), 1, 0, ident), "comm_init")
s = torch.cuda.current_stream().cuda_stream
for dt, code_ in ((torch.float32, 0), (torch.bfloat16, 1)):
    x = torch.randn(1000003, device="cuda").to(dt)
    ref = x.clone()
    check(L.unet_allreduce_bucket(comm, x.data_ptr(), x.numel(), code_, s), "allreduce")
    check(L.unet_broadcast(comm, x.data_ptr(), x.numel(), code_, 0, s), "broadcast")
    torch.cuda.synchronize()
    assert torch.equal(x, ref), dt
assert L.unet_allreduce_bucket(comm, None, 5, 0, s) != 0 and b"bad arguments" in L.unet_last_error()
assert L.unet_allreduce_bucket(comm, x.data_ptr(), 5, 7, s) != 0
check(L.unet_comm_destroy(comm), "destroy")
print("RCCL C ABI OK")
''' % ROOT
    r = subprocess.run([sys.executable, "-u", "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=170)
    print(r.stdout[-3000:])
    assert r.returncode == 0 and "RCCL C ABI OK" in r.stdout, r.stdout[-3000:]
