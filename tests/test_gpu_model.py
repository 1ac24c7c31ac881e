"""Whole-model parity on the GPU against golden vectors captured from the real reference
(tests/golden/make_goldens.py): forward, loss, every gradient, BN buffers, one full step_batch.

  fp32 mode : logits / loss / grads within 1e-4 relative (max-abs error over max-abs reference per
              tensor); argmax mask bit-exact; conv biases in front of train-mode BN have noise-only
              gradients (|g| ~ 1e-8 in the reference) and are held to an absolute bound instead.
  bf16 mode : stated tolerance: logits rel-L2 <= 3e-2, loss <= 2e-2 relative, argmax agreement >= 98 %
              against the fp32 reference golden.  Gradients: this network amplifies storage-rounding noise
              layer by layer on the way back (train-mode BN + ReLU at random init; even fp32-vs-fp64 is 1e-3
              at the deepest layers, see tests/diag_bf16_precision.py), so a fixed bound would be meaningless.  The
              bound is relative to a CPU emulation of the same storage precision: per tensor,
              relL2(hip_bf16, fp32 ref) <= 1.6 * relL2(oracle with bf16 rounding at the same points, fp32 ref) + 1e-2.
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(G, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def split(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rell2(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


CASES = {
    "unet_f8x16": dict(in_channels=1, n_classes=2, features=[8, 16]),
    "unet_f8x16x32x64": dict(in_channels=1, n_classes=2, features=[8, 16, 32, 64]),
    "unet_f16_c3": dict(in_channels=1, features=[16]),
}


def build(name, dev):
    from unet_amd.models import REGISTRY
    g = load(name)
    torch.manual_seed(42)
    m = REGISTRY["unet"](**CASES[name])
    init = split(g, "init/")
    sd = m.state_dict()
    assert list(sd) == list(init)
    for k in sd:                                  # same seed + same registration order => same init
        assert torch.equal(sd[k], init[k]), k
    return m.to(dev), g


def noise_bias(k):
    return k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias")


@pytest.mark.parametrize("name", list(CASES))
def test_unet_fp32_matches_reference(dev, name):
    from unet_amd.losses import get_loss
    m, g = build(name, dev)
    m.train()
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
    logits = m(g["x"].to(dev))
    loss = loss_fn(logits, g["y"].to(dev))
    loss.backward()
    assert logits.dtype == torch.float32 and tuple(logits.shape) == tuple(g["logits"].shape)
    assert relmax(logits.detach().cpu(), g["logits"]) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert torch.equal(logits.argmax(1).cpu(), g["argmax"])                  # bit-exact mask
    from unet_amd import ops
    assert torch.equal(ops.argmax(logits.detach()).cpu(), g["argmax"])
    gmax = max(float(v.abs().max()) for k, v in g.items() if k.startswith("grad/"))
    for k, p in m.named_parameters():
        ref = g["grad/" + k]
        assert p.grad is not None, k
        if noise_bias(k):
            assert float((p.grad.cpu() - ref).abs().max()) < 1e-5 * gmax, k
        else:
            assert relmax(p.grad.cpu(), ref) < 1e-4, (k, relmax(p.grad.cpu(), ref))
    for k, b in m.named_buffers():
        ref = g["buf1/" + k]
        if ref.dtype == torch.int64:
            assert torch.equal(b.cpu(), ref), k
        else:
            assert relmax(b.cpu(), ref) < 1e-5, k


@pytest.mark.parametrize("name", ["unet_f8x16", "unet_f8x16x32x64"])
def test_unet_bf16_within_stated_tolerance(dev, name):
    from oracle import loss_ref, unet_ref
    from unet_amd.losses import get_loss
    m, g = build(name, dev)
    m.train()
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
    with torch.amp.autocast(device_type="cuda", dtype=torch.bfloat16):
        logits = m(g["x"].to(dev))
        loss = loss_fn(logits, g["y"].to(dev))
    loss.backward()
    assert rell2(logits.detach().float().cpu(), g["logits"]) < 3e-2
    assert abs(float(loss) - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    agree = (logits.argmax(1).cpu() == g["argmax"]).float().mean()
    assert float(agree) >= 0.98
    # CPU emulation of bf16 storage (oracle/unet_ref.py, emulate_bf16) as the yardstick for gradients
    st = {k: v.clone() for k, v in split(g, "init/").items()}
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names:
        st[k].requires_grad_(True)
    loss_ref.ce_dice(unet_ref.unet_forward(g["x"], st, True, {}, emulate_bf16=True), g["y"], 1.0, 1.0, 255).backward()
    for k, p in m.named_parameters():
        if noise_bias(k):
            continue
        ref = g["grad/" + k]
        e_hip, e_emu = rell2(p.grad.cpu(), ref), rell2(st[k].grad, ref)
        assert e_hip <= 1.6 * e_emu + 1e-2, (k, e_hip, e_emu)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", ["unet_f8x16", "unet_f8x16x32x64"])
def test_step_batch_matches_reference(dev, name, fused):
    """zero_grad -> fwd -> loss -> bwd -> clip 1.0 -> AdamW(1e-4, wd 1e-4): parameters after one step,
    with torch's AdamW and with the fused multi-tensor tail that get_optim("adamw") builds."""
    import logging
    from unet_amd.losses import get_loss
    from unet_amd.utils.builders import get_optim
    from unet_amd.utils.train_step import step_batch
    m, g = build(name, dev)
    m.train()
    make = (lambda ps, **kw: get_optim("adamw", ps, **kw)) if fused else torch.optim.AdamW
    opt = make([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    tr = types.SimpleNamespace(optimizer=opt, amp=False, model=m, grad_clip=1.0, cbs=[], current_epoch=1,
                               logger=logging.getLogger("t"),
                               loss_fn=get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255))
    seen = []
    tr.cbs.append(types.SimpleNamespace(on_batch_end=lambda t, s, info: seen.append((s, dict(info)))))
    loss, logits, info = step_batch(tr, g["x"].to(dev), g["y"].to(dev).unsqueeze(1), torch.amp.GradScaler(enabled=False), 0)
    assert tr._global_step == int(g["step/global_step"]) and seen and seen[0][0] == 1
    assert abs(info["main_loss"] - float(g["step/main_loss"])) < 1e-4 * abs(float(g["step/main_loss"]))
    assert info["lr"] == float(g["step/lr"])
    assert relmax(logits.detach().cpu(), g["logits"]) < 1e-4
    for k, p in m.named_parameters():
        # Adam's first update is lr*sign-like: elements whose gradient is near eps move by up to lr
        tol = 2.1e-4 if noise_bias(k) else 2e-5
        assert float((p.detach().cpu() - g["step/" + k]).abs().max()) < tol, k
    before = {k: p.detach().clone() for k, p in m.named_parameters()}
    moments = [opt.state[p]["exp_avg"].clone() for p in m.parameters()] if fused else None
    with pytest.raises(RuntimeError):          # label the reference's one_hot would raise on
        step_batch(tr, g["x"].to(dev), torch.full_like(g["y"], 255).to(dev), torch.amp.GradScaler(enabled=False), 1)
    if fused:
        # the reference raises inside loss_fn, before backward: the bad batch must not have moved weights or moments
        # (the loss kernel's illegal-label counter vetoes the fused update on the device)
        for k, p in m.named_parameters():
            assert torch.equal(p.detach(), before[k]), k
        for p, mo in zip(m.parameters(), moments):
            assert torch.equal(opt.state[p]["exp_avg"], mo)


def test_eval_mode_uses_running_stats_and_no_grad(dev):
    from oracle import unet_ref
    m, g = build("unet_f8x16", dev)
    st = {k: v.clone() for k, v in split(g, "init/").items()}
    gen = torch.Generator().manual_seed(3)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = torch.randn(st[k].shape, generator=gen) * 0.1
        if k.endswith("running_var"):
            st[k] = torch.rand(st[k].shape, generator=gen) + 0.5
    m.load_state_dict(st)
    m.eval()
    with torch.no_grad():
        out = m(g["x"].to(dev), current_epoch=3)
        ref = unet_ref.unet_forward(g["x"], st, train=False)
    assert relmax(out.cpu(), ref) < 1e-4
    for k, b in m.named_buffers():
        assert torch.equal(b.cpu(), st[k]), k                    # eval must not touch the buffers


def test_trainer_fit_and_evaluate_smoke(dev):
    from torch.utils.data import DataLoader, TensorDataset
    from unet_amd.losses import get_loss
    from unet_amd.models import REGISTRY
    from unet_amd.trainer import Trainer
    torch.manual_seed(0)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16]).to(dev)
    gen = torch.Generator().manual_seed(1)
    x = torch.rand(8, 1, 32, 32, generator=gen)
    y = (torch.rand(8, 32, 32, generator=gen) > 0.9).long()
    dl = DataLoader(TensorDataset(x, y), batch_size=4)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    cfg = {"metrics": ["acc_pixel", "dice_macro", "dice_class_1", "mae"], "early_stop": {"monitor": "dice_class_1"},
           "model_args": {"n_classes": 2}}
    ends = []
    cb = types.SimpleNamespace(on_train_begin=lambda t: None, on_epoch_end=lambda *a: ends.append(a[1:]))
    tr = Trainer(m, get_loss("ce_dice", ignore_index=255), opt, amp=True, grad_clip=1.0, device=dev,
                 callbacks=[cb], progress_bar=False, cfg=cfg)
    tr.fit(dl, dl, epochs=2)
    assert len(ends) == 2 and tr._global_step == 4
    assert all(np.isfinite(e[1]) and np.isfinite(e[3]) for e in ends)
    vl, score, stats = tr.evaluate(dl, epoch=2)
    assert set(stats) >= {"val_loss", "acc_pixel", "dice_macro", "dice_class_1", "mae"}
    assert score == stats["dice_class_1"]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_eval_bn_fold_matches_unfolded(dev, dtype):
    """Inference folds BatchNorm into the conv (+ReLU epilogue); it must agree with conv -> BN(eval) -> ReLU, and
    must follow parameter / running-statistics updates (training step, load_state_dict)."""
    m, g = build("unet_f8x16x32x64", dev)
    m.compute_dtype = dtype
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, b in m.named_buffers():
            if k.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=gen) * 0.1)
            if k.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=gen) + 0.5)
    x = g["x"].to(dev)
    tol = 1e-5 if dtype == "fp32" else 2e-2

    def both():
        m.eval()
        with torch.no_grad():
            m._engine.fold_bn_eval = True
            a = m(x).float().cpu()
            m._engine.fold_bn_eval = False
            b = m(x).float().cpu()
            m._engine.fold_bn_eval = True
        return a, b

    a, b = both()
    assert rell2(a, b) < tol
    m.train()                                      # a training forward moves the running statistics under raw pointers
    m(x)
    a2, b2 = both()
    assert rell2(a2, b2) < tol and rell2(a2, a) > 1e-2
    sd = {k: (v * 1.5 if k.endswith("double_conv.0.weight") else v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    a3, b3 = both()
    assert rell2(a3, b3) < tol and rell2(a3, a2) > 1e-2


def test_wgrad_side_stream_gives_identical_gradients(dev):
    """UNET_AMD_OVERLAP_WGRAD mode: weight gradients on a side stream; same numbers, several steps in a row
    (the caching allocator must not recycle a dz buffer the side stream still reads)."""
    m, g = build("unet_f8x16x32x64", dev)
    m.compute_dtype = "bf16"
    m.train()
    x, y = g["x"].to(dev), g["y"].to(dev)
    from unet_amd.losses import get_loss
    loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)

    def grads(overlap):
        m._engine.overlap_wgrad = overlap
        out = None
        for _ in range(3):
            m.zero_grad(set_to_none=True)
            loss_fn(m(x), y).backward()
            torch.cuda.synchronize()
            cur = {k: p.grad.clone() for k, p in m.named_parameters()}
            if out is not None:
                for k in cur:
                    if k.endswith(".weight") and cur[k].dim() == 4:
                        assert torch.equal(cur[k], out[k]), k          # run-to-run reproducible (fixed split order)
            out = cur
        return out

    a, b = grads(False), grads(True)
    m._engine.overlap_wgrad = False
    for k in a:
        if k.endswith(".weight") and a[k].dim() == 4:
            assert torch.equal(a[k], b[k]), k
        else:
            torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-6)


def _randomize_running_stats(st, seed=3):
    gen = torch.Generator().manual_seed(seed)
    for k in st:
        if k.endswith("running_mean"):
            st[k] = torch.randn(st[k].shape, generator=gen) * 0.1
        if k.endswith("running_var"):
            st[k] = torch.rand(st[k].shape, generator=gen) + 0.5
    return st


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_eval_mode_backward_matches_oracle(dev, dtype):
    """loss.backward() under model.eval() (frozen-BN fine-tuning, saliency): BatchNorm uses the running statistics, which
    carry no gradient -- dz = scale*dy', dgamma = sum dy'*(z-rm)*invstd, dbeta = sum dy'.  Against torch autograd through
    the oracle's eval-mode forward (F.batch_norm(training=False))."""
    from oracle import loss_ref, unet_ref
    from unet_amd.losses import get_loss
    m, g = build("unet_f8x16x32x64", dev)
    st = _randomize_running_stats({k: v.clone() for k, v in split(g, "init/").items()})
    m.load_state_dict(st)
    m.eval()
    m.compute_dtype = dtype
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names:
        st[k].requires_grad_(True)
    ref = unet_ref.unet_forward(g["x"], st, train=False)
    loss_ref.ce_dice(ref, g["y"], 1.0, 1.0, 255).backward()
    out = m(g["x"].to(dev))
    get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(out, g["y"].to(dev)).backward()
    for k, b in m.named_buffers():
        assert torch.equal(b.cpu(), st[k].detach()), k                     # eval must not touch the buffers
    if dtype == "fp32":
        assert relmax(out.detach().cpu(), ref.detach()) < 1e-4
        for k, p in m.named_parameters():
            assert p.grad is not None, k
            assert relmax(p.grad.cpu(), st[k].grad) < 1e-4, (k, relmax(p.grad.cpu(), st[k].grad))
    else:
        assert rell2(out.detach().float().cpu(), ref.detach()) < 2e-2
        for k, p in m.named_parameters():
            assert rell2(p.grad.cpu(), st[k].grad) < 6e-2, (k, rell2(p.grad.cpu(), st[k].grad))


def test_input_requiring_grad_is_refused(dev):
    from unet_amd import UNetHipError
    m, g = build("unet_f8x16", dev)
    x = g["x"].to(dev).requires_grad_(True)
    with pytest.raises(UNetHipError):
        m(x)
    with torch.no_grad():
        m(x)                                   # fine when no graph is being built
