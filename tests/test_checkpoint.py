"""Checkpoint interop: the reference's checkpoint format and key fix-ups
(/root/reference/utils/checkpoint_utils.py:15-89, utils/io.py:19-45), and deploy-mode RepUNet state_dicts."""
import types

import pytest
import torch

from unet_amd.models import REGISTRY
from unet_amd.utils import checkpoint_utils as cu
from unet_amd.utils import io as uio


def small_unet():
    torch.manual_seed(0)
    return REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])


def test_save_load_roundtrip_reference_format(tmp_path):
    m = small_unet()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=5)
    tr = types.SimpleNamespace(model=m, optimizer=opt, scheduler=sched, scaler=torch.amp.GradScaler(enabled=False),
                               latest_val_stats={"dice_class_1": 0.5}, cfg={"model": "unet", "epochs": 3})
    cu.save_checkpoint(str(tmp_path / "ck"), tr, epoch=4, is_best=True)
    assert (tmp_path / "ck" / "last.ckpt").exists() and (tmp_path / "ck" / "best.ckpt").exists()
    raw = torch.load(tmp_path / "ck" / "best.ckpt", weights_only=True)
    assert set(raw) >= {"epoch", "model_state", "optim_state", "scheduler_state", "scaler_state", "cfg", "model", "opt", "sched",
                        "trainer_backbone_frozen", "latest_val_stats"}
    assert list(raw["model_state"].keys()) == list(m.state_dict().keys())
    m2 = small_unet()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    opt2 = torch.optim.AdamW(m2.parameters(), lr=1e-3)
    out = cu.load_checkpoint(str(tmp_path / "ck" / "last.ckpt"), m2, optimizer=opt2, map_location="cpu", strict=True)
    assert out["start_epoch"] == 5 and out["raw"]["cfg"]["model"] == "unet"
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(FileNotFoundError):
        cu.load_checkpoint(str(tmp_path / "nope.ckpt"), m2)


def test_module_prefix_fixups(tmp_path):
    m = small_unet()
    sd = {"module." + k: v for k, v in m.state_dict().items()}           # saved from a DDP-wrapped reference model
    torch.save({"epoch": 1, "model": sd}, tmp_path / "ddp.ckpt")
    m2 = small_unet()
    with torch.no_grad():
        m2.final_conv.bias.fill_(7.0)
    cu.load_checkpoint(str(tmp_path / "ddp.ckpt"), m2, map_location="cpu", strict=True)
    assert torch.equal(m2.final_conv.bias, m.final_conv.bias)
    m3 = small_unet()
    with torch.no_grad():
        m3.final_conv.bias.fill_(7.0)
    uio.load_checkpoint(str(tmp_path / "ddp.ckpt"), m3, device="cpu")
    assert torch.equal(m3.final_conv.bias, m.final_conv.bias)
    torch.save({"model_state_dict": m.state_dict()}, tmp_path / "msd.ckpt")
    torch.save(m.state_dict(), tmp_path / "bare.ckpt")
    for f in ("msd.ckpt", "bare.ckpt"):
        m4 = small_unet()
        with torch.no_grad():
            m4.final_conv.bias.fill_(7.0)
        uio.load_checkpoint(str(tmp_path / f), m4, device="cpu")
        assert torch.equal(m4.final_conv.bias, m.final_conv.bias)


def test_data_parallel_wrapper_is_unwrapped(tmp_path):
    from unet_amd.parallel import DataParallel
    m = small_unet()
    wrapped = DataParallel.__new__(DataParallel)          # no process group on this host: only `.module` matters here
    torch.nn.Module.__init__(wrapped)
    wrapped.module = m
    tr = types.SimpleNamespace(model=wrapped, optimizer=None, scheduler=None)
    cu.save_checkpoint(str(tmp_path / "dp"), tr, epoch=0)
    raw = torch.load(tmp_path / "dp" / "last.ckpt", weights_only=True)
    assert not any(k.startswith("module.") for k in raw["model_state"])


def test_deployed_repunet_state_dict_loads_into_fresh_model():
    """Keys of a deployed ReparamUNet are `...reparam_conv.{0,2}.mergedConv.{weight,bias}` (reference
    models/reparm_unet.py:139-144); a fresh model grows the deploy skeleton when it meets them."""
    torch.manual_seed(1)
    kw = dict(in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=[3, 5], norm="batch")
    a = REGISTRY["reparam_unet"](**kw)
    a.make_deploy_skeleton()
    with torch.no_grad():
        for p in a.parameters():
            p.copy_(torch.randn_like(p))
    sd = a.state_dict()
    assert any(k.endswith("reparam_conv.0.mergedConv.weight") for k in sd) and not any("convSq" in k for k in sd)
    assert sd["downs.0.reparam_conv.0.mergedConv.weight"].shape == (8, 1, 5, 5)
    b = REGISTRY["reparam_unet"](**kw)
    res = b.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in b.state_dict().items():
        assert torch.equal(v, sd[k]), k
    c = REGISTRY["reparam_unet"](**kw)                    # an un-deployed checkpoint still loads as before
    c.load_state_dict(REGISTRY["reparam_unet"](**kw).state_dict(), strict=True)
    assert any("convSq" in k for k in c.state_dict())


@pytest.mark.gpu
def test_deployed_checkpoint_runs_identically_on_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    kw = dict(in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=[3, 5], norm="batch")
    a = REGISTRY["reparam_unet"](**kw).to(dev).eval()
    x = torch.rand(2, 1, 32, 32, device=dev)
    a.switch_to_deploy()
    with torch.no_grad():
        ya = a(x)
    tr = types.SimpleNamespace(model=a, optimizer=None, scheduler=None)
    cu.save_checkpoint(str(tmp_path / "dep"), tr, epoch=9, is_best=True)
    b = REGISTRY["reparam_unet"](**kw).to(dev).eval()
    out = cu.load_checkpoint(str(tmp_path / "dep" / "best.ckpt"), b, strict=True)
    assert out["start_epoch"] == 10
    with torch.no_grad():
        yb = b(x)
    assert torch.equal(ya, yb)
