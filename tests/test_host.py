"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the host
mirror keeps the reference's names / state_dict / config behaviour, and the product path refuses to
run without a GPU (no CPU fallback)."""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    from unet_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 25
    L = _lib.lib()
    for name in protos:
        assert hasattr(L, name), name
    assert L.unet_version() >= 100
    assert isinstance(L.unet_last_error(), bytes)


def test_host_argument_checks_report_errors_without_a_gpu():
    from unet_amd import _lib
    L = _lib.lib()
    # bad dtype is rejected on the host before anything touches a device
    rc = L.unet_conv_fwd(7, 3, 3, 1, 0, None, None, None, None, None, 1, 8, 8, 8, 8, 0, 8, 8, 8, 8, 0, 1, 1, None, 0, 0, 0, 0, None)
    assert rc != 0 and b"dtype" in L.unet_last_error()
    rc = L.unet_conv_fwd(1, 3, 3, 1, 0, None, None, None, None, None, 1, 8, 8, 12, 16, 0, 8, 8, 8, 8, 0, 1, 1, None, 0, 0, 0, 0, None)
    assert rc != 0 and b"multiples of 8" in L.unet_last_error()
    # a well-formed problem with null tensors never reaches a launch
    rc = L.unet_conv_fwd(1, 3, 3, 1, 0, None, None, None, None, None, 1, 8, 8, 16, 16, 0, 8, 8, 8, 8, 0, 1, 1, None, 0, 0, 0, 0, None)
    assert rc != 0 and b"null pointer" in L.unet_last_error()
    assert L.unet_conv_tiles(1, 3, 3, 1, 0, 32, 512, 512, 128, 128, 0, 0) == 32 * 32 * 16 and L.unet_conv_tiles(0, 3, 3, 1, 0, 32, 512, 512, 64, 64, 0, 0) == 32 * 64 * 16
    assert L.unet_conv_tiles(1, 3, 3, 1, 0, 32, 512, 512, 64, 64, 0, 0) == 256                     # deferred statistics: one row per workgroup
    assert L.unet_conv_tiles(1, 3, 3, 1, 0, 32, 512, 512, 64, 64, 4, 0) == 32 * 32 * 16             # GroupNorm: per-tile rows
    assert L.unet_conv_wgrad_splits(1, 3, 3, 1, 32, 512, 512, 64, 64) >= 256


def test_state_dict_and_init_match_reference_golden():
    from unet_amd.models import REGISTRY
    z = np.load(os.path.join(G, "unet_f8x16x32x64.npz"))
    torch.manual_seed(42)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16, 32, 64])
    sd = m.state_dict()
    init = [k[5:] for k in z.files if k.startswith("init/")]
    assert list(sd) == init
    for k in sd:
        assert torch.equal(sd[k], torch.from_numpy(z["init/" + k])), k
    assert [n for n, _ in m.named_parameters()][0] == "ups.0.weight"


def test_default_constructor_matches_reference_counts():
    from unet_amd.models import UNet
    m = UNet(in_channels=1, n_classes=2)
    assert sum(p.numel() for p in m.parameters()) == 31042434
    assert len(m.state_dict()) == 136 and len(list(m.parameters())) == 82
    assert UNet().final_conv.out_channels == 3                          # reference default n_classes=3


def test_no_cpu_fallback():
    from unet_amd import UNetHipError
    from unet_amd.models import UNet
    from unet_amd.losses import get_loss
    m = UNet(in_channels=1, n_classes=2, features=[8])
    with pytest.raises(UNetHipError):
        m(torch.zeros(1, 1, 16, 16))
    with pytest.raises(UNetHipError):
        get_loss("ce_dice")(torch.zeros(1, 2, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))
    with pytest.raises(ValueError):
        get_loss("nope")


def test_load_cfg_matches_reference(tmp_path):
    from unet_amd.utils.config_utils import load_cfg
    import yaml
    for fixture in ("cfg_unet.json", "cfg_repunet.json"):
        want = json.load(open(os.path.join(G, fixture)))
        # rebuild the YAML the way PyYAML would have parsed the shipped file: "1e-4" style numbers are strings
        def stringify(o):
            if isinstance(o, dict):
                return {k: stringify(v) for k, v in o.items()}
            if isinstance(o, list):
                return [stringify(v) for v in o]
            if isinstance(o, float) and o < 1e-2:
                return ("%e" % o).replace(".000000", "").replace("e-0", "e-")
            return o
        p = tmp_path / fixture.replace(".json", ".yaml")
        yaml.safe_dump(stringify(want), open(p, "w"))
        got = load_cfg(str(p))
        assert got == want
        assert isinstance(got["optim_args"]["lr"], float) and got["optim_args"]["lr"] == 1e-4
    want = json.load(open(os.path.join(G, "cfg_unet_over.json")))
    base = json.load(open(os.path.join(G, "cfg_unet.json")))
    p = tmp_path / "o.yaml"
    yaml.safe_dump(base, open(p, "w"))
    got = load_cfg(str(p), ["batch_size=32", "optim_args.lr=3e-4", "model_args.features=[8,16]", "exp_name=abc"])
    assert got == want


def test_builders_optim_and_scheduler():
    from unet_amd.utils.builders import get_optim, get_scheduler, build_model_from_cfg
    cfg = json.load(open(os.path.join(G, "cfg_unet.json")))
    cfg["model_args"]["features"] = [8, 16]
    m = build_model_from_cfg(cfg)
    opt = get_optim(cfg["optim"], m.parameters(), **cfg["optim_args"])
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 1e-4 and opt.defaults["weight_decay"] == 1e-4
    sch = get_scheduler(cfg["scheduler"], opt, **cfg["scheduler_args"])
    assert isinstance(sch, torch.optim.lr_scheduler.CosineAnnealingLR) and sch.T_max == 100
    assert get_scheduler(None, opt) is None
    with pytest.raises(ValueError):
        get_optim("lion", m.parameters())


def test_metrics_from_counts_matches_reference_golden():
    from unet_amd.utils.eval_utils import metrics_from_counts
    j = json.load(open(os.path.join(G, "metrics.json")))
    pred, tgt = torch.tensor(j["pred"]), torch.tensor(j["target"])
    K = 2
    counts = [0] * 66
    for p in range(K):
        for t in range(K):
            counts[p * K + t] = int(((pred == p) & (tgt == t)).sum())
    counts[64] = int((pred - tgt).abs().sum())
    got = metrics_from_counts(counts, tgt.numel(), K)
    for k, v in j["metrics"].items():
        assert abs(got[k] - v) < 1e-6, k
