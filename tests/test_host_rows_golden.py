"""Host-side rows of SURVEY 8 against fixtures produced by RUNNING the reference (tests/golden/make_host_rows.py):

  S2    Trainer.fit / Trainer.evaluate  (/root/reference/trainer.py:66-173, utils/eval_utils.py:59-145)  -> trainer_fit.{json,npz}
  (f)4  checkpoint interop              (/root/reference/utils/checkpoint_utils.py:15-89, utils/io.py:19-45) -> ref_last.ckpt; the
        reverse direction (our file -> the reference's loaders) needs the reference and is asserted inside the generator, its
        verdict is recorded in trainer_fit.json["reverse"]
  (f)3  dataset val path                (/root/reference/data/ct_dataset.py:25-47,82)                    -> ct_val.npz

CPU tests: the checkpoint file and the oracle's val path.  GPU tests (marked): the Trainer on the HIP engine, evaluate() on the
reference-trained weights, the device input pipeline on the reference's slices.
Flips / rotation of the TRAIN path go through torchvision, which is not installed anywhere this project runs: that part stays
"parity unpinned" (tests/test_ct_pipeline.py checks it against the oracle's restatement only).
"""
import json
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


def _info():
    with open(os.path.join(G, "trainer_fit.json")) as f:
        return json.load(f)


def _arrays(name):
    z = np.load(os.path.join(G, name))
    return {k: z[k] for k in z.files}


def _small_unet():
    from unet_amd.models import REGISTRY
    torch.manual_seed(42)
    return REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])


# ------------------------------------------------------------------------------------------------ (f)4 checkpoints, CPU

def test_reference_written_checkpoint_loads(tmp_path):
    """A `last.ckpt` written by the reference's save_checkpoint after its Trainer.fit: both of our loaders restore the
    model bit for bit (weights_only=True: nothing in the file is executed), optimizer and scheduler state, and the epoch."""
    from unet_amd.utils import checkpoint_utils as cu
    from unet_amd.utils import io as uio
    from unet_amd.utils.builders import get_optim, get_scheduler
    path = os.path.join(G, "ref_last.ckpt")
    info, final = _info(), {k[6:]: torch.from_numpy(v) for k, v in _arrays("trainer_fit.npz").items() if k.startswith("final/")}
    raw = torch.load(path, map_location="cpu", weights_only=True)
    assert set(raw) == {"epoch", "model_state", "optim_state", "scheduler_state", "scaler_state", "trainer_backbone_frozen",
                        "latest_val_stats", "cfg", "model", "opt", "sched"}          # checkpoint_utils.py:30-44
    m = _small_unet()
    assert list(raw["model_state"]) == list(m.state_dict())
    opt = get_optim("adamw", m.parameters(), lr=1.0, weight_decay=0.0)
    sched = get_scheduler("cosine", opt, T_max=3, eta_min=1e-5)
    out = cu.load_checkpoint(path, m, optimizer=opt, scheduler=sched, map_location="cpu", strict=True)
    assert out["start_epoch"] == 4
    for k, v in m.state_dict().items():
        assert torch.equal(v, final[k]), k
    g = opt.param_groups[0]
    assert g["weight_decay"] == 1e-4 and g["initial_lr"] == 3e-3 and abs(g["lr"] - info["epochs"][-1]["lr_after"]) < 1e-12
    assert sched.last_epoch == 3
    st = opt.state_dict()["state"]
    assert len(st) == len(list(m.parameters())) and all(float(s["step"]) == info["global_step"] for s in st.values())
    assert all(torch.equal(st[i]["exp_avg"], raw["optim_state"]["state"][i]["exp_avg"]) for i in st)
    assert out["raw"]["cfg"] == info["cfg"]
    assert out["raw"]["latest_val_stats"] == pytest.approx(info["epochs"][-1]["latest_val_stats"])
    # utils/io.py call shape; a `module.`-prefixed copy of the same file (saved from a wrapped model) loads as well
    m2 = _small_unet()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    assert uio.load_checkpoint(path, m2, device="cpu") is m2
    assert all(torch.equal(a, final[k]) for k, a in m2.state_dict().items())
    wrapped = dict(raw, model_state={"module." + k: v for k, v in raw["model_state"].items()})
    del wrapped["model"]
    torch.save(wrapped, tmp_path / "wrapped.ckpt")
    m3 = _small_unet()
    cu.load_checkpoint(str(tmp_path / "wrapped.ckpt"), m3, map_location="cpu", strict=True)
    assert all(torch.equal(a, final[k]) for k, a in m3.state_dict().items())


def test_our_checkpoint_was_read_by_the_reference():
    """Recorded by the generator (it needs the reference): a file from unet_amd's save_checkpoint went through the reference's
    load_checkpoint (strict=True, optimizer + scheduler) and through utils/io.py's loader, everything bit-equal."""
    rev = _info()["reverse"]
    for name in ("last.ckpt", "best.ckpt"):
        assert rev[name] == {"start_epoch": 6, "weights_equal": True, "keys_equal": True, "optimizer_equal": True,
                             "scheduler_equal": True}
    assert rev["utils.io"]["weights_equal"] is True


def test_our_writer_matches_the_reference_file_layout(tmp_path):
    """Same trainer state written by our save_checkpoint: same keys, same nesting, same tensors as the reference's file."""
    import types
    from unet_amd.utils import checkpoint_utils as cu
    from unet_amd.utils.builders import get_optim, get_scheduler
    ref = torch.load(os.path.join(G, "ref_last.ckpt"), map_location="cpu", weights_only=True)
    m = _small_unet()
    opt = get_optim("adamw", m.parameters(), lr=1.0)
    sched = get_scheduler("cosine", opt, T_max=3, eta_min=1e-5)
    cu.load_checkpoint(os.path.join(G, "ref_last.ckpt"), m, optimizer=opt, scheduler=sched, map_location="cpu", strict=True)
    tr = types.SimpleNamespace(model=m, optimizer=opt, scheduler=sched, scaler=torch.amp.GradScaler(enabled=False),
                               latest_val_stats=ref["latest_val_stats"], cfg=ref["cfg"])
    cu.save_checkpoint(str(tmp_path / "run"), tr, epoch=3)
    mine = torch.load(tmp_path / "run" / "last.ckpt", map_location="cpu", weights_only=True)

    def same(a, b, where):
        assert type(a) is type(b) or (isinstance(a, (int, float)) and isinstance(b, (int, float))), where
        if isinstance(a, dict):
            assert list(a) == list(b), where
            for k in a:
                same(a[k], b[k], f"{where}.{k}")
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b), where
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{where}[{i}]")
        elif isinstance(a, torch.Tensor):
            assert a.dtype == b.dtype and torch.equal(a, b), where
        else:
            assert a == b, where

    strip = lambda d: {k: ({kk: vv for kk, vv in v.items() if kk not in ("fused", "foreach", "capturable", "differentiable",
                                                                         "maximize", "amsgrad", "decoupled_weight_decay")}
                           if isinstance(v, dict) else v) for k, v in d.items()}
    assert list(mine) == list(ref)
    for k in ref:
        if k in ("optim_state", "opt"):       # hyper-parameter flags a torch version adds to param_groups are not part of the contract
            same(ref[k]["state"], mine[k]["state"], k + ".state")
            for a, b in zip(ref[k]["param_groups"], mine[k]["param_groups"]):
                assert strip({"g": a})["g"] == strip({"g": b})["g"], k
        else:
            same(ref[k], mine[k], k)


# ------------------------------------------------------------------------------------------------ (f)3 val path

def _ct_cases():
    z = _arrays("ct_val.npz")
    S = int(z["image_size"])
    n = sum(1 for k in z if k.startswith("raw_image/"))
    return S, [(z[f"raw_image/{i}"], z[f"raw_mask/{i}"], z[f"image/{i}"], z[f"mask/{i}"]) for i in range(n)]


def test_oracle_val_path_equals_reference_dataset():
    """oracle/ct_pipeline_ref.py (no augmentation = the val split) against CT_NpyDataset(split='val').__getitem__ of the
    reference: window, bilinear / nearest resize from non-square sizes, {0,255} and artefact labels -> {0,1}; bit-exact."""
    from oracle import ct_pipeline_ref as ref
    S, cases = _ct_cases()
    assert len(cases) == 3
    for img, mask, want_x, want_y in cases:
        x, y = ref.ct_transform(img, mask, S)
        assert x.dtype == torch.float32 and y.dtype == torch.int64
        assert np.array_equal(x.numpy(), want_x) and np.array_equal(y.numpy(), want_y)
    assert set(np.unique(cases[0][1]).tolist()) == {0, 255} and set(np.unique(cases[0][3]).tolist()) == {0, 1}


@pytest.mark.gpu
def test_device_pipeline_val_path_equals_reference_dataset():
    """unet_ct_preprocess through DeviceCTPipeline on the reference's slices: image within 2e-6 absolute of the reference's
    (fp32 bilinear weights; the kernel may contract to FMA), mask bit-exact."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from unet_amd.data import DeviceCTPipeline
    S, cases = _ct_cases()
    for img, mask, want_x, want_y in cases:
        x, y = DeviceCTPipeline(S).transform(torch.from_numpy(img)[None], torch.from_numpy(mask)[None], None, torch.device("cuda:0"))
        assert tuple(x.shape) == (1, 1, S, S) and x.dtype == torch.float32 and y.dtype == torch.int64
        assert float((x[0].cpu() - torch.from_numpy(want_x)).abs().max()) <= 2e-6
        assert torch.equal(y[0].cpu(), torch.from_numpy(want_y))


# ------------------------------------------------------------------------------------------------ S2 Trainer, GPU

class _Recorder:
    def __init__(self):
        self.batches, self.epochs, self.begun, self.ended = [], [], 0, 0

    def on_train_begin(self, trainer):
        self.begun += 1

    def on_batch_end(self, trainer, step_idx, info):
        self.batches.append([int(step_idx), float(info["main_loss"]), float(info["lr"])])

    def on_epoch_end(self, trainer, epoch, train_loss, train_acc, val_loss, main_score):
        self.epochs.append({"epoch": int(epoch), "train_loss": float(train_loss), "train_acc": float(train_acc),
                            "val_loss": float(val_loss), "main_score": float(main_score),
                            "latest_val_stats": dict(trainer.latest_val_stats),
                            "lr_after": float(trainer.optimizer.param_groups[0]["lr"])})

    def on_train_end(self, trainer):
        self.ended += 1


def _loaders(dev):
    from torch.utils.data import DataLoader, TensorDataset
    z = _arrays("trainer_fit.npz")
    t = lambda k: torch.from_numpy(z[k])
    return (DataLoader(TensorDataset(t("xt"), t("yt")), batch_size=4, shuffle=False),
            DataLoader(TensorDataset(t("xv"), t("yv")), batch_size=2, shuffle=False))


@pytest.mark.gpu
def test_trainer_fit_matches_reference_trainer():
    """unet_amd.Trainer.fit on the HIP engine (fp32 kernels, `amp: false` as both shipped configs) against what the
    reference's Trainer produced on the same data, init, optimizer and schedule: the same callback protocol (one
    on_train_begin / on_train_end, 12 on_batch_end with step 1..12, 3 on_epoch_end), epoch train loss with the reference's
    divide-by-len(dataset) (trainer.py:133; ragged last batch), pixel accuracy (:115-126), val loss / main score /
    latest_val_stats (eval_utils.py), the cosine schedule stepped per epoch.
    Tolerance: twelve optimizer steps at lr 3e-3 separate two fp32 evaluations by more than one step does: losses 2e-3
    relative, accuracies / metrics 5e-3 absolute (a handful of the 14 336 / 5 120 pixels), learning rates exact."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import logging
    from unet_amd.losses import get_loss
    from unet_amd.trainer import Trainer
    from unet_amd.utils.builders import get_optim, get_scheduler
    info = _info()
    dev = torch.device("cuda:0")
    model = _small_unet().to(dev)
    opt = get_optim("adamw", model.parameters(), lr=3e-3, weight_decay=1e-4)
    sched = get_scheduler("cosine", opt, T_max=3, eta_min=1e-5)
    rec = _Recorder()
    tr = Trainer(model, get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255), opt, scheduler=sched, amp=False,
                 grad_clip=1.0, device=dev, callbacks=[rec], progress_bar=False, cfg=dict(info["cfg"]), logger=logging.getLogger("t"))
    train_loader, val_loader = _loaders(dev)
    tr.fit(train_loader, val_loader, epochs=3)
    assert (rec.begun, rec.ended, tr._global_step) == (info["train_begin_calls"], info["train_end_calls"], info["global_step"])
    assert [b[0] for b in rec.batches] == [b[0] for b in info["batches"]]
    worst = 0.0
    for mine, ref in zip(rec.batches, info["batches"]):
        assert mine[2] == pytest.approx(ref[2], rel=1e-12)                       # lr seen by the step
        worst = max(worst, abs(mine[1] - ref[1]) / abs(ref[1]))
        assert abs(mine[1] - ref[1]) <= 2e-3 * abs(ref[1]), (mine, ref)
    assert len(rec.epochs) == len(info["epochs"]) == 3
    for mine, ref in zip(rec.epochs, info["epochs"]):
        assert mine["epoch"] == ref["epoch"] and mine["lr_after"] == pytest.approx(ref["lr_after"], rel=1e-12)
        assert abs(mine["train_loss"] - ref["train_loss"]) <= 2e-3 * ref["train_loss"]
        assert abs(mine["train_acc"] - ref["train_acc"]) <= 5e-3
        assert abs(mine["val_loss"] - ref["val_loss"]) <= 2e-3 * ref["val_loss"]
        assert abs(mine["main_score"] - ref["main_score"]) <= 5e-3
        assert set(mine["latest_val_stats"]) == set(ref["latest_val_stats"])
        for k, v in ref["latest_val_stats"].items():
            tol = 2e-3 * abs(v) if k == "val_loss" else 5e-3
            assert abs(mine["latest_val_stats"][k] - v) <= tol, (k, mine["latest_val_stats"][k], v)
    print(f"Trainer.fit vs the reference's Trainer: worst per-step loss gap {worst:.2e} relative; epochs "
          + "; ".join(f"{m['train_loss']:.5f}/{r['train_loss']:.5f} acc {m['train_acc']:.4f}/{r['train_acc']:.4f} val {m['val_loss']:.5f}/{r['val_loss']:.5f}"
                      for m, r in zip(rec.epochs, info["epochs"])))
    assert model.training == info["model_training_after_evaluate"]


@pytest.mark.gpu
def test_evaluate_on_reference_trained_weights():
    """Trainer.evaluate with the weights the REFERENCE's fit left behind (loaded from its own last.ckpt): no training drift
    between the two sides, so the comparison is tight -- val loss 1e-5 relative, every metric 1e-6 absolute, fp32 kernels."""
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    import logging
    from unet_amd.losses import get_loss
    from unet_amd.trainer import Trainer
    from unet_amd.utils import checkpoint_utils as cu
    info = _info()
    dev = torch.device("cuda:0")
    model = _small_unet()
    cu.load_checkpoint(os.path.join(G, "ref_last.ckpt"), model, map_location="cpu", strict=True)
    model = model.to(dev).train()
    tr = Trainer(model, get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255), None, amp=False, device=dev,
                 progress_bar=False, cfg=dict(info["cfg"]), logger=logging.getLogger("t"))
    val_loss, main_score, stats = tr.evaluate(_loaders(dev)[1], epoch=99)
    ref = info["final_evaluate"]
    assert abs(val_loss - ref["val_loss"]) <= 1e-5 * ref["val_loss"]
    assert set(stats) == set(ref["stats"])
    for k, v in ref["stats"].items():
        assert abs(stats[k] - v) <= (1e-5 * abs(v) if k == "val_loss" else 1e-6), (k, stats[k], v)
    assert abs(main_score - ref["main_score"]) <= 1e-6 and model.training
