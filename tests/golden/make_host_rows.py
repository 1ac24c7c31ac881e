#!/usr/bin/env python3
"""Golden fixtures for the host-side rows of SURVEY 8 (S2 Trainer.fit/evaluate, (f)3 dataset val path, (f)4 checkpoints),
produced by importing and RUNNING the real reference in the build container:

    python tests/golden/make_host_rows.py

`loguru` and `torchvision` are not installed here; the reference modules that import them get inert stand-ins in
sys.modules for the import only (SURVEY 8(c) measured that this works; none of the captured paths calls into them).
No reference source is copied: the modules are imported from /root/reference and called.  Outputs (all data):

  trainer_fit.npz / trainer_fit.json   reference Trainer(...).fit(train_loader, val_loader, epochs=3) on a tiny TensorDataset
        (UNet features [8,16], 14 train / 5 val samples of 1x32x32, ragged last batches, AdamW + cosine schedule, clip 1.0):
        every on_batch_end (step, main_loss, lr), every on_epoch_end tuple (epoch, train_loss, train_acc, val_loss,
        main_score) with trainer.latest_val_stats and the lr after the scheduler step, the final Trainer.evaluate(), and
        the model state after training  (/root/reference/trainer.py:66-173, utils/eval_utils.py:59-145)
  ref_last.ckpt                        the `last.ckpt` the reference's save_checkpoint wrote for that trainer after fit
        (/root/reference/utils/checkpoint_utils.py:15-48); loaded in the tests with torch.load(weights_only=True)
  trainer_fit.json["reverse"]          the other direction, checked HERE because it needs the reference: a checkpoint
        written by unet_amd.utils.checkpoint_utils.save_checkpoint is read by the reference's load_checkpoint (:50-89) and
        by the reference's utils/io.py load_checkpoint (:19-45) into reference models, bit-exact, optimizer and scheduler
        state included
  ct_val.npz                           reference CT_NpyDataset(split="val") items for three raw slices on disk: 32x32 with a
        {0,255} uint8 mask (no resize), 40x24 and 20x52 (bilinear image / nearest mask resize to 32), incl. HU values
        outside the window  (/root/reference/data/ct_dataset.py:25-47,82)
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))


class _Log:
    def debug(self, *a, **k):
        pass

    info = warning = error = exception = success = debug

    def bind(self, *a, **k):
        return self

    def add(self, *a, **k):
        return 0

    def remove(self, *a, **k):
        pass


def install_stubs():
    lg = types.ModuleType("loguru")
    lg.logger = _Log()
    sys.modules.setdefault("loguru", lg)
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]


def fit_data():
    """The tiny dataset of the Trainer golden; tests rebuild it from the stored arrays."""
    g = torch.Generator().manual_seed(2024)
    xt = torch.rand(14, 1, 32, 32, generator=g)
    xv = torch.rand(5, 1, 32, 32, generator=g)
    mask = lambda x: (F.avg_pool2d(x, 5, 1, 2) > 0.53).long()       # a learnable target; (N,1,H,W): step_batch squeezes it
    return xt, mask(xt), xv, mask(xv)


CFG = {"metrics": ["acc_pixel", "dice_class_1", "dice_macro", "mae", "sen_class_1", "spec_class_1"],
       "early_stop": {"monitor": "val/dice_class_1"}, "model_args": {"n_classes": 2}, "scheduler_step": "epoch",
       "model": "unet", "epochs": 3}


class Recorder:
    def __init__(self):
        self.batches, self.epochs, self.begun, self.ended = [], [], 0, 0

    def on_train_begin(self, trainer):
        self.begun += 1

    def on_batch_end(self, trainer, step_idx, info):
        self.batches.append([int(step_idx), float(info["main_loss"]), float(info["lr"])])

    def on_epoch_end(self, trainer, epoch, train_loss, train_acc, val_loss, main_score):
        self.epochs.append({"epoch": int(epoch), "train_loss": float(train_loss), "train_acc": float(train_acc),
                            "val_loss": float(val_loss), "main_score": float(main_score),
                            "latest_val_stats": {k: float(v) for k, v in trainer.latest_val_stats.items()},
                            "lr_after": float(trainer.optimizer.param_groups[0]["lr"])})

    def on_train_end(self, trainer):
        self.ended += 1


def trainer_case():
    from torch.utils.data import DataLoader, TensorDataset
    from losses import get_loss
    from models import REGISTRY
    from trainer import Trainer
    from utils.checkpoint_utils import save_checkpoint

    xt, yt, xv, yv = fit_data()
    torch.manual_seed(42)
    model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=3, eta_min=1e-5)
    rec = Recorder()
    tr = Trainer(model, get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255), opt, scheduler=sched,
                 amp=False, grad_clip=1.0, device="cpu", callbacks=[rec], progress_bar=False, cfg=dict(CFG), logger=_Log())
    train_loader = DataLoader(TensorDataset(xt, yt), batch_size=4, shuffle=False)       # batches of 4, 4, 4 and 2
    val_loader = DataLoader(TensorDataset(xv, yv), batch_size=2, shuffle=False)         # 2, 2, 1
    tr.fit(train_loader, val_loader, epochs=3)
    final = tr.evaluate(val_loader, epoch=99)
    arrays = {"xt": xt.numpy(), "yt": yt.numpy(), "xv": xv.numpy(), "yv": yv.numpy()}
    for k, v in model.state_dict().items():
        arrays["final/" + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "trainer_fit.npz"), **arrays)
    with tempfile.TemporaryDirectory() as td:
        save_checkpoint(os.path.join(td, "run"), tr, epoch=3, is_best=False)
        with open(os.path.join(td, "run", "last.ckpt"), "rb") as f, open(os.path.join(OUT, "ref_last.ckpt"), "wb") as o:
            o.write(f.read())
    info = {"cfg": CFG, "train_begin_calls": rec.begun, "train_end_calls": rec.ended, "global_step": int(tr._global_step),
            "batches": rec.batches, "epochs": rec.epochs,
            "final_evaluate": {"val_loss": float(final[0]), "main_score": float(final[1]),
                               "stats": {k: float(v) for k, v in final[2].items()}},
            "model_training_after_evaluate": bool(model.training)}
    return info, tr


def reverse_direction(ref_trainer):
    """A checkpoint written by unet_amd is read by the reference's two loaders."""
    sys.path.insert(0, ROOT)
    from unet_amd.models import REGISTRY as MY_REG
    from unet_amd.utils import checkpoint_utils as cu
    from models import REGISTRY
    from utils.checkpoint_utils import load_checkpoint as ref_load
    import importlib
    ref_io = importlib.import_module("utils.io")

    torch.manual_seed(7)
    mine = MY_REG["unet"](in_channels=1, n_classes=2, features=[8, 16])
    with torch.no_grad():
        for b in mine.buffers():
            if b.is_floating_point():
                b.add_(torch.rand_like(b))
    opt = torch.optim.AdamW(mine.parameters(), lr=3e-4, weight_decay=1e-4)
    for p in mine.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=7)
    sched.step()
    tr = types.SimpleNamespace(model=mine, optimizer=opt, scheduler=sched, scaler=torch.amp.GradScaler(enabled=False),
                               latest_val_stats={"dice_class_1": 0.25}, cfg={"model": "unet"})
    res = {}
    with tempfile.TemporaryDirectory() as td:
        cu.save_checkpoint(os.path.join(td, "run"), tr, epoch=5, is_best=True)
        for name in ("last.ckpt", "best.ckpt"):
            torch.manual_seed(1)
            theirs = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])
            opt2 = torch.optim.AdamW(theirs.parameters(), lr=1.0)
            sched2 = torch.optim.lr_scheduler.CosineAnnealingLR(opt2, T_max=7)
            out = ref_load(os.path.join(td, "run", name), theirs, optimizer=opt2, scheduler=sched2, map_location="cpu", strict=True)
            same_w = all(torch.equal(a, b) for a, b in zip(mine.state_dict().values(), theirs.state_dict().values()))
            same_keys = list(mine.state_dict()) == list(theirs.state_dict())
            s1, s2 = opt.state_dict(), opt2.state_dict()
            same_opt = s1["param_groups"] == s2["param_groups"] and all(
                torch.equal(s1["state"][i][k], s2["state"][i][k]) for i in s1["state"] for k in s1["state"][i])
            same_sched = sched.state_dict() == sched2.state_dict()
            res[name] = {"start_epoch": out["start_epoch"], "weights_equal": same_w, "keys_equal": same_keys,
                         "optimizer_equal": same_opt, "scheduler_equal": same_sched}
            assert out["start_epoch"] == 6 and same_w and same_keys and same_opt and same_sched, res
        torch.manual_seed(2)
        theirs = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])
        ref_io.load_checkpoint(os.path.join(td, "run", "last.ckpt"), theirs, device="cpu")
        res["utils.io"] = {"weights_equal": all(torch.equal(a, b) for a, b in zip(mine.state_dict().values(), theirs.state_dict().values()))}
        assert res["utils.io"]["weights_equal"]
    return res


def ct_val_case():
    import data as ref_data          # noqa: F401  (registers ct_segmentation)
    from data.ct_dataset import CT_NpyDataset

    g = np.random.default_rng(77)
    S = 32
    shapes = [(32, 32), (40, 24), (20, 52)]
    arrays = {"image_size": np.int64(S)}
    with tempfile.TemporaryDirectory() as td:
        for d in ("val/images", "val/masks", "train/images", "train/masks"):
            os.makedirs(os.path.join(td, d))
        for i, (h, w) in enumerate(shapes):
            img = (g.random((h, w), dtype=np.float32) * 3000.0 - 1800.0).astype(np.float32)      # HU, some outside [-1000, 500]
            mask = ((g.random((h, w)) > 0.7).astype(np.uint8) * np.uint8(255 if i == 0 else 1))
            if i == 2:
                mask[3:6, 10:20] = 7                                                             # artefact labels: > 0 -> 1
            name = f"p{i:02d}_slice_{i:03d}.npy"
            np.save(os.path.join(td, "val/images", name), img)
            np.save(os.path.join(td, "val/masks", name), mask)
            arrays[f"raw_image/{i}"], arrays[f"raw_mask/{i}"] = img, mask
        ds = CT_NpyDataset(td, split="val", image_size=S)
        assert len(ds) == len(shapes) and ds.augment is False
        for i in range(len(ds)):
            x, y = ds[i]
            assert x.dtype == torch.float32 and y.dtype == torch.int64 and tuple(x.shape) == (1, S, S) and tuple(y.shape) == (S, S)
            arrays[f"image/{i}"], arrays[f"mask/{i}"] = x.numpy(), y.numpy()
    np.savez_compressed(os.path.join(OUT, "ct_val.npz"), **arrays)
    return {"slices": [list(s) for s in shapes]}


def main():
    if not os.path.isdir(REF):
        print("reference not present; fixtures are committed, nothing to do")
        return
    install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    torch.use_deterministic_algorithms(True)
    info, tr = trainer_case()
    info["reverse"] = reverse_direction(tr)
    info["ct_val"] = ct_val_case()
    info["torch"] = torch.__version__
    json.dump(info, open(os.path.join(OUT, "trainer_fit.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(info["epochs"], indent=1))
    print("reverse:", info["reverse"])


if __name__ == "__main__":
    main()
