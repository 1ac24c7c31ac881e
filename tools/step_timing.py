#!/usr/bin/env python3
"""Diagnostic: per-step wall time of the bench workload (step_batch ends with its own host sync)."""
import gc, logging, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unet_amd.losses import get_loss
from unet_amd.models import REGISTRY
from unet_amd.utils.builders import get_optim
from unet_amd.utils.train_step import step_batch

nogc = "--nogc" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512]).to(dev).train()
opt = get_optim("adamw", [p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
tr = types.SimpleNamespace(optimizer=opt, amp=True, model=model, grad_clip=1.0, cbs=[], current_epoch=1,
                           logger=logging.getLogger("b"), loss_fn=get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255))
sc = torch.amp.GradScaler(enabled=False)
g = torch.Generator().manual_seed(42)
x = torch.rand(32, 1, 512, 512, generator=g).to(dev)
y = (torch.rand(32, 512, 512, generator=g) > 0.95).long().to(dev)
if nogc:
    gc.disable()
if "--overlap" in sys.argv:            # A/B: weight gradients on the side stream
    model._engine.overlap_wgrad = True
if "--late-pack" in sys.argv:          # A/B: weights re-packed at the start of the next forward instead
    model.prepack_weights = None
ts = []
for i in range(24):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step_batch(tr, x, y, sc, batch_idx=i + 1)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(sys.argv[1:]) or "default", f"mean {sum(ts[4:]) / len(ts[4:]):.2f} |", " ".join(f"{t:.1f}" for t in ts))
