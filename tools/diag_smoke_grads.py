#!/usr/bin/env python3
"""Diagnostic: per-tensor gradient error of the smoke configuration against the oracle."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import loss_ref, unet_ref
from unet_amd.losses import get_loss
from unet_amd.models import REGISTRY

dev = torch.device("cuda:0")
torch.manual_seed(42)
model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32])
state = {k: v.clone() for k, v in model.state_dict().items()}
g = torch.Generator().manual_seed(42)
x = torch.rand(2, 1, 64, 64, generator=g)
y = (torch.rand(2, 64, 64, generator=g) > 0.95).long()
model = model.to(dev).train()
loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
logits = model(x.to(dev))
loss_fn(logits, y.to(dev)).backward()
names = [k for k in state if "running_" not in k and "num_batches" not in k]
for k in names:
    state[k].requires_grad_(True)
nb = {}
ref_logits = unet_ref.unet_forward(x, state, train=True, new_buffers=nb)
loss_ref.ce_dice(ref_logits, y, 1.0, 1.0, 255).backward()
for k, p in model.named_parameters():
    r = state[k].grad
    e = float((p.grad.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-30))
    if e > 2e-5:
        print(f"{k:45s} err {e:.2e}  max|ref| {float(r.abs().max()):.3e}")
for k, b in model.named_buffers():
    if k in nb and "num_batches" not in k:
        e = float((b.cpu() - nb[k]).abs().max() / nb[k].abs().max())
        if e > 1e-6:
            print(f"buffer {k:40s} err {e:.2e}")
print("done")
