#!/usr/bin/env python3
"""Diagnostic: dump logits / grads / buffers of the smoke configuration to gpurun_out/<tag>.pt (A/B between library builds)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from unet_amd.losses import get_loss
from unet_amd.models import REGISTRY
tag = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(42)
model = REGISTRY["unet"](in_channels=1, n_classes=2, features=[16, 32])
g = torch.Generator().manual_seed(42)
x = torch.rand(2, 1, 64, 64, generator=g)
y = (torch.rand(2, 64, 64, generator=g) > 0.95).long()
model = model.to(dev).train()
loss_fn = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)
logits = model(x.to(dev))
loss_fn(logits, y.to(dev)).backward()
out = {"logits": logits.detach().cpu()}
for k, p in model.named_parameters():
    out["grad/" + k] = p.grad.cpu()
for k, b in model.named_buffers():
    out["buf/" + k] = b.cpu()
torch.save(out, f"gpurun_out/{tag}.pt")
print("saved", tag)
