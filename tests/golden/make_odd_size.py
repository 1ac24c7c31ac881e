"""Golden for the branch of UNet.forward that resizes an upsampled tensor to its skip connection
(/root/reference/models/unet.py:56-57), which the shipped 256/512/1024 inputs never take: the REAL reference
UNet(features=[8,16,32]) and ReparamUNet(features=[8,16,32], kernel_sizes=[3]) on 2x1x36x44 (36 -> 18 -> 9 -> 4 and
44 -> 22 -> 11 -> 5: the two deepest decoder levels resize), train-mode forward, CE+Dice, backward.
Run in the build container only (imports /root/reference; nothing is copied):  python tests/golden/make_odd_size.py
-> unet_odd_36x44.npz, repunet_odd_36x44.npz (inputs, initial state, logits, loss, every gradient, BN buffers)."""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from losses import get_loss                      # noqa: E402
from models import REGISTRY                      # noqa: E402


def capture(name, key, kwargs):
    torch.manual_seed(42)
    m = REGISTRY[key](**kwargs)
    m.train()
    init = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 1, 36, 44, generator=g)
    y = (torch.rand(2, 36, 44, generator=g) > 0.9).long()
    logits = m(x)
    loss = get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(logits, y)
    loss.backward()
    out = {"x": x.numpy(), "y": y.numpy(), "logits": logits.detach().numpy(), "loss": loss.detach().numpy()}
    for k, p in m.named_parameters():
        out["grad/" + k] = p.grad.numpy()
    for k, b in m.named_buffers():
        out["buf1/" + k] = b.detach().numpy()
    for k, v in init.items():
        out["init/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, float(loss), tuple(logits.shape))


capture("unet_odd_36x44", "unet", dict(in_channels=1, n_classes=2, features=[8, 16, 32]))
capture("repunet_odd_36x44", "reparam_unet", dict(in_channels=1, n_classes=2, features=[8, 16, 32], kernel_sizes=[3]))
