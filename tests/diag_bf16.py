"""Prints per-tensor bf16-vs-oracle(fp32) errors for a few model sizes (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_ref, loss_ref
from unet_amd.models import REGISTRY
from unet_amd.losses import get_loss

dev = torch.device("cuda:0")
for feats, n, h in (([8, 16], 2, 32), ([16, 32], 4, 64), ([16, 32, 64], 4, 128), ([32, 64], 8, 64)):
    torch.manual_seed(42)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=feats)
    st = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    x = torch.rand(n, 1, h, h, generator=g); y = (torch.rand(n, h, h, generator=g) > 0.95).long()
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names: st[k].requires_grad_(True)
    emu = len(sys.argv) > 1
    rl = unet_ref.unet_forward(x, st, True, {}, emulate_bf16=emu)
    loss_ref.ce_dice(rl, y, 1, 1, 255).backward()
    m = m.to(dev).train()
    for mode in ("fp32", "bf16"):
        m.zero_grad()
        m.compute_dtype = mode
        lg = m(x.to(dev)); loss = get_loss("ce_dice", ignore_index=255)(lg, y.to(dev)); loss.backward()
        e = float((lg.detach().cpu() - rl.detach()).norm() / rl.detach().norm())
        worst = []
        for k, p in m.named_parameters():
            if k.endswith("double_conv.0.bias") or k.endswith("double_conv.3.bias"): continue
            worst.append((float((p.grad.cpu() - st[k].grad).norm() / st[k].grad.norm()), k))
        worst.sort(reverse=True)
        agree = float((lg.argmax(1).cpu() == rl.argmax(1)).float().mean())
        print(feats, n, h, mode, f"logits relL2 {e:.2e} argmax agree {agree:.4f} worst grads:", [(f"{a:.2e}", b) for a, b in worst[:3]], flush=True)
