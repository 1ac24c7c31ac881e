import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import unet_ref, loss_ref
from unet_amd.models import REGISTRY
from unet_amd.losses import get_loss
dev = torch.device("cuda:0")
feats, n, h = [32, 64], 8, 64
def oracle(emu, dtype=torch.float32):
    torch.manual_seed(42)
    m = REGISTRY["unet"](in_channels=1, n_classes=2, features=feats)
    st = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(42)
    x = torch.rand(n, 1, h, h, generator=g); y = (torch.rand(n, h, h, generator=g) > 0.95).long()
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names: st[k].requires_grad_(True)
    rl = unet_ref.unet_forward(x.to(dtype), st, True, {}, emulate_bf16=emu)
    loss_ref.ce_dice(rl, y, 1, 1, 255).backward()
    return m, x, y, {k: st[k].grad.float() for k in names}
m, x, y, g32 = oracle(False)
_, _, _, gemu = oracle(True)
_, _, _, g64 = oracle(False, torch.float64)
m = m.to(dev).train(); m.compute_dtype = "bf16"
lg = m(x.to(dev)); get_loss("ce_dice", ignore_index=255)(lg, y.to(dev)).backward()
r = lambda a, b: float((a - b).norm() / b.norm())
print(f"{'tensor':40s} hip-vs-emu  hip-vs-f32  emu-vs-f32  f32-vs-f64  |g|")
for k, p in m.named_parameters():
    gh = p.grad.cpu()
    print(f"{k:40s} {r(gh, gemu[k]):9.2e}  {r(gh, g32[k]):9.2e}  {r(gemu[k], g32[k]):9.2e}  {r(g32[k], g64[k]):9.2e}  {float(g32[k].norm()):.2e}")
