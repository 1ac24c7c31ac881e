"""How many ReLU-mask / pool-argmax decisions differ between the HIP fp32 path and the CPU oracle (UNet[64..512], 2x1x64x64)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import loss_ref, unet_ref
from unet_amd.losses import get_loss
from unet_amd.models import REGISTRY
from unet_amd import ops

torch.manual_seed(42)
m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512])
state = {k: v.clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(42)
x = torch.rand(2, 1, 64, 64, generator=g)
y = (torch.rand(2, 64, 64, generator=g) > 0.95).long()
dev = torch.device("cuda:0")
m = m.to(dev).train()
m.compute_dtype = "fp32"
m._engine.keep_saved = True
lg = m(x.to(dev))
get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(lg, y.to(dev)).backward()
sv = m._engine.last_saved
# CPU activations
acts = {}
st = state
def dc(t, p):
    t = F.conv2d(t, st[p + ".double_conv.0.weight"], st[p + ".double_conv.0.bias"], padding=1)
    t = F.relu(F.batch_norm(t, None, None, st[p + ".double_conv.1.weight"], st[p + ".double_conv.1.bias"], True, 0.1, 1e-5))
    acts[p + ".a"] = t
    t = F.conv2d(t, st[p + ".double_conv.3.weight"], st[p + ".double_conv.3.bias"], padding=1)
    t = F.relu(F.batch_norm(t, None, None, st[p + ".double_conv.4.weight"], st[p + ".double_conv.4.bias"], True, 0.1, 1e-5))
    acts[p + ".b"] = t
    return t
t = x; skips = []
for i in range(4):
    t = dc(t, f"downs.{i}"); skips.append(t)
    t, idx = F.max_pool2d(t, 2, 2, return_indices=True); acts[f"pool{i}"] = idx
t = dc(t, "bottleneck")
for j in range(4):
    t = F.conv_transpose2d(t, st[f"ups.{2*j}.weight"], st[f"ups.{2*j}.bias"], stride=2)
    t = torch.cat((skips[3 - j], t), 1)
    t = dc(t, f"ups.{2*j+1}")
def nchw(a):
    return a.buf[..., a.off:a.off + a.C].float().cpu().permute(0, 3, 1, 2)
for p, (r1, r2) in sv.blocks.items():
    ya = nchw(r2.xin)
    ref = acts[p + ".a"]
    mism = (ya > 0) != (ref > 0)
    print(f"{p}.a: {int(mism.sum())} mask flips of {mism.numel()}, max |y| at a flip {float(torch.maximum(ya.abs(), ref.abs())[mism].max()) if mism.any() else 0:.2e}")
for i in range(4):
    f = [64, 128, 256, 512][i]
    ya = sv.cats[i][..., :f].float().cpu().permute(0, 3, 1, 2)
    ref = acts[f"downs.{i}.b"]
    mism = (ya > 0) != (ref > 0)
    idx = ops.pool_indices_int64(sv.codes[i], ya.shape[3]).cpu()
    pm = idx != acts[f"pool{i}"]
    print(f"downs.{i}.b: {int(mism.sum())} mask flips of {mism.numel()}; pool argmax flips {int(pm.sum())} of {pm.numel()}")
