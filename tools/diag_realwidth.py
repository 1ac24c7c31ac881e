"""Per-parameter error table for UNet[64,128,256,512] 2x1x64x64: HIP fp32 / HIP bf16 / CPU-fp32 oracle, each against the
fp64 oracle (the yardstick for how well conditioned each gradient is).  Diagnostic, not a test."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import loss_ref, unet_ref
from unet_amd.losses import get_loss
from unet_amd.models import REGISTRY

H = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.manual_seed(42)
m = REGISTRY["unet"](in_channels=1, n_classes=2, features=[64, 128, 256, 512])
state = {k: v.clone() for k, v in m.state_dict().items()}
g = torch.Generator().manual_seed(42)
x = torch.rand(N, 1, H, H, generator=g)
y = (torch.rand(N, H, H, generator=g) > 0.95).long()


def oracle(dtype):
    st = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in state.items()}
    names = [k for k in st if "running_" not in k and "num_batches" not in k]
    for k in names:
        st[k].requires_grad_(True)
    lg = unet_ref.unet_forward(x.to(dtype), st, True, {})
    loss_ref.ce_dice(lg, y, 1.0, 1.0, 255).backward()
    return lg.detach(), {k: st[k].grad for k in names}


l64, g64 = oracle(torch.float64)
l32, g32 = oracle(torch.float32)
dev = torch.device("cuda:0")
m = m.to(dev).train()
res = {}
for mode in ("fp32", "bf16"):
    m.load_state_dict(state)
    m.compute_dtype = mode
    m.zero_grad(set_to_none=True)
    lg = m(x.to(dev))
    get_loss("ce_dice", ce_weight=1.0, dice_weight=1.0, ignore_index=255)(lg, y.to(dev)).backward()
    res[mode] = (lg.detach().float().cpu(), {k: p.grad.detach().cpu() for k, p in m.named_parameters()})
rm = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))
print(f"logits relmax vs fp64: cpu32 {rm(l32, l64):.2e}  hip32 {rm(res['fp32'][0], l64):.2e}  hipbf16 {rm(res['bf16'][0], l64):.2e}")
print(f"{'param':44s} {'|g|max':>9s} {'cpu32':>9s} {'hip32':>9s} {'hip32vcpu32':>11s} {'hipbf16':>9s}")
for k in g64:
    print(f"{k:44s} {float(g64[k].abs().max()):9.2e} {rm(g32[k], g64[k]):9.2e} {rm(res['fp32'][1][k], g64[k]):9.2e} "
          f"{rm(res['fp32'][1][k], g32[k]):11.2e} {rm(res['bf16'][1][k], g64[k]):9.2e}")
