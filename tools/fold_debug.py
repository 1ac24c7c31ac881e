import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unet_amd import ops
from unet_amd.models import REGISTRY
dev = torch.device("cuda:0")
z = np.load("tests/golden/fold_k3.npz")
g = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
pre = {k[4:]: v for k, v in g.items() if k.startswith("pre/")}
post = {k[5:]: v for k, v in g.items() if k.startswith("post/")}
m = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=[3], norm="batch")
m.load_state_dict(pre); m = m.to(dev).eval(); m.switch_to_deploy()
sd = m.state_dict()
for k in post:
    a, b = sd[k].cpu(), post[k]
    nd = int((a != b).sum())
    if nd:
        d = (a - b).abs().max()
        print(k, tuple(a.shape), "mismatch", nd, "of", a.numel(), "max abs", float(d), "max rel", float(((a-b).abs()/b.abs().clamp_min(1e-30)).max()))
# elementary ops: sqrt, div, mul against CPU
p = "ups.3.reparam_conv.2.conv_list.0"
for bn in ("bnSq", "bnVe1", "bnHo1"):
    rv, w, b, rm = (pre[f"{p}.{bn}.{s}"] for s in ("running_var", "weight", "bias", "running_mean"))
    std_c = (rv + 1e-5).sqrt(); t_c = w / std_c; bb_c = b - rm * w / std_c
    rvd, wd, bd, rmd = rv.to(dev), w.to(dev), b.to(dev), rm.to(dev)
    std_g = (rvd + 1e-5).sqrt(); t_g = wd / std_g; bb_g = bd - rmd * wd / std_g
    print(bn, "torch-gpu vs cpu: std", int((std_g.cpu() != std_c).sum()), "t", int((t_g.cpu() != t_c).sum()), "b", int((bb_g.cpu() != bb_c).sum()))
