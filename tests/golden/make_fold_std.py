"""Captures std = sqrt(running_var + eps) for every BatchNorm of the fold goldens (fold_k3.npz, fold_k35.npz) with the
torch CPU build of the golden capture host (this container) -> fold_std.npz: pairs x/<i> = running_var + eps,
std/<i> = torch.sqrt(x).  torch's CPU sqrt (MKL VML) is not correctly rounded and its rounding depends on the host CPU;
the reference's fold golden was produced with THIS host's sqrt, so a test on another host replays these vectors to
reproduce the golden bit for bit (tests/test_gpu_repunet.py).  Data only: inputs and outputs of torch.sqrt."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
out, n = {}, 0
for name in ("fold_k3", "fold_k35"):
    z = np.load(os.path.join(HERE, name + ".npz"))
    for k in z.files:
        if k.startswith("pre/") and k.endswith("running_var"):
            x = torch.from_numpy(np.asarray(z[k])) + 1e-5          # mergeBN: (running_var + eps).sqrt(), eps = 1e-5
            out[f"x/{n}"], out[f"std/{n}"] = x.numpy(), x.sqrt().numpy()
            n += 1
np.savez_compressed(os.path.join(HERE, "fold_std.npz"), **out)
print(n, "vectors")
