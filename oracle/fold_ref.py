"""Oracle: RepUNet fold in numpy fp32, one separately rounded IEEE operation per step, in the
reference's order (test infrastructure only).

Reference followed: mergeBN /root/reference/models/reparm_unet.py:28-31, kernelFuse :33-37,
mergeAsyKernels :74-82, mergeMulKernels :124-135, switch_to_deploy :217-225.

Why numpy and a pluggable sqrt: torch's CPU `sqrt` goes through MKL VML, which is NOT correctly
rounded (about 0.7 % of inputs differ by one ulp from the IEEE result), so the reference's fold is not
bit-reproducible across its own backends.  With `sqrt=torch_sqrt` this restatement reproduces the
golden vectors captured from the reference BIT FOR BIT (that pins the operation order); with the
default correctly rounded `np.sqrt` it is the IEEE oracle the HIP fold kernel must match bit for bit.
"""
from typing import Callable, Dict, List

import numpy as np

f32 = np.float32


def torch_sqrt(x: np.ndarray) -> np.ndarray:
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).sqrt().numpy()


def merge_bn(conv_w, bn_w, bn_b, rm, rv, eps=1e-5, sqrt: Callable = np.sqrt):
    std = sqrt((rv + f32(eps)).astype(f32)).astype(f32)
    t = (bn_w / std).astype(f32).reshape(-1, 1, 1, 1)
    return (conv_w * t).astype(f32), (bn_b - ((rm * bn_w).astype(f32) / std).astype(f32)).astype(f32)


def kernel_fuse(target, sec):
    sh, sw = sec.shape[2], sec.shape[3]
    th, tw = target.shape[2], target.shape[3]
    h0, w0 = th // 2 - sh // 2, tw // 2 - sw // 2
    target[:, :, h0:h0 + sh, w0:w0 + sw] += sec


def fold_asymm(st: Dict[str, np.ndarray], p: str, sqrt=np.sqrt):
    def mb(conv, bn):
        return merge_bn(st[f"{p}.{conv}.weight"], st[f"{p}.{bn}.weight"], st[f"{p}.{bn}.bias"],
                        st[f"{p}.{bn}.running_mean"], st[f"{p}.{bn}.running_var"], sqrt=sqrt)
    w_ve, b_ve = mb("convVe1", "bnVe1")
    w_ho, b_ho = mb("convHo1", "bnHo1")
    w_sq, b_sq = mb("convSq", "bnSq")
    kernel_fuse(w_sq, w_ve)
    kernel_fuse(w_sq, w_ho)
    return w_sq, ((b_sq + b_ve).astype(f32) + b_ho).astype(f32)


def fold_state(st: Dict[str, np.ndarray], ks: List[int], sqrt=np.sqrt) -> Dict[str, np.ndarray]:
    st = {k: np.asarray(v) for k, v in st.items()}
    ks = sorted(ks)
    out, prefixes = {}, []
    for k in st:
        if ".conv_list." in k:
            pre = k.split(".conv_list.")[0]
            if pre not in prefixes:
                prefixes.append(pre)
        else:
            out[k] = st[k]
    for pre in prefixes:
        parts = [fold_asymm(st, f"{pre}.conv_list.{i}", sqrt) for i in range(len(ks))]
        tw, tb = parts[-1][0].copy(), parts[-1][1].copy()
        for i in range(len(ks) - 1):
            kernel_fuse(tw, parts[i][0])
            tb += parts[i][1]
        out[pre + ".mergedConv.weight"], out[pre + ".mergedConv.bias"] = tw, tb
    return out
