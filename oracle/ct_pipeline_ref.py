"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of the reference's per-slice input transform,
/root/reference/data/ct_dataset.py:25-84 (CT_NpyDataset.__getitem__), with the random draws lifted out
as arguments so a device pipeline can be checked against it on the same decisions.

Window, resize, flips and the final binarisation use exactly the reference's calls (numpy clip / divide,
torch F.interpolate, tensor flips).  The rotation is `torchvision.transforms.functional.rotate`, and
torchvision is NOT installed in this image: its tensor path is restated here from the published
algorithm (torchvision/transforms/functional.py `rotate` -> `_get_inverse_affine_matrix`;
_functional_tensor.py `rotate` -> `_gen_affine_grid` -> `_apply_grid_transform` -> F.grid_sample(...,
padding_mode="zeros", align_corners=False)); the resampling itself is torch's own grid_sample.
So: window / resize / flip parity is pinned by running the reference's own calls; the rotation GRID is
"parity unpinned" (restated, no torchvision to run), its sampling is pinned to torch.
"""
import math
import random

import numpy as np
import torch
import torch.nn.functional as F


def draw_augment(rng: random.Random):
    """The reference's draw order (ct_dataset.py:52-63): hflip?, vflip?, rotate? [, angle]."""
    hflip = rng.random() > 0.5
    vflip = rng.random() > 0.5
    angle = None
    if rng.random() > 0.5:
        angle = rng.uniform(-15, 15)
    return hflip, vflip, angle


def rotate_matrix(angle: float):
    """torchvision F.rotate: matrix = _get_inverse_affine_matrix([0,0], -angle, [0,0], 1.0, [0,0])."""
    rot = math.radians(-angle)
    sx = sy = 0.0
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / 1.0 for x in m]
    m[2] += m[0] * 0.0 + m[1] * 0.0
    m[5] += m[3] * 0.0 + m[4] * 0.0
    return m


def rescaled_theta(matrix, w: int, h: int) -> torch.Tensor:
    """_gen_affine_grid's `theta.transpose(1, 2) / [0.5 w, 0.5 h]` -> [3, 2] float32."""
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    return (theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32))[0]


def _affine_grid(matrix, w, h):
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rt = rescaled_theta(matrix, w, h).unsqueeze(0)
    return base.view(1, h * w, 3).bmm(rt).view(1, h, w, 2)


def tv_rotate(img: torch.Tensor, angle: float, nearest: bool) -> torch.Tensor:
    """img [1,H,W] float32 or int64 -> rotated, zero fill, same dtype."""
    h, w = img.shape[-2:]
    grid = _affine_grid(rotate_matrix(angle), w, h)
    src = img.unsqueeze(0).to(torch.float32)
    out = F.grid_sample(src, grid, mode="nearest" if nearest else "bilinear", padding_mode="zeros", align_corners=False)
    out = out.squeeze(0)
    if not torch.is_floating_point(img):
        out = torch.round(out).to(img.dtype)
    return out


def ct_transform(image: np.ndarray, mask: np.ndarray, image_size: int, hflip=False, vflip=False, angle=None):
    """-> (image float32 [1,S,S], mask int64 [S,S]); ct_dataset.py:30-84 with the draws given."""
    image = np.clip(image, -1000, 500)
    image = (image - (-1000)) / (500 - (-1000))
    image = torch.from_numpy(image).float().unsqueeze(0)
    mask = torch.from_numpy(mask).long()
    if image.shape[-2:] != (image_size, image_size):
        image = F.interpolate(image.unsqueeze(0), size=(image_size, image_size), mode="bilinear",
                              align_corners=False).squeeze(0)
        mask = F.interpolate(mask.unsqueeze(0).unsqueeze(0).float(), size=(image_size, image_size),
                             mode="nearest").long().squeeze(0).squeeze(0)
    if hflip:
        image = image.flip(-1); mask = mask.flip(-1)
    if vflip:
        image = image.flip(-2); mask = mask.flip(-2)
    if angle is not None:
        image = tv_rotate(image, angle, nearest=False)
        mask = tv_rotate(mask.unsqueeze(0), angle, nearest=True).squeeze(0)
    mask = (mask > 0).long()
    return image, mask
