"""CT slice dataset with the transform moved to the GPU.

The reference's CT_NpyDataset (/root/reference/data/ct_dataset.py:10-84) windows, resizes, flips and
rotates every slice on a CPU DataLoader worker.  Here `__getitem__` only reads the two `.npy` files and
draws the augmentation decisions with the reference's `random` call order (:52-63); the batch goes to the
GPU raw (4 B image + 1 B mask per pixel instead of 4 B + 8 B) and `DeviceCTPipeline` applies the whole
transform in one kernel (`unet_ct_preprocess`).

    train_ds, val_ds = REGISTRY["ct_segmentation"](data_dir, image_size=512)       # as the reference
    loader = DataLoader(train_ds, batch_size=32, collate_fn=device_collate, ...)
    pipe = DeviceCTPipeline(image_size=512)
    for raw in loader:
        x, y = pipe(raw, device)            # x fp32 [N,1,S,S], y int64 [N,S,S]: what the reference's loader yields

On-disk layout as the reference's: {root}/{train,val}/{images,masks}/<name>.npy (float32 image, uint8 mask).
"""
import math
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch

from .. import ops
from .._lib import UNetHipError, check, lib
from . import REGISTRY

MAX_BATCH = 64


def draw_augment(rng=random):
    """hflip?, vflip?, rotate? [, angle]: ct_dataset.py:52-63, same calls in the same order."""
    hflip = rng.random() > 0.5
    vflip = rng.random() > 0.5
    angle = None
    if rng.random() > 0.5:
        angle = rng.uniform(-15, 15)
    return hflip, vflip, angle


def rotation_theta(angle: float, size: int) -> np.ndarray:
    """The 3x2 float32 matrix torchvision's rotate multiplies its base grid with, flattened row-major
    ([T00 T01 T10 T11 T20 T21]): _get_inverse_affine_matrix([0,0], -angle, ...) then
    theta.T / [0.5 w, 0.5 h] in float32 (torchvision _functional_tensor._gen_affine_grid)."""
    rot = math.radians(-angle)
    m = np.array([[math.cos(rot), math.sin(rot), 0.0], [-math.sin(rot), math.cos(rot), 0.0]], dtype=np.float32)
    t = m.T / np.array([0.5 * size, 0.5 * size], dtype=np.float32)
    return np.ascontiguousarray(t, dtype=np.float32).reshape(6)


class CT_NpyDataset(torch.utils.data.Dataset):
    """Same constructor and file discovery as the reference (ct_dataset.py:11-23).  Returns the RAW slice
    and the augmentation decisions; the transform itself runs in DeviceCTPipeline."""

    def __init__(self, root_dir, split="train", image_size=512, **kwargs):
        self.split_dir = os.path.join(root_dir, split)
        self.images_dir = os.path.join(self.split_dir, "images")
        self.masks_dir = os.path.join(self.split_dir, "masks")
        self.files = sorted([f for f in os.listdir(self.images_dir) if f.endswith(".npy")])
        self.image_size = image_size
        self.augment = (split == "train")

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        name = self.files[idx]
        image = np.load(os.path.join(self.images_dir, name))
        mask = np.load(os.path.join(self.masks_dir, name))
        aug = draw_augment(random) if self.augment else (False, False, None)
        return image, mask, aug


def device_collate(samples):
    """-> (images float32 [N,H0,W0], masks uint8 [N,H0,W0], [(hflip, vflip, angle)]) as plain CPU tensors.
    Masks with values above 255 are saturated to 255 first (only `> 0` survives the transform).
    It runs inside DataLoader worker processes (num_workers > 0), which must never touch the GPU runtime, so it does not
    pin: ask the loader for that (`DataLoader(..., pin_memory=True)` pins in the parent's pin thread)."""
    imgs = np.stack([np.asarray(s[0], dtype=np.float32) for s in samples])
    masks = np.stack([np.minimum(np.asarray(s[1]), 255).astype(np.uint8) if np.asarray(s[1]).dtype != np.uint8
                      else np.asarray(s[1]) for s in samples])
    return torch.from_numpy(imgs), torch.from_numpy(masks), [s[2] for s in samples]


class DeviceCTPipeline:
    def __init__(self, image_size: int = 512):
        self.image_size = int(image_size)

    def __call__(self, raw, device=None) -> Tuple[torch.Tensor, torch.Tensor]:
        images, masks, augs = raw
        return self.transform(images, masks, augs, device)

    def transform(self, images: torch.Tensor, masks: torch.Tensor, augs: Optional[List] = None, device=None):
        """images fp32 [N,H0,W0], masks uint8 [N,H0,W0] (host or device) -> (x fp32 [N,1,S,S], y int64 [N,S,S])."""
        if device is None:
            device = images.device if images.is_cuda else torch.device("cuda", torch.cuda.current_device())
        if images.dim() != 3 or masks.shape != images.shape:
            raise UNetHipError("DeviceCTPipeline: images and masks must both be [N,H0,W0]")
        if images.dtype != torch.float32 or masks.dtype != torch.uint8:
            raise UNetHipError("DeviceCTPipeline: images must be float32 and masks uint8")
        N, H0, W0 = images.shape
        S = self.image_size
        if augs is None:
            augs = [(False, False, None)] * N
        if len(augs) != N:
            raise UNetHipError("DeviceCTPipeline: one augmentation tuple per sample")
        xs, ys = [], []
        for n0 in range(0, N, MAX_BATCH):
            n1 = min(N, n0 + MAX_BATCH)
            xs_, ys_ = self._run(images[n0:n1], masks[n0:n1], augs[n0:n1], device, H0, W0, S)
            xs.append(xs_); ys.append(ys_)
        return (xs[0], ys[0]) if len(xs) == 1 else (torch.cat(xs), torch.cat(ys))

    @staticmethod
    def _run(images, masks, augs, device, H0, W0, S):
        import ctypes
        n = images.shape[0]
        img = images.to(device, non_blocking=True).contiguous()
        msk = masks.to(device, non_blocking=True).contiguous()
        flags = (ctypes.c_int * n)()
        theta = (ctypes.c_float * (6 * n))()
        for i, (hf, vf, angle) in enumerate(augs):
            flags[i] = (1 if hf else 0) | (2 if vf else 0) | (4 if angle is not None else 0)
            if angle is not None:
                theta[6 * i:6 * i + 6] = rotation_theta(float(angle), S).tolist()
        x = torch.empty((n, 1, S, S), dtype=torch.float32, device=device)
        y = torch.empty((n, S, S), dtype=torch.int64, device=device)
        timg = tmsk = None
        if (H0, W0) != (S, S):
            timg = torch.empty((n, S, S), dtype=torch.float32, device=device)
            tmsk = torch.empty((n, S, S), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            check(lib().unet_ct_preprocess(img.data_ptr(), msk.data_ptr(), n, H0, W0, S, flags, theta,
                                           None if timg is None else timg.data_ptr(),
                                           None if tmsk is None else tmsk.data_ptr(),
                                           x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream),
                  "ct_preprocess")
        return x, y


def build_ct_dataset(data_dir, **kwargs):
    return CT_NpyDataset(data_dir, split="train", **kwargs), CT_NpyDataset(data_dir, split="val", **kwargs)


REGISTRY["ct_segmentation"] = build_ct_dataset
