"""Device input pipeline against the oracle restatement of CT_NpyDataset.__getitem__
(/root/reference/data/ct_dataset.py:25-84).  CPU tests pin the oracle / host logic; GPU tests compare
`unet_ct_preprocess` (through DeviceCTPipeline) with the oracle on the same slices and the same draws.

Tolerances: image 2e-6 absolute (values in [0,1]; fp32 bilinear weights, FMA contraction may differ);
masks bit-exact without rotation; with rotation the nearest-neighbour source index can differ where the
sampling coordinate lands within an ulp of .5, so <= 0.02 % of mask pixels may differ."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import ct_pipeline_ref as ref


def make_slice(h, w, seed):
    g = np.random.default_rng(seed)
    img = (g.random((h, w), dtype=np.float32) * 3000.0 - 1800.0).astype(np.float32)     # HU incl. values outside the window
    mask = (g.random((h, w)) > 0.8).astype(np.uint8) * np.uint8(255 if seed % 2 else 1)  # raw {0,255} and clean {0,1} data
    return img, mask


# ------------------------------------------------------------------------------------------- CPU

def test_oracle_identity_and_window():
    img, mask = make_slice(16, 16, 1)
    x, y = ref.ct_transform(img, mask, 16)
    assert x.shape == (1, 16, 16) and y.shape == (16, 16) and y.dtype == torch.int64
    assert float(x.min()) >= 0.0 and float(x.max()) <= 1.0
    exp = (np.clip(img, -1000, 500) + 1000) / 1500
    assert np.array_equal(x[0].numpy(), exp.astype(np.float32))
    assert np.array_equal(y.numpy(), (mask > 0).astype(np.int64))
    # zero-degree rotation is the identity (pins the grid convention: base grid, theta scaling, unnormalize)
    xr, yr = ref.ct_transform(img, mask, 16, angle=0.0)
    assert torch.allclose(xr, x, atol=1e-6) and torch.equal(yr, y)
    # flips commute and are involutions
    a, _ = ref.ct_transform(img, mask, 16, hflip=True, vflip=True)
    assert torch.equal(a, x.flip(-1).flip(-2))


def test_oracle_rotation_moves_mass_the_right_way():
    """A bright dot right of the centre moves DOWN-right->up? torchvision: positive angle = counter-clockwise."""
    img = np.full((33, 33), -1000.0, dtype=np.float32)
    img[16, 26] = 500.0                                   # 10 px right of the centre
    mask = np.zeros((33, 33), dtype=np.uint8); mask[16, 26] = 1
    x, y = ref.ct_transform(img, mask, 33, angle=15.0)
    r, c = np.unravel_index(int(x[0].argmax()), (33, 33))
    assert r < 16 and c > 16                              # counter-clockwise on screen: up
    assert int(y.sum()) >= 1 and int(y[:16].sum()) >= 1


def test_draw_order_matches_reference_sequence():
    from unet_amd.data.ct_device import draw_augment
    rng = random.Random(123)
    mine = [draw_augment(rng) for _ in range(50)]
    rng = random.Random(123)
    exp = []
    for _ in range(50):
        h = rng.random() > 0.5
        v = rng.random() > 0.5
        a = rng.uniform(-15, 15) if rng.random() > 0.5 else None
        exp.append((h, v, a))
    assert mine == exp and mine == [ref.draw_augment(r2) for r2 in [random.Random(123)] for _ in range(50)]
    assert any(a is not None for _, _, a in mine) and any(a is None for _, _, a in mine)


def test_dataset_reads_reference_layout(tmp_path):
    from unet_amd.data import REGISTRY, device_collate
    for split in ("train", "val"):
        for sub in ("images", "masks"):
            os.makedirs(tmp_path / split / sub)
        for i in range(3):
            img, mask = make_slice(8, 8, i)
            np.save(tmp_path / split / "images" / f"p0_slice_{i:03d}.npy", img)
            np.save(tmp_path / split / "masks" / f"p0_slice_{i:03d}.npy", mask)
        (tmp_path / split / "images" / ".hidden").write_text("x")
    tr, va = REGISTRY["ct_segmentation"](str(tmp_path), image_size=8)
    assert len(tr) == 3 and len(va) == 3 and tr.augment and not va.augment
    assert va[0][2] == (False, False, None)
    imgs, masks, augs = device_collate([tr[i] for i in range(3)])
    assert imgs.shape == (3, 8, 8) and imgs.dtype == torch.float32 and masks.dtype == torch.uint8 and len(augs) == 3


def test_collate_runs_inside_dataloader_workers(tmp_path):
    """The documented configuration: DataLoader(num_workers=2, collate_fn=device_collate).  The collate function runs
    in forked worker processes, so it must hand back plain (unpinned) CPU tensors and never touch the GPU runtime."""
    from torch.utils.data import DataLoader
    from unet_amd.data import REGISTRY, device_collate
    for split in ("train", "val"):
        for sub in ("images", "masks"):
            os.makedirs(tmp_path / split / sub)
        for i in range(6):
            img, mask = make_slice(8, 8, i)
            np.save(tmp_path / split / "images" / f"p0_slice_{i:03d}.npy", img)
            np.save(tmp_path / split / "masks" / f"p0_slice_{i:03d}.npy", mask)
    tr, _ = REGISTRY["ct_segmentation"](str(tmp_path), image_size=8)
    seen = 0
    for imgs, masks, augs in DataLoader(tr, batch_size=4, collate_fn=device_collate, num_workers=2):
        assert imgs.dtype == torch.float32 and masks.dtype == torch.uint8 and imgs.shape[1:] == (8, 8)
        assert not imgs.is_pinned() and not imgs.is_cuda and len(augs) == imgs.shape[0]
        seen += imgs.shape[0]
    assert seen == 6


def test_theta_matches_oracle():
    from unet_amd.data.ct_device import rotation_theta
    for ang in (-15.0, -3.3, 0.0, 7.5, 14.999):
        t = ref.rescaled_theta(ref.rotate_matrix(ang), 512, 512).numpy().reshape(6)
        assert np.array_equal(rotation_theta(ang, 512), t)


# ------------------------------------------------------------------------------------------- GPU

AUGS = [(False, False, None), (True, False, None), (False, True, None), (True, True, None),
        (False, False, 11.25), (True, False, -15.0), (False, True, 3.7), (True, True, -0.01), (False, False, 15.0)]


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(64, 64, 64), (40, 48, 32), (75, 50, 96), (16, 16, 48), (512, 512, 512)])
def test_device_pipeline_matches_oracle(geom):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from unet_amd.data import DeviceCTPipeline
    H0, W0, S = geom
    n = len(AUGS) if S < 512 else 3
    augs = AUGS[:n] if S < 512 else [AUGS[0], AUGS[5], AUGS[8]]
    slices = [make_slice(H0, W0, 100 + i) for i in range(n)]
    imgs = torch.from_numpy(np.stack([s[0] for s in slices]))
    masks = torch.from_numpy(np.stack([s[1] for s in slices]))
    x, y = DeviceCTPipeline(S).transform(imgs, masks, augs, torch.device("cuda:0"))
    assert x.shape == (n, 1, S, S) and x.dtype == torch.float32 and y.shape == (n, S, S) and y.dtype == torch.int64
    x, y = x.cpu(), y.cpu()
    for i, (hf, vf, ang) in enumerate(augs):
        rx, ry = ref.ct_transform(slices[i][0], slices[i][1], S, hf, vf, ang)
        assert float((x[i] - rx).abs().max()) <= 2e-6, (i, hf, vf, ang)
        bad = int((y[i] != ry).sum())
        if ang is None:
            assert bad == 0, (i, hf, vf)
        else:
            assert bad <= max(1, int(2e-4 * S * S)), (i, ang, bad)
        assert set(y[i].unique().tolist()) <= {0, 1}


@pytest.mark.gpu
def test_device_pipeline_big_batch_and_errors():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from unet_amd.data import DeviceCTPipeline
    from unet_amd._lib import UNetHipError
    n = 70                                              # more than one 64-slice launch
    rng = random.Random(5)
    augs = [ref.draw_augment(rng) for _ in range(n)]
    slices = [make_slice(32, 32, 300 + i) for i in range(n)]
    imgs = torch.from_numpy(np.stack([s[0] for s in slices]))
    masks = torch.from_numpy(np.stack([s[1] for s in slices]))
    x, y = DeviceCTPipeline(32)((imgs, masks, augs), torch.device("cuda:0"))
    for i in (0, 63, 64, 69):
        rx, ry = ref.ct_transform(slices[i][0], slices[i][1], 32, *augs[i])
        assert float((x[i].cpu() - rx).abs().max()) <= 2e-6
        assert int((y[i].cpu() != ry).sum()) <= 1
    with pytest.raises(UNetHipError):
        DeviceCTPipeline(32).transform(imgs.double(), masks, augs, torch.device("cuda:0"))
