"""/root/reference/utils/io.py:4-45 with the same names: timestamp, ensure_dir, save_config, save_state and the
inference-side load_checkpoint (state_dict under `state_dict` / `model` / `model_state_dict` or bare; `module.`
prefix stripped; strict=False)."""
import os
from datetime import datetime

import torch
import yaml


def timestamp():
    return datetime.now().strftime("%Y%m%d-%H%M%S")


def ensure_dir(p):
    os.makedirs(p, exist_ok=True)


def save_config(cfg, out_dir):
    ensure_dir(out_dir)
    with open(os.path.join(out_dir, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)


def save_state(path, state):
    torch.save(state, path)


def load_checkpoint(path, model, device="cuda", weights_only=True):
    ckpt = torch.load(path, map_location=device, weights_only=weights_only)
    if "state_dict" in ckpt:
        state_dict = ckpt["state_dict"]
    elif "model" in ckpt:
        state_dict = ckpt["model"]
    elif "model_state_dict" in ckpt:
        state_dict = ckpt["model_state_dict"]
    else:
        state_dict = ckpt
    new_state_dict = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
    missing, unexpected = model.load_state_dict(new_state_dict, strict=False)
    print(f"Loaded checkpoint from {path}")
    if missing:
        print("Missing keys:", missing)
    if unexpected:
        print("Unexpected keys:", unexpected)
    return model
