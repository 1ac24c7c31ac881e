"""Dataset registry with the reference's names (/root/reference/data/__init__.py:15-21,62):
REGISTRY["ct_segmentation"](data_dir, **kwargs) -> (train_dataset, val_dataset)."""
REGISTRY = {}


def register(name):
    def deco(fn):
        REGISTRY[name] = fn
        return fn
    return deco


from .ct_device import CT_NpyDataset, DeviceCTPipeline, build_ct_dataset, device_collate  # noqa: E402,F401
