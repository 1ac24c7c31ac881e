"""Inference-side checkpoint loading with the call shape of /root/reference/utils/io.py:19-45
(`load_checkpoint(path, model, device)` -> model), which infer.py and eval_ensemble.py use.  The file may be a training
checkpoint (any alias of the model section, see checkpoint_utils.FIELDS) or a bare state_dict; a `module.` prefix is
fixed up; loading is non-strict and reports what did not line up.  The reference's other helpers in that file
(timestamp / ensure_dir / save_config / save_state) are run-directory housekeeping outside the hot path (SURVEY 2, row
13) and are not provided."""
import logging

from . import checkpoint_utils as cu

log = logging.getLogger("unet_amd.checkpoint")


def load_checkpoint(path, model, device="cuda", weights_only=True):
    ckpt = cu.read(path, map_location=device, weights_only=weights_only)
    state = cu.section(ckpt, "model")
    if state is None:
        state = ckpt                                   # a bare state_dict
    result = model.load_state_dict(cu.match_prefix(state, model.state_dict().keys()), strict=False)
    log.info("loaded %s", path)
    for label, keys in (("missing", result.missing_keys), ("unexpected", result.unexpected_keys)):
        if keys:
            log.warning("%s keys: %s", label, keys)
    return model
