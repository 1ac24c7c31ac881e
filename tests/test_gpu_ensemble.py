"""BASELINE configs[4] as a pipeline: UNet (eval, BN folded) + deployed ReparamUNet -> per-pixel entropy fusion, against
the oracle (oracle/unet_ref.py forwards + oracle/loss_ref.py ensemble_fuse = /root/reference/eval_ensemble.py:61-99 per
pixel) at golden size.

  fp32 kernels: both members' logits <= 1e-4 relative; the fused INTEGER prediction equals the oracle's bit for bit at
                every pixel that is not within rounding of one of the rule's two discontinuities.
  bf16 kernels (the config's dtype): the fusion kernel on the bf16 members' logits equals the oracle's rule on those
                same logits bit for bit (away from the discontinuities); against the all-fp32 oracle the prediction may
                differ only where the members' bf16 logits moved a pixel across a boundary: agreement >= 97 %
                (random-init members are near-ties almost everywhere, the hard case).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _fragile(l1, l2, thresh):
    from oracle import loss_ref
    p1, p2 = torch.softmax(l1, 1), torch.softmax(l2, 1)
    e1, e2 = loss_ref.entropy(p1), loss_ref.entropy(p2)
    w1, w2 = 1 / (e1 + 1e-8), 1 / (e2 + 1e-8)
    pe = (w1[:, None] * p1 + w2[:, None] * p2) / (w1 + w2)[:, None]
    srt = pe.sort(1, descending=True).values
    s1, s2 = p1.sort(1, descending=True).values, p2.sort(1, descending=True).values
    return ((((e1 - e2).abs() - thresh).abs() < 1e-4) | ((srt[:, 0] - srt[:, 1]) < 1e-5)
            | ((s1[:, 0] - s1[:, 1]) < 1e-5) | ((s2[:, 0] - s2[:, 1]) < 1e-5))


def _members(dev):
    from unet_amd.models import REGISTRY
    z = np.load(os.path.join(G, "fold_k35.npz"))
    pre = {k[4:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith("pre/")}
    x = torch.from_numpy(np.asarray(z["x"]))
    torch.manual_seed(5)
    u = REGISTRY["unet"](in_channels=1, n_classes=2, features=[8, 16])
    gen = torch.Generator().manual_seed(6)
    ust = {k: v.clone() for k, v in u.state_dict().items()}
    for k in ust:                                           # a trained-looking eval state: non-trivial running statistics
        if k.endswith("running_mean"):
            ust[k] = torch.randn(ust[k].shape, generator=gen) * 0.1
        if k.endswith("running_var"):
            ust[k] = torch.rand(ust[k].shape, generator=gen) + 0.5
    u.load_state_dict(ust)
    r = REGISTRY["reparam_unet"](in_channels=1, n_classes=2, features=[8, 16], kernel_sizes=[3, 5], norm="batch")
    r.load_state_dict(pre)
    return u.to(dev), ust, r.to(dev), pre, x


@pytest.mark.parametrize("amp", [False, True])
def test_unet_plus_deployed_repunet_ensemble_vs_oracle(dev, amp):
    from oracle import loss_ref, unet_ref
    from unet_amd.ensemble import EnsemblePredictor
    u, ust, r, pre, x = _members(dev)
    thresh = 0.05           # the reference's 0.6 never triggers on random-init members (entropies ~ln 2): exercise all three branches
    with torch.no_grad():
        ref1 = unet_ref.unet_forward(x, ust, train=False)
        ref2 = unet_ref.repunet_forward(x, unet_ref.fold_state({k: v.clone() for k, v in pre.items()}, [3, 5]), [3, 5],
                                        norm="batch", train=False)
    ref_pred = loss_ref.ensemble_fuse(ref1, ref2, thresh)
    r.eval()
    r.switch_to_deploy()
    ens = EnsemblePredictor(u, r, entropy_thresh=thresh, amp=amp)
    l1, l2 = ens.logits(x.to(dev))
    pred, p1, p2 = ens.predict(x.to(dev))
    l1c, l2c = l1.cpu(), l2.cpu()
    assert pred.dtype == torch.int64 and tuple(pred.shape) == (x.shape[0], x.shape[2], x.shape[3])
    # the fusion kernel on the logits the members produced: the oracle's rule on the same numbers
    same_in = loss_ref.ensemble_fuse(l1c, l2c, thresh)
    frag = _fragile(l1c, l2c, thresh)
    assert torch.equal(pred.cpu()[~frag], same_in[~frag])
    assert torch.equal(p1.cpu()[~frag], l1c.argmax(1)[~frag]) and torch.equal(p2.cpu()[~frag], l2c.argmax(1)[~frag])
    branches = ((loss_ref.entropy(torch.softmax(l1c, 1)) - loss_ref.entropy(torch.softmax(l2c, 1))).abs() > thresh).float().mean()
    assert 0.02 < float(branches) < 0.98, "the case must exercise both the override and the weighted branch"
    if not amp:
        rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
        assert rel(l1c, ref1) < 1e-4 and rel(l2c, ref2) < 1e-4
        frag_ref = _fragile(ref1, ref2, thresh) | frag
        assert float(frag_ref.float().mean()) < 0.02
        assert torch.equal(pred.cpu()[~frag_ref], ref_pred[~frag_ref])
    else:
        agree = float((pred.cpu() == ref_pred).float().mean())
        print(f"bf16 ensemble vs fp32 oracle: agreement {agree:.4f}")
        assert agree >= 0.97


def test_ensemble_evaluate_counts(dev):
    from unet_amd.ensemble import EnsemblePredictor
    u, ust, r, pre, x = _members(dev)
    r.eval()
    r.switch_to_deploy()
    ens = EnsemblePredictor(u, r, entropy_thresh=0.05, amp=True)
    y = (torch.rand(x.shape[0], x.shape[2], x.shape[3], generator=torch.Generator().manual_seed(3)) > 0.5).long()
    stats = ens.evaluate([(x, y), (x, y.unsqueeze(1))], dev)           # num_classes from the logits' channel count
    pred, p1, p2 = ens.predict(x.to(dev))
    assert stats["pixels"] == 2 * y.numel()
    # eval_ensemble.py:22-30,123-130: overall = mean(pred == target), per_class[c] = correct_c / max(total_c, 1)
    for name, p in (("Model1", p1), ("Model2", p2), ("Ensemble", pred)):
        pc = p.cpu()
        assert abs(stats[name]["overall"] - float((pc == y).float().mean())) < 1e-9
        for c in range(2):
            want = float(((pc == c) & (y == c)).sum()) / max(int((y == c).sum()), 1)
            assert abs(stats[name]["per_class"][c] - want) < 1e-6
    assert abs(stats["acc_pixel_ensemble"] - float((pred.cpu() == y).float().mean())) < 1e-9
    assert abs(stats["acc_pixel_model1"] - float((p1.cpu() == y).float().mean())) < 1e-9
    assert abs(stats["acc_pixel_model2"] - float((p2.cpu() == y).float().mean())) < 1e-9
