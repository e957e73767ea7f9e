"""Image loss: the numpy oracle against the goldens produced by the reference's own
utils/loss_utils.py (CPU), and the fused HIP kernels against both (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loss.npz")
TAGS = ["a", "b", "c", "d"]


def _case(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_reference_goldens(tag):
    from oracle.loss_ref import l1_ssim
    c = _case(tag)
    l1, s, g_l1, g_s = l1_ssim(c["img"], c["gt"])
    assert abs(l1 - float(c["l1"])) < 1e-6 and abs(s - float(c["ssim"])) < 2e-6
    grad = 0.8 * g_l1 - 0.2 * g_s                                      # d[(1-l) L1 + l (1 - SSIM)], l = 0.2
    assert np.abs(grad - c["grad"]).max() < 2e-7 + 2e-4 * np.abs(c["grad"]).max()
    # fp64 restatement: same numbers, i.e. the fp32 ones are not a rounding accident
    l1d, sd, _, g_sd = l1_ssim(c["img"], c["gt"], dtype=np.float64)
    assert abs(sd - s) < 2e-6 and np.abs(g_sd - g_s).max() < 1e-4 * np.abs(g_s).max() + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_matches_goldens_and_oracle(tag):
    import torch
    from contextgs_amd.loss_utils import l1_loss, l1_ssim, ssim
    c = _case(tag)
    img = torch.tensor(c["img"], device="cuda", requires_grad=True)
    gt = torch.tensor(c["gt"], device="cuda")
    l1, s = l1_ssim(img, gt)
    assert abs(float(l1) - float(c["l1"])) < 1e-6 and abs(float(s) - float(c["ssim"])) < 2e-6
    loss = 0.8 * l1 + 0.2 * (1.0 - s)
    (g,) = torch.autograd.grad(loss, [img])
    assert np.abs(g.cpu().numpy() - c["grad"]).max() < 2e-7 + 2e-4 * np.abs(c["grad"]).max()
    # the reference's separate entry points
    assert abs(float(ssim(img, gt)) - float(c["ssim"])) < 2e-6 and abs(float(l1_loss(img, gt)) - float(c["l1"])) < 1e-6


@pytest.mark.gpu
def test_hip_full_hd_properties_and_torch_reference():
    """1080p: SSIM(x, x) == 1 with zero gradient; against a plain torch fp32 conv2d restatement of the same op."""
    import torch
    import torch.nn.functional as F
    from contextgs_amd.loss_utils import l1_ssim
    from oracle.loss_ref import window
    g = torch.Generator(device="cuda").manual_seed(3)
    gt = torch.rand(3, 1080, 1920, device="cuda", generator=g)
    img = (gt + 0.1 * torch.randn(3, 1080, 1920, device="cuda", generator=g)).clamp(0, 1).requires_grad_()
    l1_same, s_same = l1_ssim(gt.clone().requires_grad_(), gt)
    assert float(l1_same) == 0.0 and abs(float(s_same) - 1.0) < 1e-6
    w1 = torch.tensor(window(), device="cuda")
    w2 = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t[None], w2, padding=5, groups=3)[0]
    mu1, mu2 = conv(img), conv(gt)
    s1, s2, s12 = conv(img * img) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(img * gt) - mu1 * mu2
    ref_s = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    ref_l1 = (img - gt).abs().mean()
    l1, s = l1_ssim(img, gt)
    assert abs(float(l1) - float(ref_l1)) < 1e-6 and abs(float(s) - float(ref_s)) < 5e-6
    (ga,) = torch.autograd.grad(0.8 * l1 + 0.2 * (1 - s), [img])
    (gb,) = torch.autograd.grad(0.8 * ref_l1 + 0.2 * (1 - ref_s), [img])
    assert float((ga - gb).abs().max()) < 2e-4 * float(gb.abs().max())
