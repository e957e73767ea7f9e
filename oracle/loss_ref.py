"""TEST INFRASTRUCTURE ONLY — CPU (numpy) restatement of the reference's image loss, the checker for the HIP
kernels of contextgs_amd/csrc/loss.hip.  Imported only by tests/ (never by the product path).

Follows utils/loss_utils.py of the reference:
  l1_loss  :17-18   mean |a - b|
  gaussian :23-25   11 taps, sigma 1.5, normalised
  _ssim    :43-63   zero-padded per-channel 11x11 filtering of x, y, x^2, y^2, xy; C1 = 0.01^2, C2 = 0.03^2;
                    ssim_map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)); mean
Pinned by tests/golden/loss.npz (values and gradients produced by the reference's own functions,
tools/make_loss_golden.py).  The gradient below is the analytic adjoint of the same expressions.

`scaling_reg` / `mask_reg` restate the two regularisers train.py:203,209 adds to the image terms (they are inline torch
expressions there, not functions): pinned in tests/test_loss.py against torch's CPU autograd of the same expressions.
"""
from math import exp

import numpy as np


def window(dtype=np.float32):
    g = np.array([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], dtype=np.float32)
    g = g / g.sum(dtype=np.float32)
    return g.astype(dtype)


def _filt(a, w):
    """zero-padded 11x11 separable filtering of a [C,H,W] array (rows then columns)."""
    C, H, W = a.shape
    p = np.zeros((C, H, W + 10), dtype=a.dtype)
    p[:, :, 5:5 + W] = a
    r = np.zeros_like(a)
    for k in range(11):
        r += w[k] * p[:, :, k:k + W]
    p = np.zeros((C, H + 10, W), dtype=a.dtype)
    p[:, 5:5 + H, :] = r
    o = np.zeros_like(a)
    for k in range(11):
        o += w[k] * p[:, k:k + H, :]
    return o


def l1_ssim(img, gt, dtype=np.float32):
    """(l1 mean, ssim mean, dl1/dimg, dssim/dimg) for [C,H,W] arrays."""
    x, y = img.astype(dtype), gt.astype(dtype)
    w = window(dtype)
    n = x.size
    C1, C2 = dtype(0.01 ** 2), dtype(0.03 ** 2)
    mu1, mu2 = _filt(x, w), _filt(y, w)
    e11, e22, e12 = _filt(x * x, w), _filt(y * y, w), _filt(x * y, w)
    s1, s2, s12 = e11 - mu1 * mu1, e22 - mu2 * mu2, e12 - mu1 * mu2
    a1, a2 = 2 * mu1 * mu2 + C1, 2 * s12 + C2
    b1, b2 = mu1 * mu1 + mu2 * mu2 + C1, s1 + s2 + C2
    m = (a1 * a2) / (b1 * b2)
    dm_ds1 = -(a1 * a2) / (b1 * b2 * b2)
    dm_ds12 = 2 * a1 / (b1 * b2)
    dm_dmu1 = (2 * mu2 * a2 * b1 - 2 * mu1 * a1 * a2) / (b1 * b1 * b2) - 2 * mu1 * dm_ds1 - mu2 * dm_ds12
    g_ssim = (_filt(dm_dmu1, w) + 2 * x * _filt(dm_ds1, w) + y * _filt(dm_ds12, w)) / dtype(n)
    g_l1 = np.sign(x - y) / dtype(n)
    return float(np.abs(x - y).mean(dtype=np.float64)), float(m.mean(dtype=np.float64)), g_l1, g_ssim


def scaling_reg(scaling):
    """train.py:203 `scaling.prod(dim=1).mean()` for [P,3]: (value, d value / d scaling) in float64."""
    s = np.asarray(scaling, dtype=np.float64)
    P = s.shape[0]
    d = np.stack([s[:, 1] * s[:, 2], s[:, 0] * s[:, 2], s[:, 0] * s[:, 1]], axis=1) / P
    return float((s[:, 0] * s[:, 1] * s[:, 2]).mean()), d


def mask_reg(mask):
    """train.py:209 `torch.mean(torch.sigmoid(gaussians._mask))`: (value, d value / d mask) in float64."""
    x = np.asarray(mask, dtype=np.float64)
    sg = 1.0 / (1.0 + np.exp(-x))
    return float(sg.mean()), sg * (1.0 - sg) / x.size
