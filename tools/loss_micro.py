"""Fused L1 + SSIM (csrc/loss.hip) vs the torch composition of the reference's utils/loss_utils.py at 1920x1080,
forward + backward, CUDA-event timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from contextgs_amd.loss_utils import training_image_loss
from math import exp
gt = torch.rand(3, 1080, 1920, device="cuda")
img = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1).requires_grad_()
w1 = torch.tensor([exp(-(x - 5) ** 2 / 4.5) for x in range(11)], device="cuda"); w1 = w1 / w1.sum(); w2 = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
def ref():
    conv = lambda t: F.conv2d(t, w2, padding=5, groups=3)
    mu1, mu2 = conv(img), conv(gt)
    s1, s2, s12 = conv(img * img) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(img * gt) - mu1 * mu2
    s = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    loss = 0.8 * (img - gt).abs().mean() + 0.2 * (1 - s)
    img.grad = None; loss.backward()
def fused():
    loss = training_image_loss(img, gt, 0.2)[0]
    img.grad = None; loss.backward()
def t(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print(f"torch composition (grouped conv2d) {t(ref):.3f} ms   fused HIP {t(fused):.3f} ms   per 1080p image, fwd+bwd")
