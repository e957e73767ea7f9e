"""Golden vectors for the image loss from the reference's own Python
(utils/loss_utils.py: l1_loss, ssim), run on CPU in the authoring container.  Writes tests/golden/loss.npz
(inputs, the two loss values and the gradient of train.py:199-204's combination w.r.t. the rendered image)."""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference/utils/loss_utils.py"
spec = importlib.util.spec_from_file_location("ref_loss_utils", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for tag, (C, H, W), seed in (("a", (3, 37, 53), 0), ("b", (3, 16, 16), 1), ("c", (3, 5, 70), 2), ("d", (1, 64, 48), 3)):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(C, H, W, generator=g)
    img = (gt + 0.15 * torch.randn(C, H, W, generator=g)).clamp(0, 1).requires_grad_()
    l1 = ref.l1_loss(img, gt)
    s = ref.ssim(img, gt)
    loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - s)                       # train.py:199-204, lambda_dssim = 0.2
    (grad,) = torch.autograd.grad(loss, [img])
    out[f"{tag}_img"], out[f"{tag}_gt"] = img.detach().numpy(), gt.numpy()
    out[f"{tag}_l1"], out[f"{tag}_ssim"], out[f"{tag}_grad"] = l1.item(), s.item(), grad.numpy()
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loss.npz")
np.savez_compressed(path, **out)
print(path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("a_")})
