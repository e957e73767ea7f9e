"""weighted_image_sum (the bench's linear objective) alone at 3 x 1080 x 1920: HIP events around 100 forward + backward pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.loss_utils import weighted_image_sum
img = torch.rand(3, 1080, 1920, device="cuda", requires_grad=True); w = torch.randn(3, 1080, 1920, device="cuda")
rate = torch.tensor(3.0, device="cuda", requires_grad=True)
def step():
    img.grad = None; rate.grad = None
    weighted_image_sum(img, w, rate, 0.001).backward()
for _ in range(5): step()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): step()
b.record(); torch.cuda.synchronize()
print(f"weighted_image_sum fwd + bwd: {a.elapsed_time(b) / 100 * 1e3:.1f} us per pair")
