"""What the optimizer step of the reference's training iteration (train.py:255, torch.optim.Adam over the 13 parameter groups of
scene/gaussian_model.py:426-525) costs next to the render step, at the headline size (1 M anchors): torch's default (foreach)
implementation against `fused=True`.  Outside SURVEY section 8's hot path (the optimizer stays the reference's own) — measured
for the "what comes next" list.  python tools/adam_micro.py [anchors]  ->  gpurun_out/adam_micro.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.synth import make_scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pc = make_scene(N, seed=0); pc.train()
params = [p for p in pc.parameters() if p.requires_grad]
n_el = sum(p.numel() for p in params)
lines = ["%d anchors: %d trainable tensors, %.1f M elements" % (N, len(params), n_el / 1e6)]
for fused in (False, True):
    opt = torch.optim.Adam([{"params": [p], "lr": 1e-3} for p in params], lr=0.0, eps=1e-15, fused=fused)
    for p in params:
        p.grad = torch.randn_like(p) * 1e-3
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    # Adam touches p, g, m, v (reads) and p, m, v (writes): 28 B per element
    lines.append("Adam(fused=%s): %.3f ms per step  (%.2f TB/s of the 28 B/element minimum)" % (fused, ms, 28 * n_el / ms / 1e9))
    del opt
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/adam_micro.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
