"""Soak: N training iterations (Adam on every parameter, the real training loss) at 200 k anchors; prints the loss
trend, the peak memory and checks for NaNs / allocator growth."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.loss_utils import training_image_loss, scaling_reg
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pc = make_scene(200_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 960, 540)]
with torch.no_grad():
    pc.eval()
    gts = []
    for c in cams:
        vis = prefilter_voxel(c, pc, pipe, bg)
        gts.append((render(c, pc, pipe, bg, visible_mask=vis)["render"] * 0.7 + 0.1).clamp(0, 1))
    pc.train()
params = [p for p in pc.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=2e-3)
torch.cuda.synchronize(); t0 = time.perf_counter(); hist = []
for it in range(steps):
    c, gt = cams[it % 8], gts[it % 8]
    vis = prefilter_voxel(c, pc, pipe, bg)
    pkg = render(c, pc, pipe, bg, visible_mask=vis, step=20000)
    loss = training_image_loss(pkg["render"], gt, 0.2)[0] + 0.01 * scaling_reg(pkg["scaling"]) + 0.001 * pkg["bit_per_param"]
    opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if it % 50 == 0 or it == steps - 1:
        hist.append((it, float(loss.detach()), float(pkg["bit_per_param"].detach()), torch.cuda.memory_allocated() / 2**20))
torch.cuda.synchronize()
for h in hist: print("it %4d loss %.5f bpp %.3f alloc %.0f MiB" % h)
print("%.1f it/s, peak %.0f MiB, finite %s" % (steps / (time.perf_counter() - t0), torch.cuda.max_memory_allocated() / 2**20,
      all(torch.isfinite(p).all().item() for p in params)))
