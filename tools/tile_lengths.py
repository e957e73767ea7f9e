"""Tile list lengths of the headline view (and the heavy-pair view with --heavy): distribution and what it says about the blend
kernels' schedule — total entries / concurrent workgroups against the longest tile.  GPU box.
usage: python tools/tile_lengths.py [--heavy]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contextgs_amd import rasterizer
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

heavy = "--heavy" in sys.argv
pc = make_scene(1_000_000, seed=0, voxel_size=0.01) if heavy else make_scene(1_000_000, seed=0)
pc.train()
cam = orbit_cameras(8, 1920, 1080)[0].to_torch("cuda")
pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
vis = prefilter_voxel(cam, pc, pipe, bg)
pkg = render(cam, pc, pipe, bg, visible_mask=vis, step=20000)
torch.cuda.synchronize()
lc = rasterizer.last_call
nt = ((1080 + 15) // 16) * ((1920 + 15) // 16)
r = lc["img_ws"][: nt * 8].view(torch.int32).view(nt, 2).long()
ln = (r[:, 1] - r[:, 0]).clamp(min=0).float()
q = torch.quantile(ln, torch.tensor([0.5, 0.9, 0.99, 0.999], device="cuda"))
print(f"{'heavy' if heavy else 'headline'}: tiles {nt}, pairs {int(ln.sum())}, mean {float(ln.mean()):.0f}, median {float(q[0]):.0f}, p90 {float(q[1]):.0f}, "
      f"p99 {float(q[2]):.0f}, p99.9 {float(q[3]):.0f}, max {float(ln.max()):.0f}")
for wg_per_cu, name in ((7, "backward (7 workgroups per CU)"), (8, "forward (8 per CU)")):
    conc = 256 * wg_per_cu
    print(f"  {name}: {conc} concurrent tiles; total / concurrency = {float(ln.sum()) / conc:.0f} entries per slot, longest tile {float(ln.max()):.0f} "
          f"-> a perfectly balanced schedule is bound by the {'longest tile' if float(ln.max()) > float(ln.sum()) / conc else 'total'}")
