"""Per-kernel times of the rasterizer's binning stage (depth sort, offsets, tile binning, ranges) on the headline scene and
on the heavy-pair variant (voxel 0.01): run under `rocprofv3 --kernel-trace --stats` by tools/bin_prof.sh.
python tools/bin_prof.py [--voxel 0.01] [--iters 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.rasterizer import last_call

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=1_000_000)
ap.add_argument("--voxel", type=float, default=None)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--mode", type=int, default=0, help="cgs_debug_set_bin_mode: 0 auto, 1 radix passes, 2 two-level (buckets)")
a = ap.parse_args()
from contextgs_amd import _lib
_lib.check(_lib.lib().cgs_debug_set_bin_mode(a.mode), "cgs_debug_set_bin_mode")
pc = make_scene(a.anchors, seed=0, **({"voxel_size": a.voxel} if a.voxel else {}))
pc.eval()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
with torch.no_grad():
    for i in range(a.iters):
        c = cams[i % len(cams)]
        render(c, pc, pipe, bg, visible_mask=prefilter_voxel(c, pc, pipe, bg), step=1000)
    torch.cuda.synchronize()
print("P", int(last_call["num_points"]) if "num_points" in last_call else "?", "R", int(last_call["num_rendered"]))
