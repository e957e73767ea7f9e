"""Per-step wall time of the heavy-pair training step (voxel 0.01) with the pair count / capacity of each view: shows what the
pair-count speculation does when consecutive views need different numbers of pairs.  python tools/heavy_steps.py [--voxel 0.01]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
from contextgs_amd.rasterizer import last_call
ap = argparse.ArgumentParser(); ap.add_argument("--voxel", type=float, default=0.01); ap.add_argument("--steps", type=int, default=24)
a = ap.parse_args()
pc = make_scene(1_000_000, seed=0, voxel_size=a.voxel); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(a.steps):
    torch.cuda.synchronize(); t = time.perf_counter()
    bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"step {i:2d} cam {i % 8}: {dt * 1e3:7.2f} ms  R {int(last_call['num_rendered']):>10d}  carved for {int(last_call['bin_R']):>10d}  "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")
