"""CUDA-event timing of the phases of one training view (forward phases; backward as a whole)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
from contextgs_amd import renderer as R
pc = make_scene(1_000_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = {}
def run(i, rec):
    cam = cams[i % 8]
    for p in params: p.grad = None
    marks = [("start", ev())]; marks[-1][1].record()
    def mark(n):
        e = ev(); e.record(); marks.append((n, e))
    vis = R.prefilter_voxel(cam, pc, pipe, bg); mark("prefilter")
    out = R.generate_neural_gaussians(cam, pc, vis, is_training=True, step=20000); mark("generate(ctx+mlp3+expand)")
    xyz, color, opacity, scaling, rot = out[:5]
    rs = R._raster_settings(cam, pipe, bg, 1.0)
    from contextgs_amd.rasterizer import GaussianRasterizer
    sp = torch.zeros_like(xyz, requires_grad=True) + 0
    img, radii = GaussianRasterizer(rs)(means3D=xyz, means2D=sp, shs=None, colors_precomp=color, opacities=opacity, scales=scaling, rotations=rot, cov3D_precomp=None)
    mark("raster fwd")
    loss = (img * w).sum() + 0.001 * out[7]; mark("loss")
    loss.backward(); mark("backward")
    torch.cuda.synchronize()
    if rec:
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc[n1] = acc.get(n1, 0) + e0.elapsed_time(e1)
for i in range(3): run(i, False)
for i in range(5): run(i, True)
for k, v in acc.items(): print(f"{k:32s} {v/5:7.3f} ms")
