"""Wave iterations the blend kernels actually execute (variant libraries built with -DRB_COUNT: tools/variant_lib.sh rbcnt
raster_blend_rows.hip -DRB_COUNT), one training view of the headline scene.  CGS_LIB_PATH selects the library."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
pc = make_scene(1_000_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cam = orbit_cameras(8, 1920, 1080)[0].to_torch("cuda")
L = C.CDLL(_lib.LIB_PATH)
w = torch.randn(3, 1080, 1920, device="cuda")
for rep in range(2):
    L.cgs_debug_rb_iters(None, 1)
    vis = prefilter_voxel(cam, pc, pipe, bg)
    pkg = render(cam, pc, pipe, bg, visible_mask=vis, step=1000)
    (pkg["render"] * w).sum().backward()
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 4)()
    L.cgs_debug_rb_iters(out, 0)
print(os.environ.get("CGS_LIB_PATH", "product"), "forward iterations", out[0], "backward iterations", out[1], "of which past the all-inactive skip", out[2])
