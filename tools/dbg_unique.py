import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from contextgs_amd import multi_level, context_model
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
calls = []
orig = multi_level.torch_unique_with_indices
def counted(*a, **k):
    calls.append("".join(traceback.format_stack(limit=6)[:-1]))
    return orig(*a, **k)
multi_level.torch_unique_with_indices = counted
context_model.torch_unique_with_indices = counted
pc = make_scene(200_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 640, 360)]
w = torch.randn(3, 360, 640, device="cuda") / (360 * 640)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(6):
    n0 = len(calls)
    bench.one_step(pc, cams[i], pipe, bg, w, 20000, params, False)
    print("step", i, "unique calls", len(calls) - n0)
print(calls[-1])
