"""GPU-side attribution of one full training-view step by torch op (aten::*) and by our autograd
Functions: python tools/torch_prof.py [--anchors N] -> gpurun_out/torch_prof.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=1_000_000)
ap.add_argument("--step", type=int, default=20000)
ap.add_argument("--stacks", action="store_true")
a = ap.parse_args()
pc = make_scene(a.anchors, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(3):
    bench.one_step(pc, cams[i], pipe, bg, w, a.step, params, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=a.stacks) as prof:
    for i in range(4):
        bench.one_step(pc, cams[i], pipe, bg, w, a.step, params, None)
    torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/torch_prof.txt", "w") as f:
    f.write(prof.key_averages(group_by_stack_n=6 if a.stacks else 0).table(sort_by="self_cuda_time_total", row_limit=70,
                                       max_name_column_width=60, max_src_column_width=110))
print(open("gpurun_out/torch_prof.txt").read()[:200])
