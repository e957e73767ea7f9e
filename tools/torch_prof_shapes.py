"""aten-level attribution of one training step grouped by input shapes (which call site a launch comes from)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
pc = make_scene(1_000_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(3):
    bench.one_step(pc, cams[i], pipe, bg, w, 20000, params, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(4):
        bench.one_step(pc, cams[i], pipe, bg, w, 20000, params, None)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        rows.append((e.self_device_time_total / 4, e.count / 4, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/torch_prof_shapes.txt", "w") as f:
    tot = 0
    for us, n, k, sh in rows[:70]:
        tot += us
        f.write(f"{us:8.1f} us/step  x{n:5.1f}  {k:24s} {sh}\n")
    f.write(f"total listed {tot:.0f} us/step; all aten {sum(r[0] for r in rows):.0f} us/step\n")
print(open("gpurun_out/torch_prof_shapes.txt").read())
