"""Where the HOST time of a training step goes (the step is near host-bound: ~9 ms of Python per ~10 ms of GPU work):
cProfile over N headline steps, sorted by own time and by cumulative time.  GPU box.
usage: python tools/host_profile.py [steps] > gpurun_out/host_profile.txt"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pc = make_scene(1_000_000, seed=0)
pc.train()
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
w = torch.randn(3, 1080, 1920, device="cuda")
params = [p for p in pc.parameters() if p.requires_grad]
f = lambda i: bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
for i in range(8):
    f(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    f(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
total = time.perf_counter() - t0
print(f"{steps} steps: host enqueue loop {host / steps * 1e3:.2f} ms/step, with final drain {total / steps * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    f(i)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    txt = s.getvalue()
    print(f"==== sorted by {key} (totals over {steps} steps) ====")
    print("\n".join(l[:170] for l in txt.splitlines()[4:60]))
