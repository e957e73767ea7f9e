"""Why does the heavy-pair leg of bench.py sometimes run at 19 ms per step instead of 7.8?  The bench's sequence in one process —
the headline scene for a few steps, then the heavy-pair scene — with per-step wall time, the host's enqueue time, device
allocations and reserved memory.  python tools/heavy_diag.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("ALLOC_SET"):
    torch.cuda.memory._set_allocator_settings(os.environ["ALLOC_SET"])
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
from contextgs_amd.rasterizer import last_call
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
def run(pc, n, tag):
    params = [p for p in pc.parameters() if p.requires_grad]
    for i in range(n):
        st0 = torch.cuda.memory_stats()
        torch.cuda.synchronize(); t = time.perf_counter()
        bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
        th = time.perf_counter() - t
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        st1 = torch.cuda.memory_stats()
        if i >= n - 12:
            print(f"{tag} step {i:2d}: {dt * 1e3:7.2f} ms (host loop {th * 1e3:6.2f})  R {int(last_call['num_rendered']):>10d} carved {int(last_call['bin_R']):>10d} "
                  f"device allocs +{st1['num_device_alloc'] - st0['num_device_alloc']} frees +{st1['num_device_free'] - st0['num_device_free']} "
                  f"retries +{st1['num_alloc_retries'] - st0['num_alloc_retries']} reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
pc = make_scene(1_000_000, seed=0); pc.train()
run(pc, 30, "headline")
pc2 = make_scene(1_000_000, seed=0, voxel_size=0.01); pc2.train()
run(pc2, 30, "heavy   ")

import collections
free, used = collections.Counter(), collections.Counter()
for seg in torch.cuda.memory_snapshot():
    for b in seg["blocks"]:
        (free if b["state"] == "inactive" else used)[round(b["size"] / 2**20)] += 1
print("free blocks (MiB: count), largest first:", sorted(free.items(), reverse=True)[:24])
print("live blocks (MiB: count), largest first:", sorted(used.items(), reverse=True)[:24])
