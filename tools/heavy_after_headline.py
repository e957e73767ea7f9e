import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
from contextgs_amd.rasterizer import last_call
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
for vox, steps in ((None, 12), (0.01, 30)):
    pc = make_scene(1_000_000, seed=0, **({"voxel_size": vox} if vox else {})); pc.train()
    params = [p for p in pc.parameters() if p.requires_grad]
    for i in range(steps):
        torch.cuda.synchronize(); t = time.perf_counter()
        bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"vox {vox} step {i:2d}: {dt*1e3:7.2f} ms R {int(last_call['num_rendered'])} carved {int(last_call['bin_R'])} reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB alloc retries {torch.cuda.memory_stats().get('num_alloc_retries')}")
    del pc, params
    torch.cuda.empty_cache()
