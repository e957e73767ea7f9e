"""What would storing the anchors in CODING order buy?  Times the headline step on the synthetic scene as generated (anchors in
random order: every per-anchor access of the context model goes through the coding permutation) and on the same scene with the
per-anchor tensors physically permuted into coding order (same kernels, same code path: the permutation is then the identity
VALUES but still passed as an index array).  -> gpurun_out/reorder_probe.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from contextgs_amd import context_model as cm
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pc = make_scene(N, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
orig_cached = cm._cached_plan
def no_identity(pc_, a, m):
    c = orig_cached(pc_, a, m); c["identity"] = False; return c
cm._cached_plan = no_identity

def timeit(tag, steps=40):
    for i in range(10): bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag}: {dt:.3f} ms/step  {1e3 / dt:.1f} views/s", flush=True)
    return dt

out = []
out.append(("random order", timeit("anchors in generated (random) order")))
perm = pc._level_cache["perm"].clone()
assert perm.shape[0] == N
with torch.no_grad():
    for name in ("_anchor", "_offset", "_mask", "_anchor_feat", "_hyper_latent", "_scaling", "_rotation", "_opacity"):
        t = getattr(pc, name, None)
        if t is not None and t.shape[0] == N:
            t.data = t.data.index_select(0, perm).contiguous()
pc._level_cache = None
out.append(("coding order", timeit("anchors stored in coding order")))
p2 = pc._level_cache["perm"]
print("new permutation is the identity:", bool((p2 == torch.arange(N, device=p2.device)).all()))
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/reorder_probe.txt", "w").write("\n".join(f"{a}: {b:.3f} ms/step" for a, b in out) + "\n")
