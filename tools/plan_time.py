"""Cost of rebuilding the context model's level plan (what every densification step invalidates) at 1 M anchors."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from contextgs_amd.synth import make_scene
from contextgs_amd import context_model as cm
pc = make_scene(1_000_000, seed=0); pc.train()
with torch.no_grad():
    anchor = pc.get_anchor
    mask = pc.get_mask_anchor
    pc.level_scale = cm.find_divide_scale(pc, anchor[mask], pc.target_ratio, pc.level_num)
    for name, fn in (("find_divide_scale", lambda: cm.find_divide_scale(pc, anchor[mask], pc.target_ratio, pc.level_num)),
                     ("level plan (uncached)", lambda: cm._level_plan_uncached(pc, anchor, mask)),
                     ("cached plan build", lambda: (setattr(pc, "_level_cache", None), cm._cached_plan(pc, anchor, mask)))):
        for _ in range(2):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            print(f"{name}: {1e3*(time.perf_counter()-t):.1f} ms")
