"""Which objects keep per-step gradient-sized tensors alive?  Six headline steps, then every CUDA tensor of >= 20 MB that the
garbage collector can see, grouped by shape, with the types (and for dicts / lists / cells the owners) of what refers to them."""
import gc, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
pc = make_scene(1_000_000, seed=0); pc.train()
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(6):
    bench.one_step(pc, cams[i % 8], pipe, bg, w, 20000, params, None)
    torch.cuda.synchronize()
    print(f"step {i}: allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB", flush=True)
gc.collect()
print(f"after gc.collect(): allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB")
groups = collections.defaultdict(list)
for o in gc.get_objects():
    try:
        if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() >= 20 * 2**20:
            groups[(tuple(o.shape), str(o.dtype))].append(o)
    except Exception:
        pass
def describe(r, depth=0):
    t = type(r).__name__
    if depth < 2 and isinstance(r, (dict, list, tuple)) or t == "cell":
        owners = [describe(x, depth + 1) for x in gc.get_referrers(r) if x is not groups and not isinstance(x, type(sys._getframe()))][:3]
        return f"{t}<-{owners}"
    return t + (":" + getattr(r, "__qualname__", "") if hasattr(r, "__qualname__") else "")
for k, v in sorted(groups.items(), key=lambda kv: -len(kv[1]))[:12]:
    refs = collections.Counter()
    for o in v[:4]:
        for r in gc.get_referrers(o):
            if r is v or r is groups or isinstance(r, type(sys._getframe())):
                continue
            refs[describe(r)] += 1
    print(len(v), k, dict(refs))
print("---- attributes on tensors that hold tensors / objects ----")
seen = collections.Counter()
for o in gc.get_objects():
    try:
        if isinstance(o, torch.Tensor):
            d = getattr(o, "__dict__", None)
            if d:
                for k, v in d.items():
                    seen[(k, type(v).__name__, tuple(o.shape))] += 1
    except Exception:
        pass
for k, n in seen.most_common(20):
    print(n, k)
print("---- who holds the RowSource objects ----")
from contextgs_amd.ctx_ops import RowSource
rs = [o for o in gc.get_objects() if isinstance(o, RowSource)]
print(len(rs), "RowSource objects alive")
def name(o):
    t = type(o).__name__
    if isinstance(o, dict):
        keys = list(o.keys())[:6]
        return f"dict(keys={keys})"
    if isinstance(o, (list, tuple)):
        return f"{t}(len={len(o)})"
    return t
def chain(o, depth, seen, path):
    if depth == 0:
        return
    for r in gc.get_referrers(o):
        if id(r) in seen or r is rs or isinstance(r, type(sys._getframe())) or r is seen:
            continue
        seen.add(id(r))
        p2 = path + [name(r)]
        print("   " * (5 - depth), " <- ", name(r))
        chain(r, depth - 1, seen, p2)
if rs:
    chain(rs[0], 4, {id(rs)}, [])
print("---- autograd ctx objects alive, and who holds a _LevelFusedBackward ----")
cnt = collections.Counter(type(o).__name__ for o in gc.get_objects() if type(o).__name__.endswith("Backward"))
print(cnt.most_common(30))
lf = [o for o in gc.get_objects() if type(o).__name__ == "_LevelFusedBackward"]
if lf:
    chain(lf[0], 3, {id(lf)}, [])
    print("its next_functions:", [type(f[0]).__name__ if f[0] is not None else None for f in lf[0].next_functions][:12])
