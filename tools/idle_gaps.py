"""Where does the GPU idle inside a training step?  (torch profiler, no python stacks: ~1 us per launch of overhead.)

For the last steps of a short run: wall time, GPU busy time, and every idle gap above --min-us with the kernel before
it, the kernel after it and the host-side op (aten / autograd node / runtime call) that issued the kernel after it.
python tools/idle_gaps.py [--anchors N] [--min-us 15] -> gpurun_out/idle_gaps.txt"""
import argparse, bisect, collections, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=1_000_000)
ap.add_argument("--step", type=int, default=20000)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--min-us", type=float, default=15.0)
a = ap.parse_args()
pc = make_scene(a.anchors, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(5):
    bench.one_step(pc, cams[i % 8], pipe, bg, w, a.step, params, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(a.steps):
        with torch.profiler.record_function(f"STEP{i}"):
            bench.one_step(pc, cams[i % 8], pipe, bg, w, a.step, params, None)
    torch.cuda.synchronize()
tmp = os.path.join(tempfile.gettempdir(), "idle_gaps_trace.json")
prof.export_chrome_trace(tmp)
ev = [e for e in json.load(open(tmp))["traceEvents"] if e.get("ph") == "X"]
gpu = sorted((e for e in ev if e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")), key=lambda e: e["ts"])
runtime = {e["args"]["correlation"]: e for e in ev if e.get("cat") in ("cuda_runtime", "cuda_driver") and "correlation" in e.get("args", {})}
ops = collections.defaultdict(list)
for e in ev:
    if e.get("cat") in ("cpu_op", "user_annotation"):
        ops[(e["pid"], e["tid"])].append(e)
for k in ops:
    ops[k].sort(key=lambda e: (e["ts"], -e["dur"]))
op_starts = {k: [e["ts"] for e in v] for k, v in ops.items()}
steps = sorted((e for e in ev if e.get("cat") == "user_annotation" and e["name"].startswith("STEP")), key=lambda e: e["ts"])


def issuer(g):
    rt = runtime.get(g.get("args", {}).get("correlation"))
    if rt is None:
        return "?"
    key = (rt["pid"], rt["tid"])
    i = bisect.bisect_right(op_starts.get(key, []), rt["ts"])
    names = []
    for j in range(i - 1, max(-1, i - 400), -1):
        e = ops[key][j]
        if e["ts"] + e["dur"] >= rt["ts"] and not e["name"].startswith("STEP"):
            names.append(e["name"].replace("autograd::engine::evaluate_function: ", "bwd:"))
    return (" < ".join(names[:3]) or rt["name"])[:90]


short = lambda n: n.split("(")[0].replace("void ", "").replace("at::native::", "")[:44]
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/idle_gaps.txt", "w") as f:
    # a step's GPU work = kernels issued by runtime calls inside the STEP annotation
    for si, st in enumerate(steps):
        t0, t1 = st["ts"], st["ts"] + st["dur"]
        mine = [g for g in gpu if (lambda rt: rt is not None and t0 <= rt["ts"] < t1)(runtime.get(g.get("args", {}).get("correlation")))]
        if not mine:
            continue
        busy, cur, gaps = 0.0, mine[0]["ts"], []
        for p, g in zip([None] + mine[:-1], mine):
            if g["ts"] > cur and p is not None:
                gaps.append((g["ts"] - cur, short(p["name"]), short(g["name"]), issuer(g)))
            busy += max(0.0, g["ts"] + g["dur"] - max(g["ts"], cur))
            cur = max(cur, g["ts"] + g["dur"])
        span = cur - mine[0]["ts"]
        nxt = steps[si + 1]["ts"] if si + 1 < len(steps) else None
        f.write(f"step {si}: host {st['dur'] / 1e3:.3f} ms, GPU first->last {span / 1e3:.3f} ms, busy {busy / 1e3:.3f} ms, "
                f"idle {(span - busy) / 1e3:.3f} ms in {len(gaps)} gaps, {len(mine)} launches\n")
        small = sum(g[0] for g in gaps if g[0] < a.min_us)
        f.write(f"   gaps < {a.min_us:.0f} us: {small / 1e3:.3f} ms total ({sum(1 for g in gaps if g[0] < a.min_us)} gaps)\n")
        if si == len(steps) - 1:
            for d, before, after, who in gaps:
                if d >= a.min_us:
                    f.write(f"   {d:7.1f} us  after {before:44s} before {after:44s} | {who}\n")
print(open("gpurun_out/idle_gaps.txt").read())
