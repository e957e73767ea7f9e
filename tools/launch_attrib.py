"""Which call site does every launch of a training step come from?

Runs a few steps of the bench's full step under the torch profiler (python stacks on), then walks the trace: every GPU
kernel / memcpy / memset is tied through its correlation id to the runtime call that launched it, and that call to
  * the autograd node being evaluated (backward thread), and
  * the innermost python frame inside this repository (forward thread, and python-implemented backward nodes).
Output gpurun_out/launch_attrib.txt: launches per step and GPU microseconds per step by (node, frame), largest first,
with the kernel names seen there.  python tools/launch_attrib.py [--anchors N] [--min-us 0]"""
import argparse, bisect, collections, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

ap = argparse.ArgumentParser()
ap.add_argument("--anchors", type=int, default=1_000_000)
ap.add_argument("--step", type=int, default=20000)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
pc = make_scene(a.anchors, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
w = torch.randn(3, 1080, 1920, device="cuda") / (1080 * 1920)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(3):
    bench.one_step(pc, cams[i], pipe, bg, w, a.step, params, None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for i in range(a.steps):
        bench.one_step(pc, cams[i], pipe, bg, w, a.step, params, None)
    torch.cuda.synchronize()
tmp = os.path.join(tempfile.gettempdir(), "launch_attrib_trace.json")
prof.export_chrome_trace(tmp)
ev = json.load(open(tmp))["traceEvents"]

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gpu, runtime, spans = [], {}, collections.defaultdict(list)
for e in ev:
    if e.get("ph") != "X":
        continue
    cat = e.get("cat", "")
    if cat in ("kernel", "gpu_memcpy", "gpu_memset"):
        gpu.append(e)
    elif cat in ("cuda_runtime", "cuda_driver"):
        c = e.get("args", {}).get("correlation")
        if c is not None:
            runtime[c] = e
    elif cat in ("cpu_op", "python_function", "user_annotation"):
        spans[(e["pid"], e["tid"])].append(e)
for k in spans:
    spans[k].sort(key=lambda e: (e["ts"], -e["dur"]))
starts = {k: [e["ts"] for e in v] for k, v in spans.items()}


def enclosing(rt):
    """(autograd node, innermost repo python frame) around a runtime call"""
    key = (rt["pid"], rt["tid"])
    lst = spans.get(key, [])
    i = bisect.bisect_right(starts.get(key, []), rt["ts"])
    node, frame, op, t = None, None, None, rt["ts"]
    # walk back over candidates that started before; containment test (bounded look-back is enough: nesting depth)
    depth_seen = 0
    for j in range(i - 1, max(-1, i - 4000), -1):
        e = lst[j]
        if e["ts"] + e["dur"] < t:
            continue
        name = e["name"]
        if e["cat"] == "python_function":
            if frame is None and ("contextgs_amd/" in name or "bench.py" in name):
                frame = name.replace(REPO + "/", "")
        elif name.startswith("autograd::engine::evaluate_function:"):
            node = name.split(":", 3)[-1].strip()
        elif e["cat"] == "cpu_op" and op is None:
            op = name + str(e.get("args", {}).get("Input Dims", ""))[:60]
        depth_seen += 1
    return node, frame, op


agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for g in gpu:
    rt = runtime.get(g.get("args", {}).get("correlation"))
    node, frame, op = enclosing(rt) if rt is not None else (None, None, None)
    k = (node or "-", frame or "-")
    agg[k][0] += 1
    agg[k][1] += g["dur"]
    kn = g["name"].split("(")[0].replace("void ", "")[:48]
    agg[k][2][(op or "?") + " -> " + kn[:28] if kn.startswith(("at::", "Mem", "rocprim")) or not kn else kn] += 1
os.makedirs("gpurun_out", exist_ok=True)
n = float(a.steps)
with open("gpurun_out/launch_attrib.txt", "w") as f:
    tot_l = sum(v[0] for v in agg.values()) / n
    tot_t = sum(v[1] for v in agg.values()) / n
    f.write(f"# {tot_l:.0f} launches/step, {tot_t:.0f} us GPU/step; columns: launches/step  us/step  autograd node | python frame\n")
    for (node, frame), (cnt, us, names) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        f.write(f"{cnt / n:6.1f} {us / n:9.1f}  {node} | {frame}\n")
        f.write("                  " + "\n                  ".join(f"{k} x{v / n:.0f}" for k, v in names.most_common(12)) + "\n")
print(open("gpurun_out/launch_attrib.txt").read()[:1500])
