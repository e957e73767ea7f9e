import contextlib, io, os, shutil, sys, tempfile, json
sys.path.insert(0, '/root/repo')
import torch
from torch.profiler import profile, ProfilerActivity
from contextgs_amd.synth import make_scene
pc = make_scene(1_000_000, seed=0); pc.eval()
dec = make_scene(1_000_000, seed=0, requires_grad=False); dec.eval()
d = tempfile.mkdtemp(prefix="cgs_bits_")
with contextlib.redirect_stdout(io.StringIO()):
    pc.conduct_encoding(d); dec.conduct_decoding(d)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with contextlib.redirect_stdout(io.StringIO()):
        dec.conduct_decoding(d)
    torch.cuda.synchronize()
prof.export_chrome_trace("/tmp/dec_trace.json")
ev = [e for e in json.load(open("/tmp/dec_trace.json"))["traceEvents"] if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
ev.sort(key=lambda e: e["ts"])
t0 = ev[0]["ts"]
for e in ev:
    if e["dur"] > 250:
        print(f"{(e['ts']-t0)/1e3:8.2f} ms  dur {e['dur']/1e3:7.2f} ms  stream {e.get('args',{}).get('stream')}  {e['name'][:70]}")
print("last event end", (max(e['ts']+e['dur'] for e in ev)-t0)/1e3)
shutil.rmtree(d, ignore_errors=True)
