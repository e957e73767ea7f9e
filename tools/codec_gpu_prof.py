"""GPU-side attribution of conduct_encoding / conduct_decoding on the bench scene: aten ops and kernels by device time."""
import contextlib, io, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from contextgs_amd.synth import make_scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pc = make_scene(N, seed=0); pc.eval()
dec = make_scene(N, seed=0, requires_grad=False); dec.eval()
d = tempfile.mkdtemp(prefix="cgs_bits_")
with contextlib.redirect_stdout(io.StringIO()):
    pc.conduct_encoding(d); dec.conduct_decoding(d)
for name, fn in (("encode", lambda: pc.conduct_encoding(d)), ("decode", lambda: dec.conduct_decoding(d))):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        with contextlib.redirect_stdout(io.StringIO()):
            fn()
        torch.cuda.synchronize()
    rows = [(e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:70]) for e in prof.key_averages(group_by_input_shape=True)
            if e.self_device_time_total > 0]
    rows.sort(reverse=True)
    print(f"==== {name}: {sum(r[0] for r in rows) / 1e3:.1f} ms of device time")
    for us, n, k, sh in rows[:28]:
        print(f"{us / 1e3:8.2f} ms x{n:3d}  {k[:40]:40s} {sh}")
shutil.rmtree(d, ignore_errors=True)
