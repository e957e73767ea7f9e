"""Host-side profile of conduct_encoding / conduct_decoding on the bench scene (cProfile, wall time by function)."""
import cProfile, contextlib, io, os, pstats, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.synth import make_scene

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pc = make_scene(N, seed=0); pc.eval()
dec = make_scene(N, seed=0, requires_grad=False); dec.eval()
d = tempfile.mkdtemp(prefix="cgs_bits_")
with contextlib.redirect_stdout(io.StringIO()):
    pc.conduct_encoding(d); dec.conduct_decoding(d)
for name, fn in (("encode", lambda: pc.conduct_encoding(d)), ("decode", lambda: dec.conduct_decoding(d))):
    pr = cProfile.Profile()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    print(f"==== {name}: {time.perf_counter() - t0:.3f} s")
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[5:60]))
shutil.rmtree(d, ignore_errors=True)
