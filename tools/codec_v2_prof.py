"""One warm + one profiled conduct_encoding / conduct_decoding of the bench scene for a container version (argv[1], default 2):
run under `rocprofv3 --kernel-trace --stats` (tools/r04_codec_prof.sh) for the per-kernel device time of the container codec."""
import os, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.codec_driver import conduct_encoding
from contextgs_amd.synth import make_scene
version = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
pc = make_scene(N, seed=0, requires_grad=False); pc.eval()
d = tempfile.mkdtemp(prefix="cgs_prof_")
try:
    for _ in range(2):
        conduct_encoding(pc, d, container_version=version)
        dec = make_scene(N, seed=0, requires_grad=False); dec.eval()
        dec.conduct_decoding(d)
        torch.cuda.synchronize()
finally:
    shutil.rmtree(d, ignore_errors=True)
