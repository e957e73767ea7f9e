cd $GRAFT_REPO_ROOT
CGS_CODEC_TRACE=1 CGS_CONTAINER_VERSION=2 timeout -k 5 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "encode +\|decode +\|time:" | tail -75
import os, sys, tempfile, shutil
sys.path.insert(0, os.getcwd())
import torch
from contextgs_amd.codec_driver import conduct_encoding
from contextgs_amd.synth import make_scene
pc = make_scene(1_000_000, seed=0, requires_grad=False); pc.eval()
d = tempfile.mkdtemp(prefix="cgs_tr_")
for i in range(3):
    conduct_encoding(pc, d, container_version=2)
    dec = make_scene(1_000_000, seed=0, requires_grad=False); dec.eval()
    dec.conduct_decoding(d)
    torch.cuda.synchronize()
shutil.rmtree(d, ignore_errors=True)
PY
