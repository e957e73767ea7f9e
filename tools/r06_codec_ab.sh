# container version 2: ranged coder launches (product) against one launch (CGS_RANGED_ENCODE=0): codec tests, trace, rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_codec_gpu.py tests/test_configs_gpu.py tests/test_dist_codec_gpu.py -x -q 2>&1 | tail -3)
for e in 1 0 1 0; do
CGS_RANGED_ENCODE=$e CGS_CODEC_TRACE=1 CGS_CONTAINER_VERSION=2 timeout -k 5 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | grep "encoding time\|range of\|coder launch done\|files written\|levels done" | tail -12
import os, sys, tempfile, shutil
sys.path.insert(0, os.getcwd())
import torch
from contextgs_amd.codec_driver import conduct_encoding
from contextgs_amd.synth import make_scene
pc = make_scene(1_000_000, seed=0, requires_grad=False); pc.eval()
d = tempfile.mkdtemp(prefix="cgs_tr_")
for i in range(4):
    conduct_encoding(pc, d, container_version=2)
    torch.cuda.synchronize()
shutil.rmtree(d, ignore_errors=True)
PY
echo "---- ranged=$e above"
done
