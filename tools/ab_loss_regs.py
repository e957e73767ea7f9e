"""Same-box A/B of the training step with the full loss of train.py:199-209: the two regularisers as torch expressions
(`scaling.prod(dim=1).mean()`, `torch.mean(torch.sigmoid(_mask))`) against loss_utils.scaling_reg / mask_reg.
python tools/ab_loss_regs.py [reps]  ->  gpurun_out/ab_loss_regs.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from contextgs_amd import loss_utils
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pc = make_scene(1_000_000, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 1920, 1080)]
gts = [torch.rand(3, 1080, 1920, device="cuda") for _ in range(8)]
params = [p for p in pc.parameters() if p.requires_grad]
fused = (loss_utils.scaling_reg, loss_utils.mask_reg)
plain = (lambda s: s.prod(dim=1).mean(), lambda m: torch.mean(torch.sigmoid(m)))


def run(ops, steps=40):
    loss_utils.scaling_reg, loss_utils.mask_reg = ops
    for i in range(5):
        bench.one_step(pc, cams[i % 8], pipe, bg, None, 20000, params, None, gt=gts[i % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps):
        bench.one_step(pc, cams[i % 8], pipe, bg, None, 20000, params, None, gt=gts[i % 8])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/ab_loss_regs.txt", "w") as f:
    for r in range(reps):
        for name, ops in (("torch expressions", plain), ("scaling_reg / mask_reg", fused)):
            ms = run(ops)
            line = "rep=%d %-24s %.3f ms/step  %.2f views/s" % (r + 1, name, ms, 1e3 / ms)
            print(line); f.write(line + "\n")
