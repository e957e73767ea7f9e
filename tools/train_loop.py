"""A miniature of the reference's train.py loop (train.py:140-260) driven entirely by contextgs_amd.model.GaussianModel:
training_setup -> per iteration update_learning_rate / prefilter_voxel / render / loss / backward / optimizer step /
training_statis, adjust_anchor every `interval` iterations between `update_from` and `update_until`, through the three
training phases (plain, noise, context model).  Prints loss / anchors / bits and checks that everything stays finite.
usage: python tools/train_loop.py [iterations=360] [anchors=100000]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd.loss_utils import training_image_loss, scaling_reg, mask_reg
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 360
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
pc = make_scene(N, seed=0)
pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
cams = [c.to_torch("cuda") for c in orbit_cameras(8, 640, 360)]
with torch.no_grad():
    pc.eval()
    gts = [(render(c, pc, pipe, bg, visible_mask=prefilter_voxel(c, pc, pipe, bg))["render"] * 0.7 + 0.1).clamp(0, 1) for c in cams]
pc.train()
A = types.SimpleNamespace(
    percent_dense=0.01, position_lr_init=0.0, position_lr_final=0.0, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
    offset_lr_init=0.01, offset_lr_final=0.0001, offset_lr_delay_mult=0.01, offset_lr_max_steps=30000,
    mask_lr_init=0.01, mask_lr_final=0.0001, mask_lr_delay_mult=0.01, mask_lr_max_steps=30000,
    feature_lr=0.0075, hyper_latent_lr=0.0075, opacity_lr=0.02, scaling_lr=0.007, rotation_lr=0.002,
    mlp_opacity_lr_init=0.002, mlp_opacity_lr_final=0.00002, mlp_opacity_lr_delay_mult=0.01, mlp_opacity_lr_max_steps=30000,
    mlp_cov_lr_init=0.004, mlp_cov_lr_final=0.004, mlp_cov_lr_delay_mult=0.01, mlp_cov_lr_max_steps=30000,
    mlp_color_lr_init=0.008, mlp_color_lr_final=0.00005, mlp_color_lr_delay_mult=0.01, mlp_color_lr_max_steps=30000,
    latent_codec_lr_init=0.005, latent_codec_lr_final=0.00001, latent_codec_lr_delay_mult=0.33, latent_codec_lr_max_steps=30000,
    mlp_grid_lr_init=0.005, mlp_grid_lr_final=0.00001, mlp_grid_lr_delay_mult=0.01, mlp_grid_lr_max_steps=30000)
pc.spatial_lr_scale = 1.0
pc.training_setup(A)
pc.update_init_factor = 16
# the schedule of train.py compressed: phase switches at 1/3 and 2/3 of the run, statistics from 10 %, densification
# every `interval` iterations until 80 %
third = iters // 3
sem = lambda it: 1000 if it < third else (5000 if it < 2 * third else 20000)
start_stat, update_from, interval, update_until = iters // 10, iters // 5, max(20, iters // 9), int(iters * 0.8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(1, iters + 1):
    c, gt = cams[it % 8], gts[it % 8]
    step = sem(it)
    pc.update_learning_rate(step)
    vis = prefilter_voxel(c, pc, pipe, bg)
    pkg = render(c, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step)
    loss = training_image_loss(pkg["render"], gt, 0.2)[0] + 0.01 * scaling_reg(pkg["scaling"])
    if pkg["bit_per_param"] is not None:
        loss = loss + 0.001 * pkg["bit_per_param"] + 5e-4 * mask_reg(pc._mask)
    pc.optimizer.zero_grad(set_to_none=True)
    loss.backward()
    with torch.no_grad():
        if start_stat < it < update_until:
            pc.training_statis(pkg["viewspace_points"], pkg["neural_opacity"], pkg["visibility_filter"], pkg["selection_mask"], vis)
            if it > update_from and it % interval == 0:
                n0 = pc._anchor.shape[0]
                pc.adjust_anchor(check_interval=interval, success_threshold=0.8, grad_threshold=2e-4, min_opacity=0.005)
                print(f"   it {it}: adjust_anchor {n0} -> {pc._anchor.shape[0]} anchors")
        pc.optimizer.step()
    if it % max(1, iters // 12) == 0 or it == iters:
        bpp = pkg["bit_per_param"]
        print(f"it {it:4d} sem {step:5d} loss {float(loss):.5f} bpp {float(bpp) if bpp is not None else float('nan'):.3f} "
              f"anchors {pc._anchor.shape[0]} gaussians {pkg['radii'].shape[0]} alloc {torch.cuda.memory_allocated() / 2**20:.0f} MiB")
torch.cuda.synchronize()
params = [p for g in pc.optimizer.param_groups for p in g["params"]]
ok = all(torch.isfinite(p).all().item() for p in params)
print(f"{iters / (time.perf_counter() - t0):.1f} it/s, peak {torch.cuda.max_memory_allocated() / 2**20:.0f} MiB, finite {ok}")
# the trained model still encodes and decodes bit-exactly
import tempfile
pc.eval()
with tempfile.TemporaryDirectory() as d, torch.no_grad():
    pc.update_anchor_bound() if False else None
    pc.conduct_encoding(d)
    from contextgs_amd.model import GaussianModel
    q = GaussianModel(voxel_size=pc.voxel_size)
    q.conduct_decoding(d)
    print("decoded anchors", q._anchor.shape[0], "of", int(pc.get_mask_anchor.sum()) if hasattr(pc, "get_mask_anchor") else "?")
assert ok
