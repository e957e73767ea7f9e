"""Worker of tests/test_dist_train_gpu.py: the control flow of `bench.py --gpus 2` (views shard over ranks, replicated
parameters, gradient all-reduce) carried through a few REAL optimiser steps and one densification round on world size 2
(both ranks on cuda:0, gloo): after every step, and after growing anchors from the all-reduced statistics with the
shared random draw, the replicas must hold bit-identical parameters and the same number of anchors."""
import hashlib
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from contextgs_amd import densify                            # noqa: E402
from contextgs_amd import dist as mgpu                       # noqa: E402
from contextgs_amd.renderer import prefilter_voxel, render   # noqa: E402
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras   # noqa: E402


def digest(tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main():
    N = int(sys.argv[1])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    pc = make_scene(N, seed=0)                                # identical replicas (same seed on every rank)
    pc.train()
    K = pc.n_offsets
    cams = [c.to_torch("cuda") for c in orbit_cameras(8, 320, 180)]
    pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
    per_anchor = [pc._anchor, pc._offset, pc._mask, pc._anchor_feat, pc._hyper_latent, pc._scaling]
    params = [p for p in pc.parameters() if p.requires_grad]
    mgpu.BIG_TENSOR = N                                       # per-anchor tensors in place, MLPs in the bucket
    opt = torch.optim.Adam([{"params": per_anchor[1:], "lr": 1e-3}, {"params": [pc._anchor], "lr": 0.0},
                            {"params": [p for p in params if all(p is not q for q in per_anchor)], "lr": 1e-3}])
    sync = mgpu.GradientSync(params, average=True)
    pc.opacity_accum = torch.zeros(N, 1, device="cuda")
    pc.anchor_demon = torch.zeros(N, 1, device="cuda")
    pc.offset_gradient_accum = torch.zeros(N * K, 1, device="cuda")
    pc.offset_denom = torch.zeros(N * K, 1, device="cuda")
    torch.manual_seed(1000 + rank)                            # the ranks' RNG streams differ from here on (as in training)
    for it, step_sem in enumerate((2000, 5000, 20000, 20000)):
        cam = cams[mgpu.view_for(it, len(cams))]
        opt.zero_grad(set_to_none=True)
        vis = prefilter_voxel(cam, pc, pipe, bg)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step_sem)
        loss = (1.0 - pkg["render"]).abs().mean() + 0.01 * pkg["scaling"].prod(dim=1).mean()
        if pkg["bit_per_param"] is not None:
            loss = loss + 0.001 * pkg["bit_per_param"]
        loss.backward()
        sync.finish()
        opt.step()
        densify.training_statis(pc, pkg["viewspace_points"], pkg["neural_opacity"], pkg["visibility_filter"],
                                pkg["selection_mask"], vis)
        mine = digest(params)
        every = mgpu.gather_objects(mine, dst=0)
        if rank == 0:
            assert len(set(every)) == 1, f"replicas diverged after step {it}"
    # densification: statistics summed over ranks, then the same candidates everywhere (shared random draw)
    stats = [pc.offset_gradient_accum, pc.offset_denom, pc.opacity_accum, pc.anchor_demon]
    mgpu.allreduce_stats(stats)
    grads = (pc.offset_gradient_accum / pc.offset_denom.clamp(min=1)).squeeze(1)
    offset_mask = (pc.offset_denom > 0).squeeze(1)
    thr = float(torch.quantile(grads[offset_mask], 0.7)) if bool(offset_mask.any()) else 0.0
    rounds = densify.growing_rounds(pc._anchor.detach(), pc._offset.detach(), pc._scaling.detach(), pc._anchor_feat.detach(),
                                    pc._hyper_latent.detach(), pc.x_bound_min, pc.x_bound_max, grads, thr, offset_mask,
                                    pc.voxel_size, K, update_depth=3, init_factor=16, hier=4)
    grown = sum(int(r["anchor"].shape[0]) for r in rounds)
    sig = digest([r["anchor"] for r in rounds] + [r["anchor_feat"] for r in rounds]) if rounds else "none"
    every = mgpu.gather_objects((grown, sig), dst=0)
    if rank == 0:
        assert len(set(every)) == 1, f"replicas grew different anchors: {every}"
        assert every[0][0] > 0, "the test scene must actually grow anchors"
    sync.close()
    dist.barrier()
    print(f"rank {rank}: replicas identical after 4 optimiser steps; grew {grown} anchors identically", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
