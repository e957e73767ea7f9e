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
    import types
    # the model's own optimizer set-up (scene/gaussian_model.py:426-525) with the reference's default learning rates
    pc.spatial_lr_scale = 1.0
    pc.training_setup(types.SimpleNamespace(
        percent_dense=0.01, position_lr_init=0.0, position_lr_final=0.0, position_lr_delay_mult=0.01, position_lr_max_steps=30000,
        offset_lr_init=0.01, offset_lr_final=0.0001, offset_lr_delay_mult=0.01, offset_lr_max_steps=30000,
        mask_lr_init=0.01, mask_lr_final=0.0001, mask_lr_delay_mult=0.01, mask_lr_max_steps=30000,
        feature_lr=0.0075, hyper_latent_lr=0.0075, opacity_lr=0.02, scaling_lr=0.007, rotation_lr=0.002,
        mlp_opacity_lr_init=0.002, mlp_opacity_lr_final=0.00002, mlp_opacity_lr_delay_mult=0.01, mlp_opacity_lr_max_steps=30000,
        mlp_cov_lr_init=0.004, mlp_cov_lr_final=0.004, mlp_cov_lr_delay_mult=0.01, mlp_cov_lr_max_steps=30000,
        mlp_color_lr_init=0.008, mlp_color_lr_final=0.00005, mlp_color_lr_delay_mult=0.01, mlp_color_lr_max_steps=30000,
        latent_codec_lr_init=0.005, latent_codec_lr_final=0.00001, latent_codec_lr_delay_mult=0.33, latent_codec_lr_max_steps=30000,
        mlp_grid_lr_init=0.005, mlp_grid_lr_final=0.00001, mlp_grid_lr_delay_mult=0.01, mlp_grid_lr_max_steps=30000))
    opt = pc.optimizer
    all_params = lambda: [p for g in opt.param_groups for p in g["params"] if p.requires_grad]
    params = all_params()
    mgpu.BIG_TENSOR = N                                       # per-anchor tensors in place, MLPs in the bucket
    sync = mgpu.GradientSync(params, average=True)
    torch.manual_seed(1000 + rank)                            # the ranks' RNG streams differ from here on (as in training)
    if len(sys.argv) > 2 and sys.argv[2] == "seq":
        return collective_sequences(pc, opt, sync, params, cams, pipe, bg, rank)
    # (the fifth step: rank 1's camera sees NO anchor — its backward produces no per-anchor gradient at all, the collective
    #  sequence must stay aligned and the replicas identical; the sixth: back to normal)
    for it, step_sem in enumerate((2000, 5000, 20000, 20000, 20000, 20000)):
        cam = cams[mgpu.view_for(it, len(cams))]
        pc.update_learning_rate(step_sem)
        opt.zero_grad(set_to_none=True)
        vis = prefilter_voxel(cam, pc, pipe, bg)
        if it == 4 and rank == 1:
            vis = torch.zeros_like(vis)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=step_sem)
        loss = (1.0 - pkg["render"]).abs().mean() + 0.01 * pkg["scaling"].prod(dim=1).mean()
        if pkg["bit_per_param"] is not None:
            loss = loss + 0.001 * pkg["bit_per_param"]
        loss.backward()
        sync.finish()
        opt.step()
        pc.training_statis(pkg["viewspace_points"], pkg["neural_opacity"], pkg["visibility_filter"], pkg["selection_mask"], vis)
        mine = digest(params)
        every = mgpu.gather_objects(mine, dst=0)
        if rank == 0:
            assert len(set(every)) == 1, f"replicas diverged after step {it}"
    # densification: statistics summed over ranks, then the same candidates everywhere (shared random draw)
    pc.reduce_statistics()
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
    # ---- the model's own adjust_anchor (grow + prune + optimizer surgery) on the summed statistics: replicas must end
    # up with the same anchors, parameters and Adam moments; thresholds scaled to the 4 x 2 views these statistics hold
    sync.close()
    ratio = (pc.opacity_accum / pc.anchor_demon.clamp(min=1)).squeeze(1)
    seen = pc.anchor_demon.squeeze(1) > 2
    min_op = float(torch.quantile(ratio[seen], 0.2)) if bool(seen.any()) else 0.0
    n0 = int(pc._anchor.shape[0])
    pc.update_init_factor = 16
    pc.adjust_anchor(check_interval=4, success_threshold=0.5, grad_threshold=thr, min_opacity=min_op, reduce_stats=False)
    n1 = int(pc._anchor.shape[0])
    moments = [v for p in all_params() for k, v in sorted(opt.state.get(p, {}).items()) if k != "step"]
    stats = [pc.offset_gradient_accum, pc.offset_denom, pc.opacity_accum, pc.anchor_demon]
    every = mgpu.gather_objects((n1, digest(all_params()), digest(moments), digest(stats)), dst=0)
    if rank == 0:
        assert len(set(every)) == 1, f"replicas differ after adjust_anchor: {every}"
        assert n1 != n0 and pc.offset_denom.shape[0] == n1 * K and pc.opacity_accum.shape[0] == n1
    # ... and keep training on the new anchor set (fresh hooks on the new Parameter objects, level plan rebuilt)
    params = all_params()
    sync = mgpu.GradientSync(params, average=True)
    cam = cams[mgpu.view_for(4, len(cams))]
    opt.zero_grad(set_to_none=True)
    vis = prefilter_voxel(cam, pc, pipe, bg)
    pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=20000)
    ((1.0 - pkg["render"]).abs().mean() + 0.001 * pkg["bit_per_param"]).backward()
    sync.finish()
    opt.step()
    every = mgpu.gather_objects(digest(params), dst=0)
    if rank == 0:
        assert len(set(every)) == 1, "replicas diverged in the step after adjust_anchor"
    sync.close()
    dist.barrier()
    print(f"rank {rank}: replicas identical after 6 optimiser steps (one of them with an empty view on rank 1); grew {grown} anchors identically; "
          f"adjust_anchor {n0} -> {n1} anchors identically", flush=True)
    dist.destroy_process_group()


def collective_sequences(pc, opt, sync, params, cams, pipe, bg, rank):
    """with deferred weight gradients (GradientSync switches them on) and a rank whose view sees nothing,
    both ranks must issue the SAME sequence of collectives (kind, element count, dtype) in every step — a mismatch is a
    deadlock or a silent mis-pairing on RCCL.  Three consecutive steps in each of the three training phases (the active
    parameter set changes at the phase boundaries), the middle step of every phase with an empty view on rank 1."""
    log = []
    real = {k: getattr(mgpu.dist, k) for k in ("all_reduce", "broadcast", "all_gather", "all_gather_into_tensor", "reduce_scatter_tensor")
            if hasattr(mgpu.dist, k)}

    def wrap(name, fn):
        def f(t, *a, **kw):
            first = t[0] if isinstance(t, (list, tuple)) else t
            log.append((name, int(first.numel()), str(first.dtype), str(kw.get("op", a[0] if a and name == "all_reduce" else ""))))
            return fn(t, *a, **kw)
        return f
    for k, fn in real.items():
        setattr(mgpu.dist, k, wrap(k, fn))
    from contextgs_amd import mlp, _lib
    #  where in the sequence the context model's level backward starts: the per-anchor tensor that is FINAL
    # before it — `_mask`: its gradient comes from the expansion's and the rate node's backward, both in front of the level
    # kernels — must be on the wire by then.  (`_offset` / `_scaling` / `_anchor_feat` / `_anchor` are NOT: from iteration
    # 10 000 on they pass through the level kernels (quantisation noise, scene/gaussian_model.py:1610-1616) and their gradients
    # leave the LAST of those kernels.)
    Lc = _lib.lib()
    real_bwd = Lc.cgs_ctx_level_bwd2          # (the entry the training path calls: cgs_ctx_level_bwd + the hyper latents' gradient rows)

    class _LogBwd:
        def __call__(self, *a):
            log.append(("ctxl_bwd", 0, "", ""))
            return real_bwd(*a)
    Lc.cgs_ctx_level_bwd2 = _LogBwd()
    n_mask = int(pc._mask.numel())
    try:
        assert mlp._Deferred.on, "GradientSync did not switch the deferred weight gradients on"
        it = 0
        for phase in (2000, 5000, 20000):
            for k in range(3):
                cam = cams[mgpu.view_for(it, len(cams))]
                pc.update_learning_rate(phase)
                opt.zero_grad(set_to_none=True)
                vis = prefilter_voxel(cam, pc, pipe, bg)
                if k == 1 and rank == 1:
                    vis = torch.zeros_like(vis)
                del log[:]
                pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=True, step=phase)
                loss = (1.0 - pkg["render"]).abs().mean()
                if pkg["bit_per_param"] is not None:
                    loss = loss + 0.001 * pkg["bit_per_param"]
                loss.backward()
                sync.finish()
                mine = list(log)
                opt.step()
                for k_, fn in real.items():          # the comparison itself must not land in the log
                    setattr(mgpu.dist, k_, fn)
                every = mgpu.gather_objects((mine, digest(params)), dst=0)
                for k_, fn in real.items():
                    setattr(mgpu.dist, k_, wrap(k_, fn))
                if phase == 20000 and k == 2:
                    # third step of the context phase: the issue order was re-adopted from this phase's completion order
                    names = [e[0] for e in mine]
                    assert "ctxl_bwd" in names, "the level backward did not run in the context phase"
                    first_bwd = names.index("ctxl_bwd")
                    mask_at = [i for i, e in enumerate(mine) if e[0] == "all_reduce" and e[1] == n_mask and "float32" in e[2]]
                    assert mask_at and mask_at[0] < first_bwd, (
                        f"rank {rank}: `_mask`'s all-reduce (entry {mask_at}) was not issued before the level backward (entry {first_bwd})")
                    print(f"rank {rank}: `_mask` all-reduce issued at entry {mask_at[0]}, level backward starts at entry {first_bwd}", flush=True)
                if rank == 0:
                    assert every[0][0] == every[1][0], f"phase {phase} step {k}: collective sequences differ:\n{every[0][0]}\n{every[1][0]}"
                    assert len(every[0][0]) >= 2 and every[0][1] == every[1][1], f"phase {phase} step {k}: replicas diverged"
                it += 1
    finally:
        for k_, fn in real.items():
            setattr(mgpu.dist, k_, fn)
        Lc.cgs_ctx_level_bwd2 = real_bwd
    sync.close()
    dist.barrier()
    print(f"rank {rank}: identical collective sequences in 9 steps (3 phases x 3, one empty view per phase)", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
