"""A 28-iteration TRAINING TRAJECTORY against the reference's own loop: tests/golden/trajectory_n3000.npz
holds what the reference's `training_setup` / `update_learning_rate` / `prefilter_voxel` / `render` / `l1_loss` / `ssim` /
`training_statis` / `adjust_anchor` + torch's Adam produced over 28 consecutive optimizer steps across the step-3000 and
step-10000 phase switches with one densification round (tools/make_trajectory_golden.py; the CUDA rasterizer behind the
reference's call sites is oracle/raster_ref.c there — parity unpinned for that part).  Here the drop-in path replays the loop of
train.py:144-256 with the same arguments, cameras, targets and random draws: LR schedules, optimizer surgery, statistics
accumulation, the three training phases and the rasterizer + context model + loss kernels are pinned TOGETHER."""
import os
import types

import numpy as np
import pytest
import torch

import golden_inputs as gi
import trajectory_common as tc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "trajectory_n3000.npz")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _model():
    from contextgs_amd.model import GaussianModel
    pc = GaussianModel(feat_dim=gi.D, n_offsets=gi.K, voxel_size=0.01, level_num=gi.LEVELS, target_ratio=0.2)
    sd = pc.state_dict()
    for k, v in gi.mlp_weights(tc.SEED, positive_scales=True).items():
        sd[k] = T(v)
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(tc.N, tc.SEED)
    pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
    pc.update_anchor_bound()
    pc.train()
    return pc


def test_training_trajectory_matches_the_reference_loop(monkeypatch):
    from contextgs_amd import context_model as cm
    from contextgs_amd import ctx_ops
    from contextgs_amd.loss_utils import l1_loss, mask_reg, scaling_reg, ssim
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe
    g = np.load(GOLD)
    opt = types.SimpleNamespace(**{str(k): float(v) for k, v in zip(g["args_names"], g["args_values"])})
    for k in ("iterations", "start_stat", "update_from", "update_interval", "update_until"):
        setattr(opt, k, int(getattr(opt, k)))
    pc = _model()
    pc.spatial_lr_scale = tc.SPATIAL_LR_SCALE
    pc.update_init_factor = tc.UPDATE_INIT_FACTOR
    pc.training_setup(opt)
    cams = tc.cameras("cuda")
    gts = tc.gt_images(cams, "cuda")
    pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")

    state = {"it": 0, "seeds": iter(())}
    monkeypatch.setattr(ctx_ops, "next_seed", lambda: next(state["seeds"]))

    def provider(anchor, mab):
        m = T(tc.choose_draw(state["it"], int(anchor.shape[0])) <= 0.15)
        return m & mab if mab is not None else m

    monkeypatch.setattr(cm, "choose_mask_provider", provider)
    its = [int(v) for v in g["it"]]
    assert its == tc.ITERATIONS
    worst = {"loss": 0.0, "bpp": 0.0, "sum": 0.0}
    names = [str(n) for n in g["sum_names"]]
    for idx, it in enumerate(its):
        state["it"] = it
        s = tc.seeds(it)
        # draw order of the product: mid phase one seed (the three tensors of noise_quant), context phase the hyper prior then
        # one per level, coarsest first
        state["seeds"] = iter([s["mid"]] if it <= 10000 else [s["hyper"]] + s["levels"])
        pc.update_learning_rate(it)
        cam, gt = cams[idx % len(cams)], gts[idx % len(cams)]
        vis = prefilter_voxel(cam, pc, pipe, bg)
        retain = it < opt.update_until and it >= 0
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=retain, step=it)
        if 3000 < it:
            assert next(state["seeds"], None) is None, f"iteration {it}: a noise seed was not drawn"
        image, scaling = pkg["render"], pkg["scaling"]
        Ll1 = l1_loss(image, gt)
        ssim_loss = 1.0 - ssim(image, gt)
        loss = tc.LMBDA_REC * ((1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * ssim_loss) + 0.01 * scaling_reg(scaling)
        bpp = pkg["bit_per_param"]
        if bpp is not None:
            loss = loss + tc.LMBDA * bpp + 5e-4 * mask_reg(pc._mask)
        loss.backward()
        n_before = int(pc._anchor.shape[0])
        assert n_before == int(g["n_before"][idx]), (it, n_before, int(g["n_before"][idx]))
        assert int(vis.sum()) == int(g["n_visible"][idx]), (it, int(vis.sum()), int(g["n_visible"][idx]))
        # the number of Gaussians is a count of opacity > 0 decisions on MLP outputs: a handful may flip at round-off
        dP = abs(int(pkg["radii"].shape[0]) - int(g["P"][idx]))
        assert dP <= max(2, int(g["P"][idx]) // 2000), (it, int(pkg["radii"].shape[0]), int(g["P"][idx]))
        with torch.no_grad():
            if it < opt.update_until and it > opt.start_stat:
                pc.training_statis(pkg["viewspace_points"], pkg["neural_opacity"], pkg["visibility_filter"], pkg["selection_mask"], vis)
                if it not in range(3000, 4000) and it > opt.update_from and it % opt.update_interval == 0:
                    rand_fn = lambda i, like, it=it: T(tc.grow_draw(it, i, like.numel())).reshape(like.shape)
                    pc.adjust_anchor(check_interval=opt.update_interval, success_threshold=opt.success_threshold,
                                     grad_threshold=opt.densify_grad_threshold, min_opacity=opt.min_opacity, rand_fn=rand_fn)
                    ref_anchors = g[f"anchors_after_adjust_{it}"]
                    got = pc._anchor.detach().cpu().numpy()
                    print(f"[trajectory] it {it}: adjust_anchor {n_before} -> {got.shape[0]} anchors (reference {ref_anchors.shape[0]})")
                    assert got.shape == ref_anchors.shape, "the densification round grew / pruned a different number of anchors"
                    assert np.array_equal(got, ref_anchors), "the densification round produced different anchors"
            elif it == opt.update_until:
                del pc.opacity_accum, pc.offset_gradient_accum, pc.offset_denom
            if it < opt.iterations:
                pc.optimizer.step()
                pc.optimizer.zero_grad(set_to_none=True)
        assert int(pc._anchor.shape[0]) == int(g["n_after"][idx])
        rel = abs(float(loss) - float(g["loss"][idx])) / abs(float(g["loss"][idx]))
        worst["loss"] = max(worst["loss"], rel)
        line = f"[trajectory] it {it:6d} loss {float(loss):.6f} (reference {float(g['loss'][idx]):.6f}, rel {rel:.1e})"
        if bpp is not None:
            relb = abs(float(bpp) - float(g["bpp"][idx])) / abs(float(g["bpp"][idx]))
            worst["bpp"] = max(worst["bpp"], relb)
            line += f" bpp {float(bpp):.4f} (reference {float(g['bpp'][idx]):.4f}, rel {relb:.1e})"
        else:
            assert np.isnan(g["bpp"][idx])
        # parameter checksums after the step: |sum| of every parameter tensor within 1e-4 relative (Adam normalises the
        # gradients, so a parameter whose gradient is round-off noise on both sides may move by its learning rate either way —
        # the abs-sums average that out; the loss of the NEXT iteration is the stricter witness)
        sums = tc.checksums(pc)
        for j, name in enumerate(names):
            ref_abs = float(g["sums"][idx, j, 1])
            d = abs(sums[name][1] - ref_abs) / max(ref_abs, 1e-12)
            worst["sum"] = max(worst["sum"], d)
            assert d <= 2e-4, (it, name, sums[name][1], ref_abs)
        print(line)
        assert rel <= 1e-4, (it, float(loss), float(g["loss"][idx]))
        if bpp is not None:
            # the rate is a mean of -log2 of differences of fp32 normal CDFs under a level MLP that Adam moves by its learning rate
            # per step whatever the gradient's size: it drifts a few 1e-4 relative over the context iterations (the loss, which
            # carries it with weight 1e-3, stays inside 1e-4)
            assert relb <= 1e-3, (it, float(bpp), float(g["bpp"][idx]))
    print(f"[trajectory] worst relative deviations over {len(its)} iterations: loss {worst['loss']:.2e}, bit_per_param {worst['bpp']:.2e}, "
          f"parameter |sum| {worst['sum']:.2e}")
    # final per-anchor tensors.  Adam moves an entry by ~its learning rate per step whatever the gradient's SIZE, so an entry whose
    # gradient is round-off noise may end many learning rates apart — on BOTH sides: two runs of the reference's own script
    # (tools/make_trajectory_golden.py, CPU thread order of its reductions) differ by up to 3.5e-3 of the tensor maximum in
    # `offset`, 2.6e-3 in `hyper`, 1.5e-3 in `scaling` (round 6, recorded in the generator's docstring).  With Adam's state young
    # (28 steps, half the anchors born in the densification round) ONE iteration whose gradient is round-off noise — its sign
    # then differs between two implementations — moves the entry a full learning rate the other way, however strong its
    # gradient was in the other iterations.  The fixture therefore carries, per entry, gweakest = the SMALLEST non-zero
    # |gradient| / max |gradient of the tensor| the reference saw since the tensors took their final shape, and the bound
    # depends on it (no blanket allowance; measured maxima on the MI355X in brackets):
    #   gweakest >= 1e-3 (never moved on noise; 36 k offset entries, a few dozen to ~1000 in the other tensors): EVERY such entry
    #                     within TIGHT = 1e-4 of the tensor's maximum [3.2e-5]
    #   gweakest >= 1e-5: every such entry within MID = 2e-3 [4.8e-4];   all entries: within 2e-2 [1.2e-2].
    TIGHT, MID = 1e-4, 2e-3
    bad = []
    for name, attr in tc.PER_ANCHOR.items():
        a, b = getattr(pc, attr).detach().cpu().numpy(), g["final_" + name]
        assert a.shape == b.shape, name
        big = max(float(np.abs(b).max()), 1e-12)
        err = np.abs(a - b) / big
        st, wk = g["gstrength_" + name], g["gweakest_" + name]
        strong, mid = (wk >= 1e-3) | (st == 0), (wk >= 1e-5) | (st == 0)          # (st == 0: never received a gradient)
        e_s = float(err[strong].max()) if strong.any() else 0.0
        e_m = float(err[mid].max()) if mid.any() else 0.0
        print(f"[trajectory] final {name:8s} max err / max: all {float(err.max()):.2e} | weakest gradient >= 1e-5 of the largest ({int(mid.sum())} entries) "
              f"{e_m:.2e} | >= 1e-3 ({int(strong.sum())} entries) {e_s:.2e}; outside 2e-5: {int((err > 2e-5).sum())} of {err.size} "
              f"(strong: {int((err[strong] > 2e-5).sum())})")
        if not (float(err.max()) <= 2e-2 and e_m <= MID and e_s <= TIGHT):
            bad.append((name, float(err.max()), e_m, e_s))
    assert not bad, bad
