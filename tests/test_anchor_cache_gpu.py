"""model.get_anchor quantises once per version of the anchors (round 6): the second access of a step (render(), with a graph, after
prefilter_voxel's without one) wraps the first one's values — same values, same straight-through gradient, no stale values."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _launches(fn):
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    return out, [e.key for e in prof.key_averages() if "quantize_anchor" in e.key]


def test_get_anchor_is_quantised_once_per_version_and_never_stale(monkeypatch):
    from contextgs_amd import model as M
    from contextgs_amd.synth import make_scene
    pc = make_scene(5000, seed=3)
    pc.train()
    with torch.no_grad():
        first, k1 = _launches(lambda: pc.get_anchor)
    assert len(k1) == 1
    again, k2 = _launches(lambda: pc.get_anchor)                  # with a graph: the node around the cached values
    assert k2 == [] and torch.equal(again, first) and again.requires_grad and again.grad_fn is not None
    w = torch.randn_like(again)
    (again * w).sum().backward()
    assert torch.equal(pc._anchor.grad, w)                         # straight-through
    monkeypatch.setattr(M, "ANCHOR_Q_CACHE", False)
    plain, k3 = _launches(lambda: pc.get_anchor)
    assert len(k3) == 1 and torch.equal(plain, first)
    monkeypatch.setattr(M, "ANCHOR_Q_CACHE", True)
    # an optimizer-style in-place update (version bump), a write through .data with a new storage, new bounds: all recompute
    with torch.no_grad():
        pc._anchor.add_(0.37)
    moved, k4 = _launches(lambda: pc.get_anchor)
    assert len(k4) == 1 and not torch.equal(moved, first)
    monkeypatch.setattr(M, "ANCHOR_Q_CACHE", False)
    assert torch.equal(moved.detach(), pc.get_anchor.detach())
    monkeypatch.setattr(M, "ANCHOR_Q_CACHE", True)
    _ = pc.get_anchor
    pc._anchor.data = pc._anchor.data.clone() * 1.01
    replaced, k5 = _launches(lambda: pc.get_anchor)
    assert len(k5) == 1
    pc.update_anchor_bound()
    rebound, k6 = _launches(lambda: pc.get_anchor)
    assert len(k6) == 1
    monkeypatch.setattr(M, "ANCHOR_Q_CACHE", False)
    assert torch.equal(rebound.detach(), pc.get_anchor.detach())


def test_training_step_is_the_same_with_and_without_the_anchor_cache(monkeypatch):
    """Same seeds, same scene: a training step (prefilter_voxel + render + backward) whose second get_anchor wraps the cached values
    and one that quantises twice — the same image and rate bit for bit, gradients up to the blend backward's atomic order."""
    import itertools
    from contextgs_amd import ctx_ops, model as M
    from test_training_gpu import _ctx_step, _setup
    outs = []
    for on in (True, False):
        monkeypatch.setattr(M, "ANCHOR_Q_CACHE", on)
        pc, cams, pipe, bg = _setup(N=12000, W=256, H=144, seed=9)
        _ctx_step(pc, cams[0], pipe, bg)                     # builds the plan
        ctx_ops._seed_counter = itertools.count(1000)        # the same noise stream ids for both runs
        pkg, loss = _ctx_step(pc, cams[1], pipe, bg)
        outs.append((pkg["render"].detach().clone(), float(pkg["bit_per_param"]), pc._anchor.grad.clone(), pc._anchor_feat.grad.clone()))
    a, b = outs
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    close = lambda x, y: float((x - y).abs().max()) <= 2e-5 * float(x.abs().max()) + 1e-12
    assert close(a[2], b[2]) and close(a[3], b[3])
