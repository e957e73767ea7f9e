"""Fused factorised-prior likelihood kernel (csrc/eb.hip) vs the torch composition of the same
density (EntropyBottleneck._likelihood, itself the maths of utils/entropy_models.py:103-138)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 1000, 70001])
@pytest.mark.parametrize("training", [False, True])
def test_fused_likelihood_matches_torch_path(n, training):
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck, _LowerBound
    torch.manual_seed(n)
    eb = EntropyBottleneck(12).cuda()
    with torch.no_grad():
        for p in list(eb.matrices) + list(eb.factors):
            p.add_(0.3 * torch.randn_like(p))
    x = (torch.randn(n, 12, device="cuda") * 4).requires_grad_(True)
    w = torch.randn(n, 12, device="cuda")
    torch.manual_seed(7)
    xh, lik = eb(x, training=training)                       # fused path (CUDA tensor, filters 3,3,3,3)
    (-(torch.log2(lik)) * w).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in eb.parameters() if p.grad is not None]
    names = [k for k, p in eb.named_parameters() if p.grad is not None]
    x.grad = None
    eb.zero_grad()
    # reference: the torch composition on the SAME quantised values
    v = xh.detach().t().reshape(12, 1, -1)
    xr = x if training else None
    vv = xh.detach().clone().requires_grad_(True)
    lik_ref = _LowerBound.apply(eb._likelihood(vv.t().reshape(12, 1, -1)), eb.likelihood_bound).reshape(12, -1).t()
    (-(torch.log2(lik_ref)) * w).sum().backward()
    assert torch.allclose(lik, lik_ref, rtol=2e-4, atol=1e-9)
    ref = [vv.grad] + [p.grad for p in eb.parameters() if p.grad is not None]
    if training:     # noise path is differentiable w.r.t. x; the rounded path is not (gradient of round is dropped)
        assert torch.allclose(got[0], ref[0], rtol=2e-3, atol=2e-5 * float(ref[0].abs().max()))
    for a, b, k in zip(got[1:], ref[1:], names):
        assert (a - b).abs().max() <= 3e-4 * max(1e-6, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))
