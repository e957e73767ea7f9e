"""Image loss: the numpy oracle against the goldens produced by the reference's own
utils/loss_utils.py (CPU), and the fused HIP kernels against both (GPU)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loss.npz")
TAGS = ["a", "b", "c", "d"]


def _case(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: z[k] for k in z.files if k.startswith(tag + "_")}


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_matches_reference_goldens(tag):
    from oracle.loss_ref import l1_ssim
    c = _case(tag)
    l1, s, g_l1, g_s = l1_ssim(c["img"], c["gt"])
    assert abs(l1 - float(c["l1"])) < 1e-6 and abs(s - float(c["ssim"])) < 2e-6
    grad = 0.8 * g_l1 - 0.2 * g_s                                      # d[(1-l) L1 + l (1 - SSIM)], l = 0.2
    assert np.abs(grad - c["grad"]).max() < 2e-7 + 2e-4 * np.abs(c["grad"]).max()
    # fp64 restatement: same numbers, i.e. the fp32 ones are not a rounding accident
    l1d, sd, _, g_sd = l1_ssim(c["img"], c["gt"], dtype=np.float64)
    assert abs(sd - s) < 2e-6 and np.abs(g_sd - g_s).max() < 1e-4 * np.abs(g_s).max() + 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_hip_matches_goldens_and_oracle(tag):
    import torch
    from contextgs_amd.loss_utils import l1_loss, l1_ssim, ssim
    c = _case(tag)
    img = torch.tensor(c["img"], device="cuda", requires_grad=True)
    gt = torch.tensor(c["gt"], device="cuda")
    l1, s = l1_ssim(img, gt)
    assert abs(float(l1) - float(c["l1"])) < 1e-6 and abs(float(s) - float(c["ssim"])) < 2e-6
    loss = 0.8 * l1 + 0.2 * (1.0 - s)
    (g,) = torch.autograd.grad(loss, [img])
    assert np.abs(g.cpu().numpy() - c["grad"]).max() < 2e-7 + 2e-4 * np.abs(c["grad"]).max()
    # the reference's separate entry points
    assert abs(float(ssim(img, gt)) - float(c["ssim"])) < 2e-6 and abs(float(l1_loss(img, gt)) - float(c["l1"])) < 1e-6


@pytest.mark.gpu
def test_hip_full_hd_properties_and_torch_reference():
    """1080p: SSIM(x, x) == 1 with zero gradient; against a plain torch fp32 conv2d restatement of the same op."""
    import torch
    import torch.nn.functional as F
    from contextgs_amd.loss_utils import l1_ssim
    from oracle.loss_ref import window
    g = torch.Generator(device="cuda").manual_seed(3)
    gt = torch.rand(3, 1080, 1920, device="cuda", generator=g)
    img = (gt + 0.1 * torch.randn(3, 1080, 1920, device="cuda", generator=g)).clamp(0, 1).requires_grad_()
    l1_same, s_same = l1_ssim(gt.clone().requires_grad_(), gt)
    assert float(l1_same) == 0.0 and abs(float(s_same) - 1.0) < 1e-6
    w1 = torch.tensor(window(), device="cuda")
    w2 = (w1[:, None] * w1[None, :]).expand(3, 1, 11, 11).contiguous()
    conv = lambda t: F.conv2d(t[None], w2, padding=5, groups=3)[0]
    mu1, mu2 = conv(img), conv(gt)
    s1, s2, s12 = conv(img * img) - mu1 * mu1, conv(gt * gt) - mu2 * mu2, conv(img * gt) - mu1 * mu2
    ref_s = (((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean()
    ref_l1 = (img - gt).abs().mean()
    l1, s = l1_ssim(img, gt)
    assert abs(float(l1) - float(ref_l1)) < 1e-6 and abs(float(s) - float(ref_s)) < 5e-6
    (ga,) = torch.autograd.grad(0.8 * l1 + 0.2 * (1 - s), [img])
    (gb,) = torch.autograd.grad(0.8 * ref_l1 + 0.2 * (1 - ref_s), [img])
    assert float((ga - gb).abs().max()) < 2e-4 * float(gb.abs().max())


# ---- the two regularisers train.py:203,209 adds to the image terms ----------------------------------------------------

def _reg_inputs(P, seed):
    r = np.random.default_rng(seed)
    scaling = np.exp(r.normal(-3.0, 1.0, size=(P, 3))).astype(np.float32)
    if P > 2:
        scaling[1, 1] = 0.0           # torch's prod backward has a separate zero-safe path: d/d(that entry) = product of the others
    mask = r.normal(0.5, 2.0, size=(P, 10, 1)).astype(np.float32)
    return scaling, mask


@pytest.mark.parametrize("P", [1, 3, 4, 5, 1001])
def test_reg_oracle_matches_the_reference_expressions(P):
    """oracle/loss_ref.py against torch's CPU autograd of the literal expressions of train.py:203,209 (fp64)."""
    import torch
    from oracle.loss_ref import scaling_reg, mask_reg
    scaling, mask = _reg_inputs(P, 7 + P)
    s = torch.tensor(scaling, dtype=torch.float64, requires_grad=True)
    m = torch.tensor(mask, dtype=torch.float64, requires_grad=True)
    a, b = s.prod(dim=1).mean(), torch.mean(torch.sigmoid(m))
    ga, gb = torch.autograd.grad(a, s)[0], torch.autograd.grad(b, m)[0]
    va, da = scaling_reg(scaling)
    vb, db = mask_reg(mask)
    assert abs(va - float(a)) <= 1e-14 * abs(float(a)) + 1e-300 and np.allclose(da, ga.numpy(), rtol=1e-13, atol=0)
    assert abs(vb - float(b)) <= 1e-14 and np.allclose(db, gb.numpy(), rtol=1e-12, atol=1e-300)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 3, 4, 5, 1001, 1_000_003])
def test_hip_regularisers_match_oracle(P):
    import torch
    from contextgs_amd.loss_utils import scaling_reg, mask_reg
    from oracle import loss_ref
    scaling, mask = _reg_inputs(P, 7 + P)
    for x_np, op, ref in ((scaling, scaling_reg, loss_ref.scaling_reg), (mask, mask_reg, loss_ref.mask_reg)):
        v_ref, d_ref = ref(x_np)
        for misaligned in (False, True):
            if misaligned:        # a view that does not start on 16 bytes goes through an aligned copy
                buf = torch.zeros(x_np.size + 1, device="cuda")
                buf[1:] = torch.tensor(x_np.reshape(-1), device="cuda")
                x = buf[1:].view(x_np.shape).requires_grad_()
            else:
                x = torch.tensor(x_np, device="cuda", requires_grad=True)
            v = op(x)
            (d,) = torch.autograd.grad(3.0 * v, [x])
            assert abs(float(v) - v_ref) <= 2e-6 * abs(v_ref) + 1e-12, (float(v), v_ref)       # fp32 sums of P terms
            d = d.cpu().numpy().astype(np.float64) / 3.0
            # s (1 - s) in fp32 loses relative accuracy where s -> 1 (the reference's sigmoid backward forms it the same way):
            # absolute tolerance of a few fp32 roundings of the largest entry for the mask term
            atol = (1e-12 if op is scaling_reg else 5e-7) * np.abs(d_ref).max()
            assert np.allclose(d, d_ref, rtol=3e-6, atol=atol)
        # the reference's own expression in fp32 on the device
        t = torch.tensor(x_np, device="cuda", requires_grad=True)
        e = t.prod(dim=1).mean() if op is scaling_reg else torch.mean(torch.sigmoid(t))
        (g,) = torch.autograd.grad(e, [t])
        assert abs(float(op(t)) - float(e)) <= 3e-6 * abs(float(e)) + 1e-12
        assert np.allclose(g.cpu().numpy(), d_ref, rtol=2e-5, atol=(1e-10 if op is scaling_reg else 5e-7) * np.abs(d_ref).max())


@pytest.mark.gpu
def test_hip_regularisers_reject_bad_arguments():
    import torch
    from contextgs_amd import _lib
    from contextgs_amd.loss_utils import scaling_reg, mask_reg
    with pytest.raises(ValueError):
        scaling_reg(torch.ones(4, 2, device="cuda"))
    with pytest.raises(Exception):
        scaling_reg(torch.ones(4, 3))                      # host tensor: no CPU path
    L = _lib.lib()
    x = torch.ones(8, 3, device="cuda")
    out = torch.empty(int(L.cgs_reg_partials(8)), device="cuda")
    assert L.cgs_scaling_reg_fwd(_lib.ptr(x), 0, _lib.ptr(out), _lib.current_stream()) != 0
    assert L.cgs_scaling_reg_fwd(None, 8, _lib.ptr(out), _lib.current_stream()) != 0
    assert L.cgs_sigmoid_mean_fwd(x.data_ptr() + 4, 8, _lib.ptr(out), _lib.current_stream()) != 0      # misaligned
    assert torch.isnan(scaling_reg(torch.ones(0, 3, device="cuda")))     # empty view: nan, like the reference's expression
    assert torch.isnan(mask_reg(torch.ones(0, 10, 1, device="cuda")))


@pytest.mark.gpu
@pytest.mark.parametrize("n,with_rate", [(1, True), (1023, False), (3 * 180 * 320, True), (3 * 1080 * 1920, True)])
def test_weighted_image_sum_matches_torch_fp64(n, with_rate):
    """sum(image * w) + lam * rate (cgs_weighted_sum_*, the bench's linear objective) against the fp64 expression: value within the
    fp32 products' rounding, gradients exact (g * w, g * lam), two calls bit-identical (deterministic reduction order)."""
    import torch
    from contextgs_amd.loss_utils import weighted_image_sum
    gen = torch.Generator(device="cuda").manual_seed(n)
    img = torch.rand(n, device="cuda", generator=gen).requires_grad_(True)
    w = torch.randn(n, device="cuda", generator=gen)
    rate = torch.tensor(3.25, device="cuda", requires_grad=True) if with_rate else None
    out = weighted_image_sum(img, w, rate, 0.001)
    ref = (img.detach().double() * w.double()).sum() + (0.001 * 3.25 if with_rate else 0.0)
    assert abs(float(out) - float(ref)) <= 1e-6 * float((img.detach().double() * w.double()).abs().sum()) + 1e-7
    assert torch.equal(weighted_image_sum(img, w, rate, 0.001), out)
    (2.0 * out).backward()
    assert torch.equal(img.grad, 2.0 * w)
    if with_rate:
        assert abs(float(rate.grad) - 0.002) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_training_image_loss_one_node_matches_goldens(tag):
    """training_image_loss (train.py:199-204 as one node: forward kernel + finish + one backward launch) against the reference's
    goldens and against the two-output node it replaces; the extra outputs (L1, SSIM) stay differentiable."""
    import torch
    from contextgs_amd.loss_utils import l1_ssim, training_image_loss
    c = _case(tag)
    img = torch.tensor(c["img"], device="cuda", requires_grad=True)
    gt = torch.tensor(c["gt"], device="cuda")
    loss, l1, s = training_image_loss(img, gt, 0.2)
    assert abs(float(l1) - float(c["l1"])) < 1e-6 and abs(float(s) - float(c["ssim"])) < 2e-6
    assert abs(float(loss) - (0.8 * float(c["l1"]) + 0.2 * (1.0 - float(c["ssim"])))) < 2e-6
    (g,) = torch.autograd.grad(loss, [img], retain_graph=True)
    assert np.abs(g.cpu().numpy() - c["grad"]).max() < 2e-7 + 2e-4 * np.abs(c["grad"]).max()
    l1b, sb = l1_ssim(img, gt)
    (gb,) = torch.autograd.grad(0.8 * l1b + 0.2 * (1.0 - sb), [img])
    assert float((g - gb).abs().max()) <= 1e-6 * float(gb.abs().max()) + 1e-12
    # an objective that also reads L1 and SSIM directly
    (g2,) = torch.autograd.grad(2.0 * loss + 0.5 * l1 - 0.25 * s, [img])
    l1c, sc = l1_ssim(img, gt)
    (g2b,) = torch.autograd.grad(2.0 * (0.8 * l1c + 0.2 * (1.0 - sc)) + 0.5 * l1c - 0.25 * sc, [img])
    assert float((g2 - g2b).abs().max()) <= 2e-6 * float(g2b.abs().max()) + 1e-12
