"""Densification pieces at bench scale (1 M anchors, K = 10): the HIP statistics pass vs the reference's torch
composition (scene/gaussian_model.py:696-713 restated), and the sorted-key voxel de-duplication vs the
reference's chunked all-pairs compare (:793-802)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import densify
N, K = 1_000_000, 10
g = torch.Generator(device="cuda").manual_seed(0)
vis = torch.rand(N, device="cuda", generator=g) < 0.99
n_vis = int(vis.sum())
opacity = torch.randn(n_vis * K, 1, device="cuda", generator=g) * 0.5 + 0.1
sel = opacity.view(-1) > 0
P = int(sel.sum())
uf = torch.rand(P, device="cuda", generator=g) < 0.8
vsp = torch.zeros(P, 3, device="cuda", requires_grad=True); vsp.grad = torch.randn(P, 3, device="cuda", generator=g) * 1e-3
class M: n_offsets = K
def fresh():
    pc = M()
    pc.opacity_accum = torch.zeros(N, 1, device="cuda"); pc.anchor_demon = torch.zeros(N, 1, device="cuda")
    pc.offset_gradient_accum = torch.zeros(N * K, 1, device="cuda"); pc.offset_denom = torch.zeros(N * K, 1, device="cuda")
    return pc
def ref_statis(pc):
    t = opacity.clone().view(-1).detach(); t[t < 0] = 0; t = t.view([-1, K])
    pc.opacity_accum[vis] += t.sum(dim=1, keepdim=True); pc.anchor_demon[vis] += 1
    m = vis.unsqueeze(dim=1).repeat([1, K]).view(-1)
    comb = torch.zeros_like(pc.offset_gradient_accum, dtype=torch.bool).squeeze(dim=1)
    comb[m] = sel; tm = comb.clone(); comb[tm] = uf
    gn = torch.norm(vsp.grad[uf, :2], dim=-1, keepdim=True)
    pc.offset_gradient_accum[comb] += gn; pc.offset_denom[comb] += 1
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a, b = fresh(), fresh()
ref_statis(a); densify.training_statis(b, vsp, opacity, uf, sel, vis)
same = all(torch.allclose(x, y, rtol=1e-6, atol=1e-9) for x, y in ((a.opacity_accum, b.opacity_accum), (a.anchor_demon, b.anchor_demon),
                                                                    (a.offset_gradient_accum, b.offset_gradient_accum), (a.offset_denom, b.offset_denom)))
print(f"training_statis @ {N} anchors, P={P}: torch composition {t(lambda: ref_statis(a)):.3f} ms, HIP {t(lambda: densify.training_statis(b, vsp, opacity, uf, sel, vis)):.3f} ms, equal={same}")
grid = torch.randint(-200, 200, (N, 3), device="cuda", generator=g, dtype=torch.int32)
cand = torch.unique(torch.randint(-210, 210, (100_000, 3), device="cuda", generator=g, dtype=torch.int32), dim=0)
def slow():
    out = torch.zeros(cand.shape[0], dtype=torch.bool, device="cuda")
    for s in range(0, N, 4096):
        out |= (cand.unsqueeze(1) == grid[s:s + 4096]).all(-1).any(-1)
    return out
f = densify.voxels_already_present(cand, grid)
print(f"voxel de-duplication, {cand.shape[0]} candidates vs {N} anchors: chunked all-pairs {t(slow, 1):.1f} ms, sorted keys {t(lambda: densify.voxels_already_present(cand, grid)):.3f} ms, equal={bool(torch.equal(f, slow()))}")
