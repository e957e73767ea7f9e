"""_lib.current_stream() is torch's current stream — also inside a torch.cuda.stream(...) block (the library enqueues there)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_current_stream_follows_torch():
    from contextgs_amd import _lib
    assert _lib.current_stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert _lib.current_stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
        # and work issued through the library lands on it: a scan on the side stream, ordered by that stream alone
        x = torch.ones(1 << 16, dtype=torch.int32, device="cuda")
        out = torch.empty_like(x)
        ws = torch.empty(int(_lib.lib().cgs_scan_scratch_bytes(x.numel())), dtype=torch.uint8, device="cuda")
        _lib.check(_lib.lib().cgs_scan_exclusive_u32(_lib.ptr(x), _lib.ptr(out), x.numel(), _lib.ptr(ws), ws.numel(),
                                                      _lib.current_stream()), "cgs_scan_exclusive_u32")
        side.synchronize()
        assert int(out[-1]) == x.numel() - 1
    assert _lib.current_stream() == torch.cuda.current_stream().cuda_stream


def test_training_step_on_a_side_stream_equals_the_default_stream():
    """Everything a step enqueues (kernels, the count copies and their events, torch's allocations) follows torch's current
    stream: the same step inside torch.cuda.stream(side) gives the same image and, up to the blend backward's atomic order,
    the same gradients."""
    import itertools
    from contextgs_amd import ctx_ops
    from contextgs_amd.renderer import prefilter_voxel, render
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    pipe, bg = SynthPipe(), torch.zeros(3, device="cuda")
    cam = orbit_cameras(2, 160, 96)[0].to_torch("cuda")
    pc = make_scene(5000, seed=1)
    pc.train()

    def step():
        torch.manual_seed(0)
        ctx_ops._seed_counter = itertools.count(1)
        pc.zero_grad()
        vis = prefilter_voxel(cam, pc, pipe, bg)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, step=20000)
        (pkg["render"].sum() + pkg["bit_per_param"]).backward()
        return pkg["render"].detach().clone(), {k: p.grad.clone() for k, p in pc.named_parameters() if p.grad is not None}

    img0, g0 = step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        img1, g1 = step()
    side.synchronize()
    assert torch.equal(img0, img1) and set(g0) == set(g1)
    for k in g0:
        assert float((g0[k] - g1[k]).abs().max()) <= 1e-5 * float(g0[k].abs().max()) + 1e-12, k


def test_library_loaded_before_torch_shares_torch_s_hip_runtime():
    """A process that touches the library before it imports torch (the driver's build() then smoke()) must end up with ONE
    HIP runtime: _lib.lib() imports torch first.  Run in a fresh interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import __graft_entry__ as g\n"
            "g.build()\n"
            "assert 'torch' in sys.modules\n"
            "g.smoke()\n"
            "print('both ok')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "both ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
