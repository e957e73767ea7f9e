"""The drop-in layer: shim packages carry the reference's import names, and — in the authoring
container only, where /root/reference exists — the reference's own GaussianModel / render code
imports and constructs on top of them."""
import importlib
import os
import sys

import pytest


def test_shims_expose_reference_import_names():
    import contextgs_amd.dropin as dropin
    dropin.install(patch_reference_python=False)
    dgr = importlib.import_module("diff_gaussian_rasterization")
    assert hasattr(dgr, "GaussianRasterizationSettings") and hasattr(dgr, "GaussianRasterizer")
    fields = dgr.GaussianRasterizationSettings._fields
    assert fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                      "projmatrix", "sh_degree", "campos", "prefiltered", "debug")      # gaussian_renderer/__init__.py:179-192
    ta = importlib.import_module("torchac")
    assert callable(ta.encode_float_cdf) and callable(ta.decode_float_cdf)
    em = importlib.import_module("compressai.entropy_models")
    assert callable(importlib.import_module("torch_scatter").scatter_max)              # scene/gaussian_model.py:24
    eb = em.EntropyBottleneck(channels=12)
    for m in ("forward", "quantize", "compress", "decompress", "update", "_get_medians"):
        assert hasattr(eb, m)


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="reference checkout only exists in the authoring container")
def test_reference_python_imports_on_top_of_the_shims():
    import types
    import contextgs_amd.dropin as dropin
    for name in ("plyfile", "simple_knn", "simple_knn._C", "colorama"):                         # not on the hot path
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["simple_knn._C"].distCUDA2 = None
    sys.modules["colorama"].Fore = sys.modules["colorama"].Style = types.SimpleNamespace(YELLOW="", RESET_ALL="")
    sys.modules["colorama"].init = lambda *a, **k: None
    sys.path.insert(0, "/root/reference")
    try:
        dropin.install(patch_reference_python=False)
        gm = importlib.import_module("scene.gaussian_model")
        patched = dropin.install()
        assert "scene.gaussian_model.multi_scale_generating" in patched
        assert "scene.gaussian_model.GaussianModel.training_statis" in patched
        assert "scene.gaussian_model.GaussianModel.anchor_growing" in patched
        assert "utils.loss_utils.ssim" in patched
        from contextgs_amd import context_model
        assert gm.multi_scale_generating is context_model.multi_scale_generating
        assert gm.EntropyBottleneck.__module__ == "contextgs_amd.entropy_bottleneck"
    finally:
        sys.path.remove("/root/reference")
        for k in [k for k in sys.modules if k.split(".")[0] in ("scene", "utils", "gaussian_renderer", "arguments")]:
            del sys.modules[k]


def test_install_repoints_a_loaded_script_without_importing_train():
    """train.py binds render / l1_loss / ssim by name (train.py:36-37): install() re-points modules that already hold
    them (`train`, `__main__`) and never imports `train` itself (that would run the script's top level twice)."""
    import types
    import contextgs_amd.dropin as dropin
    from contextgs_amd import loss_utils, renderer
    sys.modules.pop("train", None)
    dropin.install()
    assert "train" not in sys.modules                      # not imported as a side effect
    fake = types.ModuleType("train")
    def ref_fn(module):                                    # stands for a function the reference's module defined
        f = lambda *a, **k: None
        f.__module__ = module
        return f
    fake.render, fake.prefilter_voxel = ref_fn("gaussian_renderer"), ref_fn("gaussian_renderer")
    fake.l1_loss, fake.ssim = ref_fn("utils.loss_utils"), ref_fn("utils.loss_utils")
    fake.unrelated = 1
    own_render = ref_fn("my_viewer")                       # an unrelated script's own `render` must survive install()
    other = types.ModuleType("__main__")
    other.render = own_render
    real_main = sys.modules["__main__"]
    sys.modules["__main__"] = other
    sys.modules["train"] = fake
    try:
        patched = dropin.install()
        assert fake.render is renderer.render and fake.prefilter_voxel is renderer.prefilter_voxel
        assert fake.l1_loss is loss_utils.l1_loss and fake.ssim is loss_utils.ssim and fake.unrelated == 1
        assert "train.render" in patched and "train.ssim" in patched
        assert other.render is own_render and "__main__.render" not in patched
    finally:
        sys.modules["__main__"] = real_main
        del sys.modules["train"]
        for k in [k for k in sys.modules if k.split(".")[0] in ("scene", "utils", "gaussian_renderer", "arguments")]:
            del sys.modules[k]


def test_latent_codec_checkpoint_must_carry_the_density_parameters():
    import torch
    from contextgs_amd.codec_driver import _load_latent_codec
    from contextgs_amd.entropy_bottleneck import EntropyBottleneck
    src = EntropyBottleneck(12)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.1)
    sd = src.state_dict()
    dst = EntropyBottleneck(12)
    _load_latent_codec(dst, sd)
    assert all(torch.equal(a, b) for a, b in zip(src.parameters(), dst.parameters()))
    # names of older compressai releases are remapped
    legacy = {}
    for k, v in sd.items():
        head, _, idx = k.partition(".")
        legacy["_" + {"matrices": "matrix", "biases": "bias", "factors": "factor"}[head] + idx if head in ("matrices", "biases", "factors") else k] = v
    dst = EntropyBottleneck(12)
    _load_latent_codec(dst, legacy)
    assert all(torch.equal(a, b) for a, b in zip(src.parameters(), dst.parameters()))
    # a checkpoint without them is an error, not a silently random prior
    with pytest.raises(RuntimeError, match="density parameters"):
        _load_latent_codec(EntropyBottleneck(12), {k: v for k, v in sd.items() if not k.startswith("factors")})
