"""Drop-in glue: make the reference checkout (wyf0912/ContextGS) run on contextgs_amd.

Two levels:
  * put this directory on PYTHONPATH -> the shim packages `diff_gaussian_rasterization`,
    `torchac`, `compressai` replace the reference's absent native wheels;
  * call `install()` -> additionally rebind the reference's own Python hot-path functions
    to the accelerated ones.
"""
from __future__ import annotations

import importlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install(patch_reference_python: bool = True) -> list[str]:
    """Register the shims and (optionally) rebind the reference's hot-path functions.
    Returns the list of names that were patched."""
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    for name in ("diff_gaussian_rasterization", "torchac", "compressai", "compressai.entropy_models",
                 "compressai.latent_codecs", "torch_scatter", "simple_knn", "simple_knn._C"):
        importlib.import_module(name)
    patched = []
    if not patch_reference_python:
        return patched
    from .. import codec_driver, context_model, densify, encodings, entropy_models, loss_utils, multi_level, renderer

    def rebind(modname, src, names):
        try:
            mod = importlib.import_module(modname)
        except Exception:
            return                      # reference module not importable in this process: nothing to patch
        for n in names:
            if hasattr(src, n):
                setattr(mod, n, getattr(src, n))
                patched.append(f"{modname}.{n}")

    rebind("utils.entropy_models", entropy_models,
           ["Entropy_gaussian", "Entropy_gaussian_clamp", "Entropy_bernoulli", "Entropy_factorized", "Low_bound", "UniverseQuant"])
    rebind("utils.encodings", encodings,
           ["STE_binary", "STE_multistep", "Quantize_anchor", "encoder", "decoder", "encoder_gaussian", "decoder_gaussian",
            "get_binary_vxl_size"])
    rebind("utils.multi_level", multi_level, ["torch_unique_with_indices"])
    ctx_names = ["multi_scale_generating", "extract_context_feat", "find_divide_scale", "divide_levels", "mapping_to_orign",
                 "index_of_level_L_in_orign"]
    rebind("scene.gaussian_model", context_model, ctx_names)
    rebind("scene.gaussian_model", encodings, ["STE_binary", "STE_multistep", "Quantize_anchor", "encoder", "decoder",
                                               "encoder_gaussian", "decoder_gaussian", "get_binary_vxl_size"])
    rebind("scene.gaussian_model", entropy_models, ["Entropy_gaussian", "Entropy_bernoulli", "Entropy_factorized"])
    rebind("scene.gaussian_model", multi_level, ["torch_unique_with_indices"])
    try:
        gm = importlib.import_module("scene.gaussian_model")
        gm.GaussianModel.conduct_encoding = lambda self, p: codec_driver.conduct_encoding(self, p)
        gm.GaussianModel.conduct_decoding = lambda self, p: codec_driver.conduct_decoding(self, p)
        gm.GaussianModel.estimate_final_bits = lambda self: codec_driver.estimate_final_bits(self)
        patched += ["scene.gaussian_model.GaussianModel.conduct_encoding", "scene.gaussian_model.GaussianModel.conduct_decoding",
                    "scene.gaussian_model.GaussianModel.estimate_final_bits"]
        # densification: statistics kernel + sort-based voxel de-duplication
        gm.GaussianModel.training_statis = lambda self, *a, **k: densify.training_statis(self, *a, **k)
        gm.GaussianModel.anchor_growing = lambda self, *a, **k: densify.anchor_growing(self, *a, **k)
        patched += ["scene.gaussian_model.GaussianModel.training_statis", "scene.gaussian_model.GaussianModel.anchor_growing"]
        # round 3: the optimizer surgery around them (one-launch row compaction of parameters + Adam moments + statistics)
        gm.GaussianModel.adjust_anchor = lambda self, *a, **k: densify.adjust_anchor(self, *a, **k)
        gm.GaussianModel.prune_anchor = lambda self, mask: densify.prune_anchor(self, mask)
        gm.GaussianModel.cat_tensors_to_optimizer = lambda self, d: densify.cat_tensors_to_optimizer(self, d)
        patched += ["scene.gaussian_model.GaussianModel.adjust_anchor", "scene.gaussian_model.GaussianModel.prune_anchor",
                    "scene.gaussian_model.GaussianModel.cat_tensors_to_optimizer"]
    except Exception:
        pass
    rebind("gaussian_renderer", renderer, ["render", "prefilter_voxel", "generate_neural_gaussians"])
    rebind("gaussian_renderer", context_model, ["multi_scale_generating"])
    # image loss: train.py does `from utils.loss_utils import l1_loss, ssim` (train.py:37)
    rebind("utils.loss_utils", loss_utils, ["l1_loss", "ssim"])

    # train.py binds `render`, `prefilter_voxel`, `l1_loss`, `ssim` by name at import time (train.py:36-37).  It is never
    # imported from here — when it is the running script that would execute its top level a second time as a separate
    # module and patch only the copy.  Modules that ALREADY hold those names are re-pointed instead: `train` if some
    # caller imported it, and `__main__` when it is the script.  Called before the script's own imports, install()
    # needs neither: the names then resolve to the patched modules above.

    def rebind_loaded(modname, src, names):
        mod = sys.modules.get(modname)
        if mod is None:
            return
        for n in names:
            cur = getattr(mod, n, None)
            # only names that ARE the reference's functions (defined in gaussian_renderer / utils.loss_utils): an unrelated
            # script that happens to have its own `render` or `ssim` is left alone
            owner = getattr(cur, "__module__", "") or ""
            if hasattr(src, n) and cur is not None and owner.split(".")[0] in ("gaussian_renderer", "utils"):
                setattr(mod, n, getattr(src, n))
                patched.append(f"{modname}.{n}")

    for script in ("train", "__main__"):
        rebind_loaded(script, renderer, ["render", "prefilter_voxel"])
        rebind_loaded(script, loss_utils, ["l1_loss", "ssim"])
    return patched
