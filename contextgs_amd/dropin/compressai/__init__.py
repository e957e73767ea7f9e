"""Shim for the part of `compressai` the reference imports (scene/gaussian_model.py:19,26-27)."""
from . import entropy_models, latent_codecs  # noqa: F401
