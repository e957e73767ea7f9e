"""Shim with the import name the reference uses (gaussian_renderer/__init__.py:20)."""
from contextgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401
