"""Shim for the `simple_knn` wheel the reference imports (scene/gaussian_model.py:22): `simple_knn._C.distCUDA2`
on the HIP device (contextgs_amd.knn)."""
