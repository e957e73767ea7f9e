"""Shim for the `torchac` wheel (utils/encodings.py:6): same two functions, backed by
libcgs_hip.so's host arithmetic coder."""
from contextgs_amd.codec import decode_float_cdf, encode_float_cdf  # noqa: F401
