"""ctypes binding of libcgs_hip.so (include/cgs.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised.  torch is used only for device memory and the
current stream; every entry point receives raw device pointers.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGS_LIB_PATH: another build of the same library (tools/variant_lib.sh: one translation unit recompiled with timing-experiment
# defines, kept under tools/variants/); the default is the in-tree product library.  A CGS_LIB_PATH library is only accepted
# when it was built from the sources next to this file (cgs_build_info() carries their digest) — a stale A/B library must be
# asked for explicitly with CGS_LIB_ALLOW_STALE=1 (tools/ab_lib.sh does) — and its use is announced on stderr.
LIB_PATH = os.environ.get("CGS_LIB_PATH") or os.path.join(_HERE, "libcgs_hip.so")

_lib = None
_lock = threading.Lock()

c_void_p, c_int, c_int32, c_int64, c_size_t, c_float = C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_size_t, C.c_float


class RasterCfg(C.Structure):
    """struct cgs_raster_cfg — mirrors GaussianRasterizationSettings
    (reference: gaussian_renderer/__init__.py:179-192)."""
    _fields_ = [
        ("image_height", c_int32),
        ("image_width", c_int32),
        ("tanfovx", c_float),
        ("tanfovy", c_float),
        ("scale_modifier", c_float),
        ("prefiltered", c_int32),
        ("debug", c_int32),
        ("viewmatrix", c_void_p),
        ("projmatrix", c_void_p),
        ("campos", c_void_p),
        ("bg", c_void_p),
    ]


# name -> (restype, argtypes); the authoritative list of exported symbols.
# tests/test_abi.py checks it against include/cgs.h.
SIGNATURES = {
    "cgs_version": (c_int, []),
    "cgs_last_error": (C.c_char_p, []),
    "cgs_filter": (c_int, [C.POINTER(RasterCfg), c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_filter_voxel": (c_int, [C.POINTER(RasterCfg), c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_raster_geom_bytes": (c_size_t, [c_int64]),
    "cgs_raster_bin_bytes": (c_size_t, [c_int64, c_int64]),
    "cgs_raster_img_bytes": (c_size_t, [c_int32, c_int32]),
    "cgs_raster_bwd_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_raster_preprocess": (c_int, [C.POINTER(RasterCfg), c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p, C.POINTER(c_int64), c_void_p]),
    "cgs_raster_preprocess_launch": (c_int, [C.POINTER(RasterCfg), c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, C.POINTER(C.c_uint64)]),
    "cgs_raster_preprocess_wait": (c_int, [C.c_uint64, C.POINTER(c_int64)]),
    "cgs_raster_preprocess_wait2": (c_int, [C.c_uint64, C.POINTER(c_int64), C.POINTER(c_int)]),
    "cgs_debug_set_depth_keys_full": (c_int, [c_int]),
    "cgs_sort_depth_keys": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                                    C.c_uint32, c_void_p]),
    "cgs_raster_render_spec": (c_int, [C.POINTER(RasterCfg), c_int64, c_int64, c_void_p, c_size_t, c_void_p, c_size_t,
                                       c_void_p, c_size_t, c_void_p, c_void_p]),
    "cgs_raster_render": (c_int, [C.POINTER(RasterCfg), c_int64, c_int64, c_void_p, c_size_t, c_void_p, c_size_t,
                                  c_void_p, c_size_t, c_void_p, c_void_p]),
    "cgs_raster_backward": (c_int, [C.POINTER(RasterCfg), c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_size_t, c_void_p]),
    "cgs_raster_preprocess_expand_launch": (c_int, [C.POINTER(RasterCfg), c_int64, c_int] + [c_void_p] * 9 + [c_int64, c_void_p,
                                                    c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, C.POINTER(C.c_uint64)]),
    "cgs_raster_stats": (c_int, [C.POINTER(RasterCfg), c_void_p, c_size_t, c_void_p, c_void_p]),
    "cgs_debug_set_bin_mode": (c_int, [c_int]),
    "cgs_debug_bin_compare": (c_int, [C.POINTER(RasterCfg), c_int64, c_int64, c_int64, c_void_p, c_size_t, c_void_p, c_size_t,
                                      c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "cgs_build_info": (C.c_char_p, []),
    "cgs_scan_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_scan_exclusive_u32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_sort_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_sort_pairs_u32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                   c_void_p, c_size_t, c_void_p]),
    "cgs_quantize_anchor": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_ste_multistep": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_entropy_gaussian_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int,
                                         c_void_p, c_void_p]),
    "cgs_entropy_gaussian_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_mlp2_forward": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "cgs_mlp2_backward": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_mlp2_backward_rows": (c_int, [c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_mlp_wgrad_scratch_bytes": (c_size_t, []),
    "cgs_mlp2_wgrad": (c_int, [c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_anchor_mlp3_wgrad": (c_int, [c_void_p, c_int64] + [c_void_p] * 9 + [c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_anchor_mlp3_forward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int64, c_void_p]),
    "cgs_anchor_mlp3_backward": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_anchor_mlp3_forward_rows": (c_int, [c_void_p] * 13 + [c_int64, c_void_p]),
    "cgs_anchor_mlp3_backward_rows": (c_int, [c_void_p] * 22 + [c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_anchor_mlp3_forward_rows_t": (c_int, [c_void_p] * 13 + [c_int64, c_int, c_void_p]),
    "cgs_anchor_mlp3_backward_rows_t": (c_int, [c_void_p] * 22 + [c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "cgs_rowcat_fwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_rowcat_bwd": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_anchor_mlp3_layout": (c_int, [c_void_p]),
    "cgs_gather_rows_segmented": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_scatter_rows_sorted": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_add_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_weighted_sum_scratch_bytes": (c_size_t, []),
    "cgs_weighted_sum_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_size_t, c_void_p, c_void_p]),
    "cgs_weighted_sum_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p]),
    "cgs_gather_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_mark_rows": (c_int, [c_void_p, c_int64, c_int64, C.c_uint32, c_void_p, c_void_p]),
    "cgs_zero_unmarked_rows": (c_int, [c_void_p, C.c_uint32, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_rowcat_fwd_masked": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_rowcat_bwd_masked": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                      c_void_p]),
    "cgs_ctx_gather_bwd": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int, c_int, c_int, c_void_p]),
    "cgs_ctx_gather_bwd_acc": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_int, c_void_p]),
    "cgs_noise_quant_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                    C.c_uint64, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "cgs_hyper_noise_gather": (c_int, [c_void_p, c_void_p, c_int64, c_int, C.c_uint64, c_void_p, c_void_p]),
    "cgs_eb_bits_scratch_bytes": (c_size_t, []),
    "cgs_eb_bits_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "cgs_eb_bits_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_rate_finish_fwd": (c_int, [c_void_p, c_int, c_void_p, c_float, C.c_double, C.c_double, C.c_double, c_float, c_void_p,
                                    c_void_p, c_void_p]),
    "cgs_rate_finish_bwd": (c_int, [c_void_p, c_int, c_float, C.c_double, C.c_double, C.c_double, c_void_p, c_void_p, c_void_p]),
    "cgs_rate_finish_bwd4": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, C.c_double, C.c_double, C.c_double, c_void_p,
                                     c_void_p, c_void_p]),
    "cgs_ctx_choose_blocks": (c_size_t, [c_int64]),
    "cgs_ctx_choose_flags": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, C.c_uint64, c_float, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_ctx_choose_slot_ints": (c_int, []),
    "cgs_ctx_choose_flags_slots": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, C.c_uint64, c_float, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "cgs_ctx_choose_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p]),
    "cgs_means_accum_doubles": (c_size_t, []),
    "cgs_means_finalize": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "cgs_noise_quant_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                    C.c_uint64, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_ctx_level_fwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                  c_int64] + [c_void_p] * 8 + [C.c_uint64, c_float, c_float, c_float] + [c_void_p] * 7),
    "cgs_ctx_level_bwd_scratch_bytes": (c_size_t, []),
    "cgs_ctx_level_bwd": (c_int, [c_int] + [c_void_p] * 9 + [c_int64, C.c_uint64, c_float, c_float, c_float, c_void_p, c_int64] +
                          [c_void_p] * 4 + [c_int64] + [c_void_p] * 11 + [c_size_t, c_void_p]),
    "cgs_ctx_level_bwd2": (c_int, [c_int] + [c_void_p] * 9 + [c_int64, C.c_uint64, c_float, c_float, c_float, c_void_p, c_int64] +
                           [c_void_p] * 4 + [c_int64] + [c_void_p] * 12 + [c_size_t, c_void_p]),
    "cgs_level_rate_fwd": (c_int, [c_void_p] * 9 + [c_int, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p]),
    "cgs_level_rate_bwd": (c_int, [c_void_p] * 9 + [c_int, c_int64, c_int, c_int, c_int64] + [c_void_p] * 7 +
                           [c_int, c_void_p]),
    "cgs_rate_sub_fwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 10 + [c_int, c_void_p, c_void_p]),
    "cgs_rate_sub_bwd_scratch_bytes": (c_size_t, [c_int, c_int64]),
    "cgs_rate_sub_bwd": (c_int, [c_int, c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 10 + [c_int] + [c_void_p] * 12 +
                         [c_size_t, c_void_p]),
    "cgs_eb_likelihood_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "cgs_eb_likelihood_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_ac_max_bytes": (c_size_t, [c_int64]),
    "cgs_cdf_float_to_u16_host": (c_int, [c_void_p, c_int64, c_int, c_void_p]),
    "cgs_ac_encode_table_host": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_size_t, C.POINTER(c_size_t)]),
    "cgs_ac_decode_table_host": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_ac_encode_const_host": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_size_t, C.POINTER(c_size_t)]),
    "cgs_ac_decode_const_host": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_size_t, c_void_p]),
    "cgs_gaussian_stream_minmax": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_gaussian_ac_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_gaussian_ac_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_gaussian_ac_encode_lanes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_gaussian_ac_decode_lanes": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_lanes_compact": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "cgs_lanes_block_slot_bytes": (c_size_t, [c_int64, c_int]),
    "cgs_table_ac_encode_lanes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_table_ac_decode_lanes": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_bernoulli_ac_encode": (c_int, [c_void_p, C.c_uint32, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_bernoulli_ac_decode": (c_int, [C.c_uint32, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "cgs_means3_scratch_bytes": (c_size_t, []),
    "cgs_means3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_size_t, c_void_p,
                           c_void_p]),
    "cgs_mask_ste_fwd":(c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_mask_ste_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_knn_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_knn_mean_dist2": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cgs_streams_compact":(c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "cgs_pread_ranges": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cgs_pwrite_ranges": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "cgs_gaussian_cdf_table": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "cgs_rans_max_bytes": (c_size_t, [c_int64]),
    "cgs_rans_encode_host": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                     c_size_t, C.POINTER(c_size_t)]),
    "cgs_rans_decode_host": (c_int, [c_void_p, c_size_t, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p]),
    "cgs_rans_decode_rows_host": (c_int, [c_void_p, c_size_t, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                          c_void_p, c_void_p, c_int64]),
    "cgs_prof_enable": (c_int, [c_int]),
    "cgs_prof_count": (c_int, []),
    "cgs_prof_name": (C.c_char_p, [c_int]),
    "cgs_prof_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(c_int64)]),
    "cgs_densify_stats": (c_int, [c_int64, c_int] + [c_void_p] * 11),
    "cgs_nonzero_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_nonzero_launch": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p, C.POINTER(C.c_uint64)]),
    "cgs_nonzero_wait": (c_int, [C.c_uint64, C.POINTER(c_int64)]),
    "cgs_level_key_range": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "cgs_level_unique_scratch_bytes": (c_size_t, [c_int64]),
    "cgs_level_unique": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 C.POINTER(c_int64), c_void_p, c_size_t, c_void_p]),
    "cgs_compact_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "cgs_l1_ssim_partials": (c_size_t, [c_int, c_int, c_int]),
    "cgs_l1_ssim_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "cgs_l1_ssim_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cgs_l1_ssim_finish": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    "cgs_l1_ssim_bwd_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "cgs_reg_partials": (c_size_t, [c_int64]),
    "cgs_scaling_reg_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_scaling_reg_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_sigmoid_mean_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_sigmoid_mean_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "cgs_anchor_gen_count": (c_int, [c_void_p] * 14 + [c_int64, c_int, C.POINTER(c_int64), c_void_p]),
    "cgs_anchor_gen_write": (c_int, [c_void_p] * 20 + [c_int64, c_int, c_void_p]),
    "cgs_anchor_gen_bwd_scratch_bytes": (c_size_t, [c_int64, c_int]),
    "cgs_anchor_gen_backward": (c_int, [c_void_p] * 32 + [c_int64, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "cgs_expand_scratch_bytes": (c_size_t, [c_int64, c_int]),
    "cgs_expand_count": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_size_t, C.POINTER(c_int64), c_void_p]),
    "cgs_expand_count_launch": (c_int, [c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p, C.POINTER(C.c_uint64)]),
    "cgs_expand_count_wait": (c_int, [C.c_uint64, C.POINTER(c_int64)]),
    "cgs_expand_write": (c_int, [c_int64, c_int] + [c_void_p] * 15),
    "cgs_expand_backward": (c_int, [c_int64, c_int] + [c_void_p] * 22),
}


def lib() -> C.CDLL:
    """Load libcgs_hip.so (once).  Raises if it is absent: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch FIRST: it ships its own libamdhip64, and whichever HIP runtime is loaded first is the one the process uses.
        # With libcgs_hip.so loaded before torch (e.g. __graft_entry__.build() followed by smoke() in one process) the two
        # ended up with a runtime each and this library's saw "no ROCm-capable device".
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found. Build it with `python -m contextgs_amd.build` "
                "(or __graft_entry__.build()); contextgs_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError => ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if os.environ.get("CGS_LIB_PATH"):
            import sys
            from . import build as _build
            info = (handle.cgs_build_info() or b"").decode()
            want = _build.source_digest()
            if info.split("|")[0] != want and not os.environ.get("CGS_LIB_ALLOW_STALE"):
                raise RuntimeError(f"CGS_LIB_PATH={LIB_PATH} was built from other sources ({info.split('|')[0][:12]}... vs "
                                   f"{want[:12]}...): rebuild it (tools/variant_lib.sh) or set CGS_LIB_ALLOW_STALE=1")
            print(f"[contextgs_amd] variant library {LIB_PATH} ({info[:12]}...|{info.split('|')[-1]})", file=sys.stderr)
        if os.environ.get("CGS_BIN_MODE"):      # measurement switch (tools/bin_ab.sh): 1 = radix passes, 2 = two-level tile binning
            if handle.cgs_debug_set_bin_mode(int(os.environ["CGS_BIN_MODE"])) != 0:
                raise RuntimeError("CGS_BIN_MODE must be 0, 1 or 2")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().cgs_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL). The tensor must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "libcgs_hip takes dense row-major buffers"
    return t.data_ptr()


_raw_stream = None


def current_stream() -> int:
    """hipStream_t of torch's current stream on the current device (the raw handle: torch.cuda.current_stream() builds a
    Stream object per call, ~3 us, and a training step asks ~40 times)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def require_device(*tensors) -> None:
    """Operands must live on the process's CURRENT HIP device: kernels are enqueued on that device's current stream
    (current_stream()), so a tensor of another GPU would be a foreign pointer on an unrelated stream."""
    import torch
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("contextgs_amd operators run on the HIP device only (tensor on %s); "
                               "there is no CPU path" % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: call "
                               "torch.cuda.set_device(local_rank) first (one process per GPU)")
