"""Hot-path half of the reference's `GaussianModel` (scene/gaussian_model.py).

Carries exactly the state tensors, accessors and MLP definitions the
render-and-compress path reads (SURVEY §8a rows a6/a7 and the state-tensor
table): same attribute and property names, same shapes, same activations, so
`render()` / `multi_scale_generating()` / the codec accept either this class
or the reference's own GaussianModel.  The SURVEY §8(f) widenings live next to
this class: the optimizer set-up and densification (`training_setup`,
`update_learning_rate`, `training_statis`, `adjust_anchor`, `prune_anchor`,
`cat_tensors_to_optimizer` -> densify.py), anchor initialisation (knn.py,
`create_from_pcd`) and the ply files (ply_io.py).

Cited lines are scene/gaussian_model.py unless stated otherwise.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .encodings import Quantize_anchor, Quantize_anchor_attach
from .entropy_bottleneck import EntropyBottleneck
from .entropy_models import Entropy_gaussian

ANCHOR_Q_CACHE = os.environ.get("CGS_ANCHOR_Q_CACHE", "1") != "0"      # get_anchor: one quantisation per version of the anchors (A/B knob)


def expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000, step_sub=0):
    """Learning-rate schedule of utils/general_utils.py:49-82: log-linear interpolation from lr_init (step = step_sub)
    to lr_final (step = max_steps), optionally eased in over lr_delay_steps; 0 for a negative step or a disabled group."""
    import math

    def at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0.0), 1.0))
        t = min(max((step - step_sub) / (max_steps - step_sub), 0.0), 1.0)
        return delay * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)

    return at


class GaussianModel(nn.Module):
    def __init__(self, feat_dim: int = 50, n_offsets: int = 10, voxel_size: float = 0.01, level_num: int = 3,
                 hyper_divisor: int = 4, target_ratio: float = 0.2, adaptQ_per_channel: bool = False,
                 disable_hyper: bool = False, decoded_version: bool = False, device="cuda"):
        super().__init__()
        self.feat_dim, self.n_offsets, self.voxel_size = feat_dim, n_offsets, voxel_size
        self.level_num, self.hyper_divisor, self.target_ratio = level_num, hyper_divisor, target_ratio
        self.adaptQ_per_channel, self.disable_hyper = adaptQ_per_channel, disable_hyper
        self.decoded_version = decoded_version
        self.level_scale = None
        self.use_feat_bank = False
        dev = torch.device(device)
        self.x_bound_min = torch.zeros(1, 3, device=dev)           # :93-94
        self.x_bound_max = torch.ones(1, 3, device=dev)
        H = feat_dim // hyper_divisor

        e = torch.empty(0, device=dev)
        self._anchor = e; self._offset = e; self._mask = e; self._anchor_feat = e
        self._hyper_latent = e; self._scaling = e; self._rotation = e; self._opacity = e

        self.latent_codec = EntropyBottleneck(channels=H).to(dev)    # :135
        D = feat_dim
        self.mlp_opacity = nn.Sequential(nn.Linear(D + 4, D), nn.ReLU(True), nn.Linear(D, n_offsets), nn.Tanh()).to(dev)      # :153-158
        self.mlp_cov = nn.Sequential(nn.Linear(D + 4, D), nn.ReLU(True), nn.Linear(D, 7 * n_offsets)).to(dev)                  # :161-166
        self.mlp_color = nn.Sequential(nn.Linear(D + 4, D), nn.ReLU(True), nn.Linear(D, 3 * n_offsets), nn.Sigmoid()).to(dev)  # :169-174
        self.mlp_grid = nn.ModuleList()                                                                                         # :177-188
        out_dim = (D + 6 + 3 * n_offsets) * 2 + 3
        ctx_dim = D + 6 + 3
        for i in range(level_num):
            in_dim = H + 3 if i == level_num - 1 else ctx_dim + H
            self.mlp_grid.append(nn.Sequential(nn.Linear(in_dim, 2 * D), nn.ReLU(True), nn.Linear(2 * D, out_dim)).to(dev))
        self.entropy_gaussian = Entropy_gaussian(Q=1)                 # :190
        self.rotation_activation = torch.nn.functional.normalize      # :58
        # caches (SURVEY §8a "caching opportunity"): see context_model.divide_levels_cached
        self._level_cache = None
        self._anchor_q_cache = None

    # ---- train / eval switches (:200-220) ------------------------------------
    def eval(self):
        for m in (self.mlp_opacity, self.mlp_cov, self.mlp_color, self.latent_codec, self.mlp_grid):
            m.eval()
        return self

    def train(self, mode: bool = True):
        for m in (self.mlp_opacity, self.mlp_cov, self.mlp_color, self.latent_codec, self.mlp_grid):
            m.train(mode)
        return self

    # ---- accessors (:288-345) --------------------------------------------------
    @property
    def get_scaling(self):
        if self.decoded_version:
            return self._scaling
        out = torch.exp(self._scaling)           # the reference's `1.0 *` (:291) is an exact no-op: not launched
        out._cgs_exp_of = self._scaling          # lets the rate model recognise "this is get_scaling of that parameter"
        return out

    def get_mask_pair(self):
        """(get_mask, get_mask_anchor) from ONE evaluation (:295-310): the training step needs both."""
        if self.decoded_version:
            return self._mask, torch.sum(self._mask.detach(), dim=1)[:, 0] > 0
        if self._mask.is_cuda:
            from .ctx_ops import mask_ste
            return mask_ste(self._mask)                   # one HIP launch each way (csrc/mask.hip)
        s = torch.sigmoid(self._mask)
        m = ((s > 0.01).float() - s).detach() + s
        return m, torch.sum(m.detach(), dim=1)[:, 0] > 0

    @property
    def get_mask(self):
        return self.get_mask_pair()[0]

    @property
    def get_mask_anchor(self):
        with torch.no_grad():
            return self.get_mask_pair()[1]

    @property
    def get_opacity_mlp(self): return self.mlp_opacity
    @property
    def get_cov_mlp(self): return self.mlp_cov
    @property
    def get_color_mlp(self): return self.mlp_color
    @property
    def get_grid_mlp(self): return self.mlp_grid

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_anchor(self):
        if self.decoded_version:
            return self._anchor
        # (round 6) one quantisation per VERSION of the parameter and the bounds: a training step asks twice (prefilter_voxel without
        # a graph, render() with one — the reference's property re-derives it on every access, scene/gaussian_model.py:296-300); the
        # second call wraps the first call's values in the straight-through node.  Keyed like the level plan and _rot0_cache: object
        # identity + version counter (+ the storage, for a replaced .data); whoever rewrites the anchors clears _level_cache and this.
        a, lo, hi = self._anchor, self.x_bound_min, self.x_bound_max
        key = (a, a._version, a.data_ptr(), lo, lo._version, lo.data_ptr(), hi, hi._version, hi.data_ptr())
        hit = getattr(self, "_anchor_q_cache", None)
        if ANCHOR_Q_CACHE and hit is not None and len(hit[0]) == len(key) and all(
                (x is y) if isinstance(x, torch.Tensor) else (x == y) for x, y in zip(hit[0], key)):
            if torch.is_grad_enabled() and a.requires_grad:
                return Quantize_anchor_attach.apply(a, hit[1])
            return hit[1]
        anchor, _q = Quantize_anchor.apply(a, lo, hi)
        self._anchor_q_cache = (key, anchor.detach()) if (ANCHOR_Q_CACHE and a.is_cuda) else None
        return anchor

    @torch.no_grad()
    def update_anchor_bound(self):                                  # :351-361
        lo = torch.min(self._anchor, dim=0, keepdim=True)[0].detach()
        hi = torch.max(self._anchor, dim=0, keepdim=True)[0].detach()
        self.x_bound_min = torch.where(lo < 0, lo * 1.2, lo * 0.8)
        self.x_bound_max = torch.where(hi > 0, hi * 1.2, hi * 0.8)
        self._level_cache = None
        self._anchor_q_cache = None

    def get_mlp_size(self, digit: int = 32):                        # :193-198
        n = sum(p.numel() for name, p in self.named_parameters() if "mlp" in name and "deform" not in name)
        return n * digit, n * digit / 8 / 1024 / 1024

    # ---- state -------------------------------------------------------------------
    def set_state(self, anchor, offset, mask, feat, hyper, scaling, rotation=None, requires_grad=True):
        """Install the per-anchor parameters (shapes of :399-423)."""
        dev = self.x_bound_min.device
        P = lambda t, g=True: nn.Parameter(torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous(),
                                           requires_grad=requires_grad and g)
        N = anchor.shape[0]
        self._anchor = P(anchor)
        self._offset = P(offset)
        self._mask = P(mask)
        self._anchor_feat = P(feat)
        self._hyper_latent = P(hyper)
        self._scaling = P(scaling)
        if rotation is None:
            rotation = torch.zeros(N, 4)
            rotation[:, 0] = 1
        self._rotation = P(rotation, False)
        self._opacity = P(torch.zeros(N, 1), False)
        self._level_cache = None
        self._anchor_q_cache = None
        return self

    # ---- initialisation and ply I/O --------------------------
    def create_from_pcd(self, pcd, spatial_lr_scale: float = 1.0):      # :382-423
        """pcd: a BasicPointCloud (uses .points) or an [N,3] array / tensor.  voxel_size <= 0 selects the median
        3-NN distance (:388-391).  kNN and voxelisation run on the device (knn.py, csrc/knn.hip)."""
        from . import knn
        self.spatial_lr_scale = spatial_lr_scale
        pts = getattr(pcd, "points", pcd)
        pts = pts if isinstance(pts, torch.Tensor) else torch.as_tensor(pts)
        dev = self.x_bound_min.device
        H = self.feat_dim // self.hyper_divisor
        (self.voxel_size, anchor, offset, mask, feat, hyper, scaling, rot, opacity) = knn.init_from_points(
            pts.to(dev), float(self.voxel_size), self.n_offsets, self.feat_dim, H)
        print(f"Initial voxel_size: {self.voxel_size}")
        print("Number of points at initialisation : ", anchor.shape[0])
        self.set_state(anchor, offset, mask, feat, hyper, scaling, rot)
        self._opacity = nn.Parameter(opacity, requires_grad=False)
        self.max_radii2D = torch.zeros(anchor.shape[0], device=dev)
        return self

    # ---- optimizer + densification (:426-559, :656-910) ---------------------------------------------
    update_depth, update_init_factor, update_hierachy_factor, ste_binary = 3, 100, 4, True     # ctor defaults (:65-72)

    def training_setup(self, training_args):                             # :426-525
        dev = self._anchor.device
        N, K = self._anchor.shape[0], self.n_offsets
        self.percent_dense = training_args.percent_dense
        self.opacity_accum = torch.zeros(N, 1, device=dev)
        self.offset_gradient_accum = torch.zeros(N * K, 1, device=dev)
        self.offset_denom = torch.zeros(N * K, 1, device=dev)
        self.anchor_demon = torch.zeros(N, 1, device=dev)
        a, s = training_args, getattr(self, "spatial_lr_scale", 0.0)
        groups = [
            {"params": [self._anchor], "lr": a.position_lr_init * s, "name": "anchor"},
            {"params": [self._offset], "lr": a.offset_lr_init * s, "name": "offset"},
            {"params": [self._mask], "lr": a.mask_lr_init * s, "name": "mask"},
            {"params": [self._anchor_feat], "lr": a.feature_lr, "name": "anchor_feat"},
            {"params": [self._hyper_latent], "lr": a.hyper_latent_lr, "name": "hyper_latent"},
            {"params": [self._opacity], "lr": a.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": a.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": a.rotation_lr, "name": "rotation"},
            {"params": self.mlp_opacity.parameters(), "lr": a.mlp_opacity_lr_init, "name": "mlp_opacity"},
            {"params": self.mlp_cov.parameters(), "lr": a.mlp_cov_lr_init, "name": "mlp_cov"},
            {"params": self.mlp_color.parameters(), "lr": a.mlp_color_lr_init, "name": "mlp_color"},
            {"params": self.latent_codec.parameters(), "lr": a.latent_codec_lr_init, "name": "latent_codec"},
            {"params": self.mlp_grid.parameters(), "lr": a.mlp_grid_lr_init, "name": "mlp_grid"},
        ]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        late = 0 if self.ste_binary else 10000
        sched = lambda pre, scale=1.0, step_sub=0: expon_lr_func(
            getattr(a, pre + "_lr_init") * scale, getattr(a, pre + "_lr_final") * scale,
            lr_delay_mult=getattr(a, pre + "_lr_delay_mult"), max_steps=getattr(a, pre + "_lr_max_steps"), step_sub=step_sub)
        self._schedulers = {"anchor": sched("position", s), "offset": sched("offset", s), "mask": sched("mask", s),
                            "mlp_opacity": sched("mlp_opacity"), "mlp_cov": sched("mlp_cov"), "mlp_color": sched("mlp_color"),
                            "latent_codec": sched("latent_codec", step_sub=late), "mlp_grid": sched("mlp_grid", step_sub=late)}

    def update_learning_rate(self, iteration):                           # :527-559
        for group in self.optimizer.param_groups:
            f = self._schedulers.get(group["name"])
            if f is not None:
                group["lr"] = f(iteration)

    def cat_tensors_to_optimizer(self, tensors_dict):                    # :673-694
        from . import densify
        return densify.cat_tensors_to_optimizer(self, tensors_dict)

    def replace_tensor_to_optimizer(self, tensor, name):                 # :656-670
        from . import densify
        return densify.replace_tensor_to_optimizer(self, tensor, name)

    def training_statis(self, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask):
        from . import densify                                            # :696-713
        densify.training_statis(self, viewspace_point_tensor, opacity, update_filter, offset_selection_mask, anchor_visible_mask)

    def anchor_growing(self, grads, threshold, offset_mask, rand_fn=None):   # :762-855
        from . import densify
        densify.anchor_growing(self, grads, threshold, offset_mask, rand_fn)

    def prune_anchor(self, mask):                                        # :747-760
        from . import densify
        densify.prune_anchor(self, mask)

    def reduce_statistics(self):
        """Multi-GPU extension: global densification statistics on every rank (densify.reduce_statistics)."""
        from . import densify
        densify.reduce_statistics(self)

    def adjust_anchor(self, check_interval=100, success_threshold=0.8, grad_threshold=0.0002, min_opacity=0.005,
                      rand_fn=None, reduce_stats=True):                  # :856-910
        from . import densify
        densify.adjust_anchor(self, check_interval, success_threshold, grad_threshold, min_opacity, rand_fn, reduce_stats)

    def save_ply(self, path):                                           # :579-598
        import os
        from . import ply_io
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        ply_io.save_model_ply(path, self._anchor, self._offset, self._mask, self._anchor_feat, self._hyper_latent,
                              self._opacity, self._scaling, self._rotation)

    def load_ply_sparse_gaussian(self, path):                           # :600-654
        from . import ply_io
        d = ply_io.load_model_ply(path)
        self.set_state(d["anchor"], d["offset"], d["mask"], d["feat"], d["hyper"], d["scaling"], torch.from_numpy(d["rotation"]))
        dev = self.x_bound_min.device
        self._rotation.requires_grad_(True)                              # the reference re-enables both on load
        self._opacity = nn.Parameter(torch.from_numpy(d["opacity"]).to(dev), requires_grad=True)
        return self

    # ---- checkpoint tuple (scene/gaussian_model.py:222-286; train.py saves it with torch.save) --------------------
    def capture(self):
        """The reference's 19-entry checkpoint tuple, same order.  The optimizer entry is `self.optimizer.state_dict()`
        when a training driver has attached one (the driver and its optimizer are the reference's own, SURVEY 2 rows
        12-23) and None otherwise."""
        self.latent_codec.update()
        opt = getattr(self, "optimizer", None)
        return (self._anchor, self._anchor_feat, self._hyper_latent, self._offset, self._mask, self._scaling,
                self._rotation, self._opacity, getattr(self, "max_radii2D", None),
                opt.state_dict() if opt is not None else None, getattr(self, "spatial_lr_scale", 0.0),
                self.mlp_opacity.state_dict(), self.mlp_cov.state_dict(), self.mlp_color.state_dict(),
                self.latent_codec.state_dict(), self.mlp_grid.state_dict(),
                self.x_bound_min, self.x_bound_max, self.level_scale)

    def restore(self, model_args, training_args=None):
        """Inverse of capture().  `training_setup(training_args)` is called when the object has one (the reference's
        class does, :252); the optimizer state is loaded when both a state and an optimizer exist."""
        (anchor, feat, hyper, offset, mask, scaling, rotation, opacity, self.max_radii2D, opt_dict, self.spatial_lr_scale,
         sd_opacity, sd_cov, sd_color, sd_codec, sd_grid, self.x_bound_min, self.x_bound_max, self.level_scale) = model_args
        dev = self.x_bound_min.device
        as_param = lambda t, like: nn.Parameter(t.detach().to(dev).float().contiguous(),
                                                requires_grad=bool(getattr(t, "requires_grad", like)))
        self._anchor, self._anchor_feat = as_param(anchor, True), as_param(feat, True)
        self._hyper_latent, self._offset = as_param(hyper, True), as_param(offset, True)
        self._mask, self._scaling = as_param(mask, True), as_param(scaling, True)
        self._rotation, self._opacity = as_param(rotation, False), as_param(opacity, False)
        self.latent_codec.update()
        if training_args is not None and hasattr(self, "training_setup"):
            self.training_setup(training_args)
        if opt_dict is not None and getattr(self, "optimizer", None) is not None:
            self.optimizer.load_state_dict(opt_dict)
        self.mlp_opacity.load_state_dict(sd_opacity)
        self.mlp_cov.load_state_dict(sd_cov)
        self.mlp_color.load_state_dict(sd_color)
        from .codec_driver import _load_latent_codec
        _load_latent_codec(self.latent_codec, sd_codec)
        self.latent_codec.update(force=True)
        self.mlp_grid.load_state_dict(sd_grid)
        self._level_cache = None
        self._anchor_q_cache = None
        return self

    # the codec / rate-report methods live in codec_driver.py and are bound here so the
    # reference's call sites (train.py:301-314, test.py:168-177) work unchanged.
    def estimate_final_bits(self):
        from .codec_driver import estimate_final_bits
        return estimate_final_bits(self)

    def conduct_encoding(self, pre_path_name):
        from .codec_driver import conduct_encoding
        return conduct_encoding(self, pre_path_name)

    def conduct_decoding(self, pre_path_name):
        from .codec_driver import conduct_decoding
        return conduct_decoding(self, pre_path_name)
