"""Bitstream container: `conduct_encoding`, `conduct_decoding`, `estimate_final_bits`,
`save/load_mlp_checkpoints` (scene/gaussian_model.py:912-951,
980-1004, 1007-1295, 1299-1539 — cited lines are that file).

Same container as the reference — anchor.npy (uint16 [N_valid,3]), hyper.b,
feat{l}.b / scaling{l}.b / offsets{l}.b (1000-anchor chunk streams concatenated,
byte lengths in meta), masks.b, meta.b (a torch.save'd 14-item list, :1277), mlp.pt —
and the same level / chunk / mask order.  What differs is how it is scheduled: instead
of a Python loop of ~3 N/1000 serial torchac calls, each shipping a [50 000, L] float
table over PCIe (:1192-1232), chunk streams are coded concurrently by device launches,
one wave per stream (codec.gaussian_encode_groups / gaussian_decode_groups):
  * encoding never reads coded bytes back, so the level loop only predicts and
    quantises, and ALL streams (3 levels x feat/scaling/offsets) go through one launch
    whose duration is that of its longest stream;
  * decoding needs level l's feat+scaling before level l+1's context, so those are one
    launch per level; offsets feed no context and are decoded for all levels at the end;
  * the format's serial host streams (the mask stream, the 10 000-anchor hyper rANS
    strings) run on host threads next to the device work (codec.host_pool).

Deliberate deviation: the reference's decoder raises IndexError when fewer
than 10 000 anchors are valid (:1322-1331); this one decodes any size.
"""
from __future__ import annotations

import os
import re
import time

import numpy as np
import torch
import torch.nn as nn

from . import codec
from . import dist as mgpu
from .context_model import (context_rows, extract_context_feat, find_divide_scale, grid_mlp, level_plan, multi_scale_generating,
                            split_prediction)
from .encodings import Q_anchor, Quantize_anchor, STE_multistep, decoder, encoder

bit2MB_scale = 8 * 1024 * 1024
MAX_BATCH = 1_000                                              # :1071

# Container versions.  1 = the reference's container (default): one serial arithmetic-coded mask stream, 10 000-anchor hyper
# strings, 1000-anchor chunk streams for every Gaussian-coded attribute, a 14-item meta list.  2 = the same files, symbols,
# order and coder, re-cut for a device: the masks are 1000-anchor chunk streams coded by the device coder
# (codec.BernoulliEncodeJob; no serial host stream is left on either critical chain), the hyper latents go through the device
# table coder instead of host rANS strings (EntropyBottleneck.compress_lanes: same tables), and feat / scaling / offsets are cut
# into BLOCKS of V2_BLOCK consecutive symbols, each coded as 64 interleaved lane streams by one wave (csrc/codec.hip,
# "Lane-parallel Gaussian codec": the coder arithmetic runs 64-wide in vector registers instead of one serial chain per
# wave on the scalar unit).  meta.b gets a 15th item {"version": 2, "block_symbols": ..., "chunk": {"masks": ...},
# "bit_masks": [...]}; the per-stream lists of the header (bits, min, max) are per block; a 14-item list is version 1.
CONTAINER_VERSION = 1
V2_BLOCK = 64 * 512                      # symbols per block: 512 per lane stream
V2_CHUNK = {"masks": 1000}               # anchors per mask chunk stream
V2_HYPER_BLOCK = 64 * 512                # anchors of one channel per hyper.b block (lane-parallel table coder)


# 1 = the groups coded in 2-4 ranges, a range's files handed to the writers while the next range is coded (codec.gaussian_encode_groups(
# on_range=)).  Built and measured in round 6: no gain — 17.4-17.7 ms against 16.9-18.0 ms at 1 M anchors; the encode ends when the
# LARGEST file (feat0.b, 60 MB) is in the page cache, buffered writes to one file serialise on its inode lock (~10 GB/s whatever the
# number of writers), and under the coder's downloads they run slower still (profiles/r06_codec_ranges.txt).  Off.
RANGED_ENCODE = os.environ.get("CGS_RANGED_ENCODE", "0") != "0"


def default_container_version():
    return int(os.environ.get("CGS_CONTAINER_VERSION", CONTAINER_VERSION))


def save_mlp_checkpoints(pc, path):                            # :912-936
    pc.latent_codec.update()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({"opacity_mlp": pc.mlp_opacity.state_dict(), "cov_mlp": pc.mlp_cov.state_dict(),
                "color_mlp": pc.mlp_color.state_dict(), "latent_codec": pc.latent_codec.state_dict(),
                "grid_mlp": pc.mlp_grid.state_dict(), "bound": [pc.x_bound_min, pc.x_bound_max],
                "level_scale": pc.level_scale}, path)


def load_mlp_checkpoints(pc, path, ck=None, tr=None, stored_tables=False):          # :939-950
    """ck: an already loaded checkpoint dict; tr: the driver's tracer (CGS_CODEC_TRACE).
    stored_tables (container version 2): take the hyper prior's integer CDF tables the ENCODER used from the checkpoint
    (`_quantized_cdf` / `_cdf_length` / `_offset`: buffers of the state dict torch.save wrote) instead of rebuilding them with
    update(force=True) as the reference does (:946-947) — the decoder then codes with exactly the encoder's tables whatever
    device evaluated the density, and skips ~2 ms of its prologue.  A checkpoint without them falls back to update()."""
    tr = tr or (lambda label: None)
    if ck is None:
        ck = read_mlp_checkpoint(path)
        tr("mlp.pt unpickled")
    # restored on the HOST, then every tensor of the checkpoint goes to the device in ONE pinned non-blocking copy
    # (restoring ~70 small tensors on the device is ~70 blocking pageable copies: 3 ms of the decoder's prologue)
    ck = _tensors_to_device(ck, pc.x_bound_min.device)
    tr("mlp.pt on the device")
    pc.mlp_opacity.load_state_dict(ck["opacity_mlp"])
    pc.mlp_cov.load_state_dict(ck["cov_mlp"])
    pc.mlp_color.load_state_dict(ck["color_mlp"])
    _load_latent_codec(pc.latent_codec, ck["latent_codec"])
    tr("mlp.pt: state dicts loaded")
    lc, eb = ck["latent_codec"], pc.latent_codec
    tabs = [lc.get(k) for k in ("_quantized_cdf", "_cdf_length", "_offset")] if stored_tables else [None]
    if all(isinstance(t, torch.Tensor) and t.numel() > 0 for t in tabs) and tabs[0].dim() == 2 \
            and tabs[0].shape[0] == tabs[1].numel() == tabs[2].numel() == eb.channels:
        dev = eb.quantiles.device
        eb._quantized_cdf, eb._cdf_length, eb._offset = (t.to(device=dev, dtype=torch.int32) for t in tabs)
        tr("prior tables taken from the checkpoint")
    else:
        eb.update(force=True)
        tr("prior tables rebuilt")
    pc.mlp_grid.load_state_dict(ck["grid_mlp"])
    pc.x_bound_min, pc.x_bound_max = ck["bound"]
    pc.level_scale = ck["level_scale"]


_STORAGE_DTYPES = {"FloatStorage": torch.float32, "DoubleStorage": torch.float64, "HalfStorage": torch.float16,
                   "BFloat16Storage": torch.bfloat16, "LongStorage": torch.int64, "IntStorage": torch.int32,
                   "ShortStorage": torch.int16, "CharStorage": torch.int8, "ByteStorage": torch.uint8, "BoolStorage": torch.bool}


class _StorageRef:
    __slots__ = ("dtype", "key", "numel")

    def __init__(self, dtype, key, numel):
        self.dtype, self.key, self.numel = dtype, key, numel


def _zip_members(buf):
    """{name: (data offset, size)} of an uncompressed, non-zip64 archive held in `buf` — the layout torch.save writes.
    (zipfile.ZipFile takes 1.2 ms for the 80 members of mlp.pt; this is two struct reads per member.)"""
    import struct
    eocd = buf.rfind(b"PK\x05\x06")
    if eocd < 0:
        raise ValueError("no end-of-central-directory record")
    n_ent, cd_size, cd_off = struct.unpack_from("<HII", buf, eocd + 10)
    if n_ent == 0xFFFF or cd_size == 0xFFFFFFFF or cd_off == 0xFFFFFFFF:
        raise ValueError("zip64 archive")
    out, p = {}, cd_off
    for _ in range(n_ent):
        if buf[p:p + 4] != b"PK\x01\x02":
            raise ValueError("bad central directory")
        method, = struct.unpack_from("<H", buf, p + 10)
        csize, usize, nlen, elen, clen = struct.unpack_from("<IIHHH", buf, p + 20)
        hoff, = struct.unpack_from("<I", buf, p + 42)
        name = bytes(buf[p + 46:p + 46 + nlen]).decode("utf-8")
        if method != 0 or csize != usize or 0xFFFFFFFF in (csize, hoff):
            raise ValueError("compressed or zip64 member")
        lnlen, lelen = struct.unpack_from("<HH", buf, hoff + 26)          # the LOCAL header's own name / extra lengths
        out[name] = (hoff + 30 + lnlen + lelen, usize)
        p += 46 + nlen + elen + clen
    return out


def _fast_checkpoint_load(path):
    """torch.load(path, map_location="cpu") for the files THIS container holds (mlp.pt, meta.b: nested dicts / lists of
    tensors, numpy arrays and Python numbers), without torch's per-tensor Python path (~60 us a tensor: 4.6 ms of the
    decoder's 23 for mlp.pt's 74): the archive is mapped once, tensors are views into the mapping.  Only the globals such a
    file needs are resolved; anything else raises and the caller falls back to torch.load."""
    import collections
    import io
    import pickle
    import mmap
    with open(path, "rb") as f:
        buf = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)       # private pages: the tensors below keep it alive
    members = _zip_members(buf)
    pkl = [n for n in members if n.endswith("/data.pkl")]
    if len(pkl) != 1:
        raise ValueError("not a torch.save archive")
    root = pkl[0][:-len("data.pkl")]

    def rebuild(st, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
        off, nbytes = members[root + "data/" + st.key]
        if st.numel * torch.empty(0, dtype=st.dtype).element_size() > nbytes:
            raise ValueError("storage shorter than its tensor")
        if st.numel == 0:
            return torch.empty(tuple(size), dtype=st.dtype)
        return torch.frombuffer(buf, dtype=st.dtype, count=st.numel, offset=off).as_strided(tuple(size), tuple(stride),
                                                                                           storage_offset)

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "torch._utils" and name == "_rebuild_tensor_v2":
                return rebuild
            if module == "torch._utils" and name == "_rebuild_parameter":
                return lambda data, requires_grad, backward_hooks: data
            if module == "torch" and name in _STORAGE_DTYPES:
                return _STORAGE_DTYPES[name]
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module.split(".")[0] == "numpy" or (module, name) == ("_codecs", "encode") or (
                    module in ("builtins", "__builtin__") and name in ("bytes", "bytearray", "complex", "set", "frozenset")):
                return super().find_class(module, name)                     # numpy arrays (protocol 2) and plain containers
            raise pickle.UnpicklingError(f"global {module}.{name} is not one this loader resolves")

        def persistent_load(self, pid):
            if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], torch.dtype)):
                raise pickle.UnpicklingError("unexpected persistent id")
            return _StorageRef(pid[1], str(pid[2]), int(pid[4]))

    off, nbytes = members[pkl[0]]
    return Unpickler(io.BytesIO(bytes(buf[off:off + nbytes]))).load()


def read_mlp_checkpoint(path):
    """mlp.pt -> the checkpoint dict with HOST tensors: _fast_checkpoint_load, else torch.load (mapped: it copies every
    storage out of the zip otherwise; a legacy non-zip file cannot be mapped and loads the plain way)."""
    try:
        return _fast_checkpoint_load(path)
    except Exception:
        pass
    try:
        return torch.load(path, map_location="cpu", weights_only=False, mmap=True)
    except (RuntimeError, ValueError):
        return torch.load(path, map_location="cpu", weights_only=False)


def _tensors_to_device(obj, dev):
    """Nested dict / list / tuple of CPU tensors -> the same structure of device tensors, one upload for all of them."""
    flat = []

    def collect(o):
        if isinstance(o, torch.Tensor):
            if o.device.type == "cpu":
                flat.append(o)
        elif isinstance(o, dict):
            for v in o.values():
                collect(v)
        elif isinstance(o, (list, tuple)):
            for v in o:
                collect(v)
    collect(obj)
    if not flat or dev.type != "cuda":
        return obj
    up = iter(codec._upload_small([t.detach().contiguous() for t in flat], dev, key="checkpoint"))

    def rebuild(o):
        if isinstance(o, torch.Tensor):
            return next(up) if o.device.type == "cpu" else o
        if isinstance(o, dict):
            return type(o)((k, rebuild(v)) for k, v in o.items())
        if isinstance(o, (list, tuple)):
            return type(o)(rebuild(v) for v in o)
        return o
    return rebuild(obj)


_LEGACY_EB_KEY = re.compile(r"^_?(matrix|bias|factor)(\d+)$")


def _load_latent_codec(codec, state):
    """load_state_dict for the hyper prior that refuses a checkpoint without its density parameters.

    strict=False stays (the reference loads that way, :946: buffers such as the quantised CDF tables are rebuilt by
    update()), but a checkpoint whose density parameters do not land — e.g. one written by a compressai release that
    names them `_matrix0` / `_bias0` / `_factor0` instead of `matrices.0` ... — would leave the prior at its random
    initialisation and decode garbage without a word.  Legacy names are remapped; anything still missing raises."""
    own = set(codec.state_dict().keys())
    remapped = {}
    for k, v in state.items():
        if k in ("_quantized_cdf", "_offset", "_cdf_length"):
            continue      # derived tables with data-dependent shapes: rebuilt from the parameters by update(force=True)
        m = _LEGACY_EB_KEY.match(k)
        if m and k not in own:
            k = {"matrix": "matrices", "bias": "biases", "factor": "factors"}[m.group(1)] + "." + m.group(2)
        remapped[k] = v
    res = codec.load_state_dict(remapped, strict=False)
    density = [k for k in res.missing_keys if k.split(".")[0] in ("matrices", "biases", "factors", "quantiles")]
    if density:
        raise RuntimeError("latent_codec checkpoint lacks the density parameters " + ", ".join(density) +
                           (f" (unexpected keys: {', '.join(res.unexpected_keys)})" if res.unexpected_keys else ""))
    return res


@torch.no_grad()
def estimate_final_bits(pc):                                   # :980-1004
    m = pc.get_mask_anchor
    sums = multi_scale_generating(pc, pc.get_anchor[m], pc._hyper_latent[m], pc._anchor_feat[m], pc._offset[m],
                                  pc.get_scaling[m], binary_grid_masks=pc.get_mask[m], predict_bpp=True,
                                  return_sum_bits=True)
    a, h, f, s, o, mk = sums
    mlp = pc.get_mlp_size()[0]
    r = lambda v: round(v / bit2MB_scale, 4)
    return (f"\nEstimated sizes in MB: anchor {r(a)}, feat {r(f)}, scaling {r(s)}, offsets {r(o)}, hyper {r(h)}, "
            f"masks {r(mk)}, MLPs {r(mlp)}, Total {r(a + f + s + o + h + mk + mlp)}")


V2_BLOCK_MIN = 64 * 128                  # shortest block of the per-group policy: 128 symbols per lane stream
V2_BLOCK_POLICY = 2                      # 0 / absent: every group in blocks of "block_symbols"; 1, 2: _block_for


def _block_for(n_sym, policy=V2_BLOCK_POLICY, base=None):
    """Symbols per block.  A coder launch lasts as long as its longest lane stream (block / 64 serial symbols), and the decoder's
    three level launches depend on each other: the two small levels would occupy a few dozen waves for 512-symbol chains.
    Policy 2 (what the encoder writes): ONE block size per LEVEL — n_sym = the level's feature symbols (anchors x 50), the same
    size for its scaling and offset groups, which ride in the same launch — halved until the level's features make >= 256 blocks
    (one wave per CU) or lane streams are 128 symbols long: at 1 M and at 500 k anchors the big level keeps 32 768-symbol
    blocks, the middle one gets 16 384, the small one 8 192 (a block costs a 128-byte header and 64 stream ends, ~0.6 % of its
    bytes at 32 768 symbols: +0.15 % of the container for the shorter blocks).
    Policy 1 (containers written earlier in round 5; decoded, no longer written): per GROUP, n_sym = the group's own symbols,
    halved until >= 1024 blocks.  Encoder and decoder derive the size from the counts in the header alone."""
    base = V2_BLOCK if base is None else int(base)
    if not policy:
        return base
    want = 1024 if int(policy) == 1 else 256
    b = base
    while b > V2_BLOCK_MIN and n_sym // b < want:
        b //= 2
    return b


def _level_blocks(n_level, n_group, D, policy, base=None):
    """Block size of a group of n_group symbols in a level of n_level anchors (see _block_for)."""
    return _block_for(n_group if int(policy or 0) == 1 else n_level * D, policy, base)


def _block_edges(n_sym, block=None):
    """Element offsets of the version-2 blocks of a group of n_sym symbols."""
    block = V2_BLOCK if block is None else int(block)
    return torch.tensor(list(range(0, n_sym, block)) + [n_sym] if n_sym > 0 else [0], dtype=torch.int64)


def _chunk_rows(n, batch=MAX_BATCH):
    edges = list(range(0, n, batch)) + [n]
    return edges if n > 0 else [0]


def _predict(pc, level, feat_in):
    (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Qf, Qs, Qo) = \
        split_prediction(pc, grid_mlp(pc, level, feat_in))
    c = lambda t: torch.clamp(t, min=1e-9).contiguous()
    return (mean_feat.contiguous(), c(scale_feat), mean_scaling.contiguous(), c(scale_scaling),
            mean_offsets.contiguous(), c(scale_offsets), Qf.reshape(-1).contiguous(), Qs.reshape(-1).contiguous(),
            Qo.reshape(-1).contiguous())


_SIDE = {}


def _side_stream(dev, tag=""):
    """One side stream per device (and purpose) for the life of the process (a fresh stream per container costs the caching
    allocator a fresh pool, see codec.StagedFiles)."""
    key = str(dev) + tag
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def _tracer(name):
    """CGS_CODEC_TRACE=1: wall-clock milestones of the container driver on stderr (host time, no extra synchronisation)."""
    if not os.environ.get("CGS_CODEC_TRACE"):
        return lambda label: None
    import sys
    sync = os.environ.get("CGS_CODEC_TRACE") == "2"       # 2: drain the device at every milestone (attributes device time)
    t0 = time.perf_counter()

    def mark(label):
        if sync:
            torch.cuda.synchronize()
        print(f"[{name} +{(time.perf_counter() - t0) * 1e3:7.1f} ms] {label}", file=sys.stderr)
    return mark


@torch.no_grad()
def conduct_encoding(pc, pre_path_name, container_version=None):   # :1007-1295
    version = default_container_version() if container_version is None else int(container_version)
    if version not in (1, 2):
        raise ValueError(f"container version {version}: 1 (the reference's container) or 2")
    chunk = dict(V2_CHUNK) if version == 2 else {"masks": MAX_BATCH}
    lanes = version == 2
    torch.cuda.synchronize(); t1 = time.time()
    tr = _tracer("encode")
    print("Start encoding ...")
    root = mgpu.rank() == 0          # multi-GPU: every rank predicts/quantises, codes its block of streams; rank 0 writes
    os.makedirs(pre_path_name, exist_ok=True)
    K, D = pc.n_offsets, pc.feat_dim
    path = lambda name: os.path.join(pre_path_name, name)

    # the mask stream (:1265-1269) is ONE serial arithmetic-coded stream and the longest chain of the encoder (10 M symbols on
    # a host thread): it is started FIRST, before the prior tables and the other per-anchor gathers
    mask_anchor = pc.get_mask_anchor
    _mask = pc.get_mask[mask_anchor]
    prob_masks = (_mask.sum() / _mask.numel()).item() if _mask.numel() else 0.5
    if root and version == 2:
        # version 2: the same symbols as 1000-anchor chunk streams, one device launch on a side stream beside everything below
        mask_edges = torch.tensor(_chunk_rows(int(_mask.shape[0]), chunk["masks"]), dtype=torch.int64) * K
        mask_job = codec.BernoulliEncodeJob(torch.floor(((_mask * 2 - 1).view(-1) + 1) / 2), prob_masks, mask_edges,
                                            _side_stream(_mask.device))
        tr("mask chunk streams enqueued (device)")
    elif root:
        mask_sym = codec.to_host_pinned(torch.floor(((_mask * 2 - 1).view(-1) + 1) / 2).to(torch.int16), "mask symbols")
        tr("mask symbols on the host")
        mask_job = codec.host_pool().submit(codec.bernoulli_encode_host, mask_sym, prob_masks)
        # hyper: 10 000-anchor rANS chunks (:1082-1098), on host threads too — but submitted only AFTER the level loop
        # has been enqueued: ~100 short jobs finishing on the pool make the main thread queue for the GIL, which
        # stretched the 7 ms of launch code below to 38 ms when they ran beside it
        hyper_jobs = None
    pc.latent_codec.update(force=True)
    tr("tables updated")
    _anchor, quantized_anchor = Quantize_anchor.apply(pc._anchor[mask_anchor], pc.x_bound_min, pc.x_bound_max)
    _feat = pc._anchor_feat[mask_anchor]
    _grid_offsets = pc._offset[mask_anchor]
    _scaling = pc.get_scaling[mask_anchor]
    _hyper_latent = pc._hyper_latent[mask_anchor]
    tr("valid anchors gathered")

    # Q3: the encoder feeds integer SYMBOLS to the context MLP (:1040,1164)
    hyper_feat = pc.latent_codec.quantize(_hyper_latent, "symbols", means=pc.latent_codec._get_medians().permute(1, 2, 0)[0])
    if pc.level_scale is None:
        pc.level_scale = find_divide_scale(pc, _anchor, pc.target_ratio, pc.level_num)
    plan, inverse_indices_list, mapping_list = level_plan(pc, _anchor, None)

    tr("level plan built")
    feat_after_Q = torch.zeros_like(_feat)
    grid_scaling_after_Q = torch.zeros_like(_scaling)
    already_coded = torch.zeros(_feat.shape[0], dtype=torch.bool, device=_feat.device)
    writes = []                       # file writes run on host threads while the device predicts / codes
    if root:
        anchor_u16 = quantized_anchor.cpu().numpy().astype(np.uint16)                       # :1100-1101
        writes.append(codec.host_pool().submit(np.save, path("anchor.npy"), anchor_u16))

    N_levels_list, groups, tags = [], [], []
    content_pre_gathered = None
    for (level, to_code, orig, hybrid_anchor) in plan:                                   # :1112
        n_l = int(orig.shape[0])
        N_levels_list.append(n_l)
        if content_pre_gathered is None:
            feat_in = torch.cat([_anchor[orig], hyper_feat[orig].float()], dim=1)
        else:
            feat_in = torch.cat([content_pre_gathered, hyper_feat[orig].float()], dim=1)
        (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Qf, Qs, Qo) = \
            _predict(pc, level, feat_in)
        tr(f"level {level}: predicted (enqueued)")
        rows = torch.tensor(_chunk_rows(n_l), dtype=torch.int64)

        tr(f"level {level}: chunk rows")
        feat_q = STE_multistep.apply(_feat[orig], Qf.unsqueeze(1))
        scal_q = STE_multistep.apply(_scaling[orig], Qs.unsqueeze(1))
        off_q = STE_multistep.apply(_grid_offsets[orig].reshape(n_l, 3 * K), Qo.unsqueeze(1))
        tr(f"level {level}: quantised")
        m30 = _mask[orig].repeat(1, 1, 3).reshape(n_l, 3 * K).to(torch.bool)              # :1222-1223
        cnt = torch.zeros(n_l + 1, dtype=torch.int64, device=m30.device)
        cnt[1:] = torch.cumsum(m30.sum(1), 0)
        if lanes:      # version 2: blocks of V2_BLOCK symbols, whatever anchors they belong to
            n_off = int(cnt[-1].item())
            B_l = _level_blocks(n_l, n_l * D, D, V2_BLOCK_POLICY)          # one block size for the level's three groups
            edges_f, edges_s = _block_edges(n_l * D, B_l), _block_edges(n_l * 6, B_l)
            off_edges = _block_edges(n_off, B_l)
        else:
            edges_f, edges_s = rows * D, rows * 6
            off_edges = cnt[rows.to(cnt.device)].cpu()
        tr(f"level {level}: offset stream edges on the host")
        live = torch.nonzero(m30.reshape(-1))[:, 0]              # ONE compaction index for the four operands
        pick = lambda t: t.reshape(-1).index_select(0, live)

        groups += [(feat_q, mean_feat, scale_feat, Qf, edges_f, D),
                   (scal_q, mean_scaling, scale_scaling, Qs, edges_s, 6),
                   (pick(off_q), pick(mean_offsets), pick(scale_offsets), Qo.index_select(0, live // (3 * K)),
                    off_edges, 1)]
        tags += [("feat", level), ("scaling", level), ("offsets", level)]
        tr(f"level {level}: groups built")

        feat_after_Q[orig] = feat_q                                                      # :1240-1242
        grid_scaling_after_Q[orig] = scal_q
        already_coded.index_fill_(0, orig, True)   # (x[idx] = True uploads its scalar through a blocking pageable copy)
        if level != 0:
            content_pre_gathered = context_rows(pc, _anchor, feat_after_Q, grid_scaling_after_Q, already_coded,
                                                inverse_indices_list, mapping_list, level)
            tr(f"level {level}: context of the next level gathered")

    tr("levels enqueued")
    hyper_v2 = None
    if root and version == 2:
        hyper_v2 = pc.latent_codec.compress_lanes(_hyper_latent.t().contiguous(), V2_HYPER_BLOCK)
        tr("hyper blocks coded (device) and on the host")
    elif root:
        hyper_jobs = pc.latent_codec.compress_chunks(_hyper_latent.t(), MAX_BATCH * 10, lazy=True)
        tr("hyper symbols on the host, rANS jobs submitted")
    torch.cuda.synchronize(); t0 = time.time()
    tr("levels done on the device")
    def write_mlp():
        # mlp.pt on THIS thread while the coder launch runs (its small device-to-host copies on a side stream, not behind
        # the launch).  A host thread of its own was tried: torch.save holds the GIL for milliseconds at a time and the level
        # loop above stretched from 8 to 10-35 ms waiting for it.
        if root:
            with torch.cuda.stream(_side_stream(_mask.device, "mlp")):
                save_mlp_checkpoints(pc, path("mlp.pt"))
    # (version 2, one process, a large model: the groups are coded in 2-4 ranges and a range's files are handed to the writers as soon
    #  as its stream lengths are known — its download and its page-cache writes run beside the next range's coder launch instead of
    #  behind the last one; tools/codec_trace.sh)
    early_written = set()

    def on_range(idxs, results, ready_r):
        for gi, (blob_r, _l, _mn, _mx) in zip(idxs, results):
            name_r, level_r = tags[gi]
            writes.extend(codec.write_file(path(f"{name_r}{level_r}.b"), blob_r, ready=ready_r))
            early_written.add(gi)
        tr(f"range of {len(idxs)} groups coded, writers started")
    coded = codec.gaussian_encode_groups(groups, staging=True, lanes=lanes, overlap=write_mlp, deferred=True,
                                         on_range=on_range if (root and lanes and RANGED_ENCODE) else None)   # blobs alias pinned buffers
    ready = codec.stage_ready()          # the download is still in flight: the file writers below wait piece by piece
    tr("coder launch done, download queued")
    if not root:
        torch.cuda.synchronize()
        return mgpu.broadcast_object(None)            # the summary string of rank 0

    bit_d = {"feat": {}, "scaling": {}, "offsets": {}}
    min_d = {"feat": {}, "scaling": {}, "offsets": {}}
    max_d = {"feat": {}, "scaling": {}, "offsets": {}}
    for gi, ((name, level), (blob, lens, mn, mx)) in enumerate(zip(tags, coded)):
        if gi not in early_written:
            writes += codec.write_file(path(f"{name}{level}.b"), blob, ready=ready)       # :1235-1238
        if version == 2:      # arrays: thousands of blocks as Python ints cost the decoder's unpickling a garbage-collector pass
            bit_d[name][level], min_d[name][level], max_d[name][level] = lens * 8, mn.astype(np.int32), mx.astype(np.int32)
        else:
            bit_d[name][level] = (lens * 8).tolist()
            min_d[name][level] = mn.astype(np.int64).tolist()
            max_d[name][level] = mx.astype(np.int64).tolist()

    tr("file writes submitted")
    torch.cuda.synchronize(); t_codec = time.time() - t0
    tr("bitstream on the host")
    if hyper_v2 is not None:
        bit_hyper_list = hyper_v2[1] * 8
        writes += codec.write_file(path("hyper.b"), hyper_v2[0])
    else:
        hyper_bytes = [j.result() for j in hyper_jobs]                                   # :1082-1098
        tr("hyper rANS jobs done")
        bit_hyper_list = [len(b) * 8 for b in hyper_bytes]
        with open(path("hyper.b"), "wb") as f:
            f.write(b"".join(hyper_bytes))
    bit_anchor = _anchor.numel() * 16
    bit_hyper = int(np.sum(bit_hyper_list))
    bit_feat = sum(int(np.sum(v)) for v in bit_d["feat"].values())
    bit_scaling = sum(int(np.sum(v)) for v in bit_d["scaling"].values())
    bit_offsets = sum(int(np.sum(v)) for v in bit_d["offsets"].values())

    if version == 2:
        mask_blob, mask_lens = mask_job.result()
        mask_bytes = mask_blob.tobytes()
    else:
        mask_bytes = mask_job.result()
    tr("mask stream done")
    with open(path("masks.b"), "wb") as f:
        f.write(mask_bytes)
    bit_masks = len(mask_bytes) * 8
    for w in writes:
        w.result()                    # every file is on disk before the encoder reports (and raises here if one failed)
    tr("files written")

    torch.cuda.synchronize(); t2 = time.time()
    print("encoding time:", t2 - t1)
    print("codec time:", t_codec)

    meta_path = path("meta.b")                                                            # :1276-1277
    meta = [pc._anchor.shape[0], MAX_BATCH, min_d["feat"], max_d["feat"], min_d["scaling"], max_d["scaling"],
            min_d["offsets"], max_d["offsets"], prob_masks, bit_hyper_list, bit_d["feat"], bit_d["scaling"],
            bit_d["offsets"], N_levels_list]
    if version == 2:
        meta.append({"version": 2, "block_symbols": V2_BLOCK, "block_policy": V2_BLOCK_POLICY, "hyper_block": V2_HYPER_BLOCK, "chunk": chunk,
                     "bit_masks": mask_lens * 8})
    torch.save(meta, meta_path)
    bit_meta = os.path.getsize(meta_path) * 8
    mlp = pc.get_mlp_size()[0]
    r = lambda v: round(v / bit2MB_scale, 4)
    return mgpu.broadcast_object(
            f"\nEncoded sizes in MB: meta {r(bit_meta)}, hyper {r(bit_hyper)}, anchor {r(bit_anchor)}, "
            f"feat {r(bit_feat)}, scaling {r(bit_scaling)}, offsets {r(bit_offsets)}, masks {r(bit_masks)}, "
            f"MLPs {r(mlp)}, Total {r(bit_meta + bit_hyper + bit_anchor + bit_feat + bit_scaling + bit_offsets + bit_masks + mlp)}, "
            f"EncTime {round(t2 - t1, 4)}")


@torch.no_grad()
def conduct_decoding(pc, pre_path_name):                       # :1299-1539
    torch.cuda.synchronize(); t1 = time.time()
    tr = _tracer("decode")
    print("Start decoding ...")
    path = lambda name: os.path.join(pre_path_name, name)
    # anchor.npy (12 MB at 1 M anchors) is read and converted on a host thread while this one loads the header, the MLPs and
    # the prior tables: the level plan — the first thing the device chain waits for — needs nothing else from the files
    def read_anchors():
        a = np.load(path("anchor.npy"))
        pinned = codec._pinned_staging(a.size * 4, "anchors")[:a.size * 4].view(torch.int32).view(a.shape)
        np.copyto(pinned.numpy(), a, casting="unsafe")
        return pinned
    anchor_job = codec.host_pool().submit(read_anchors)
    meta = read_mlp_checkpoint(path("meta.b"))        # (the same loader: nested lists / dicts of numbers, arrays, tensors)
    tr("meta.b unpickled")
    (N_full, max_batch, min_feat_d, max_feat_d, min_scaling_d, max_scaling_d, min_offsets_d, max_offsets_d, prob_masks,
     bit_hyper_list, bit_feat_d, bit_scaling_d, bit_offsets_d, N_levels_list) = meta[:14]
    extra = meta[14] if len(meta) > 14 else {"version": 1}
    version = int(extra.get("version", 1))
    if version not in (1, 2):
        raise RuntimeError(f"meta.b: container version {version} is newer than this decoder (1, 2)")
    # all coded streams: file -> pinned buffer -> device by the staging threads on a side stream, in the order the coder
    # launches consume them; started before anything else is loaded (reading ~120 MB is the longest chain of the prologue)
    dev = pc.x_bound_min.device
    n_lv = len(N_levels_list)
    if version == 2:       # masks first (their launch runs beside the prologue), then level by level incl. the offsets
        order = ["masks.b", "hyper.b"] + [f"{a}{l}.b" for l in reversed(range(n_lv)) for a in ("feat", "scaling", "offsets")]
    else:
        order = [f"{a}{l}.b" for l in reversed(range(n_lv)) for a in ("feat", "scaling")] + \
                [f"offsets{l}.b" for l in reversed(range(n_lv))]
    # (only masks.b starts now: the checkpoint is read first, on a quiet interpreter — beside eight reader threads that read
    #  took 4-8 ms instead of 0.3; the other files are released right after it and are still early for the first level)
    staged = codec.StagedFiles([path(f_) for f_ in order if os.path.exists(path(f_))], dev, start=1 if version == 2 else 0)
    codec.decode_status(dev, reset=True)     # the lane decoders report a malformed block here (read once, after the last launch)
    tr("file staging prepared")
    chunk = extra["chunk"] if version == 2 else {"masks": max_batch}
    lanes = version == 2
    block = int(extra.get("block_symbols", V2_BLOCK))
    policy = int(extra.get("block_policy", 0))
    block_of = lambda n_level, n_group: _level_blocks(n_level, n_group, D, policy, block)
    # (map_location: whatever tensors a header holds — the reference stores its minima / maxima as device tensors — are only
    #  ever read as Python numbers here: restoring them on the device would cost a copy each and a stream drain per .item())
    tr("meta.b loaded")
    K, D, H = pc.n_offsets, pc.feat_dim, pc.feat_dim // pc.hyper_divisor
    N_levels_list = list(reversed(N_levels_list))
    N_valid = sum(N_levels_list)
    # the mask stream (:1348-1353) is serial and only the offsets need it: decode it on a host thread meanwhile
    # (version 2: chunk streams, decoded by one device launch as soon as masks.b is staged — see below)
    mask_job = None
    if version == 1:
        mask_job = codec.host_pool().submit(codec.bernoulli_decode_host, np.fromfile(path("masks.b"), dtype=np.uint8),
                                            N_valid * K, float(prob_masks))
        tr("mask job submitted")
    # mlp.pt gates everything below (bounds -> anchors -> level plan; prior tables -> hyper launch) and unpickling it is the
    # longest host step of the prologue (on a host thread it only moved the time: unpickling holds the GIL).  The mask launch
    # needs the header and masks.b only — staged by the time the checkpoint is unpickled — so it is queued right after,
    # and its ~3 ms run beside the rest of the prologue instead of in front of the first level.
    ck = read_mlp_checkpoint(path("mlp.pt"))
    tr("mlp.pt unpickled")
    staged.release()
    side_stream = _side_stream(dev)
    masks_decoded, masks_ready = None, None
    if version == 2:
        with torch.cuda.stream(side_stream):
            mask_edges = torch.tensor(_chunk_rows(N_valid, chunk["masks"]), dtype=torch.int64) * K
            mask_lens = np.asarray(extra["bit_masks"], dtype=np.int64) // 8
            blob = staged.get(path("masks.b"))
            assert int(mask_lens.sum()) == int(blob.numel())
            masks_decoded = codec.bernoulli_decode_packed(float(prob_masks), mask_edges, blob, mask_lens).view(-1, K, 1)
            masks_ready = side_stream.record_event()
        tr("mask chunk streams: device launch enqueued")
    load_mlp_checkpoints(pc, path("mlp.pt"), ck=ck, tr=tr, stored_tables=version == 2)     # (incl. the hyper prior's CDF tables)
    # the level plan (sorts and compactions: milliseconds of device work) needs the anchors and the checkpoint's bounds only:
    # queued now, it runs while the host goes on with the mask / hyper launches
    q = anchor_job.result().to(dev, non_blocking=True)           # :1340-1342 (pinned: the copy is queued, not waited for)
    interval = (pc.x_bound_max - pc.x_bound_min) * Q_anchor + 1e-6
    anchor_decoded = q * interval + pc.x_bound_min
    tr("anchors on the device")
    if pc.level_scale is None:
        pc.level_scale = find_divide_scale(pc, anchor_decoded, pc.target_ratio, pc.level_num)
    plan, inverse_indices_list, mapping_list = level_plan(pc, anchor_decoded, None)
    tr("level plan built")
    dev = pc.x_bound_min.device
    if version == 2:
        # version 2: lane-parallel table blocks, one device launch (EntropyBottleneck.decompress_lanes_rows)
        hyper_rows = pc.latent_codec.decompress_lanes_rows(staged.get(path("hyper.b")), np.asarray(bit_hyper_list, dtype=np.int64) // 8,
                                                           N_valid, int(extra.get("hyper_block", V2_HYPER_BLOCK)))
        hyper_job = lambda: hyper_rows
        tr("hyper blocks: device launch enqueued")
    else:
        # the hyper strings are decoded by host threads (straight into a pinned [N_valid, H] buffer) while this thread stages the
        # files, loads the anchors and builds the level plan; submitted FIRST: the first level's prediction waits for them
        with open(path("hyper.b"), "rb") as f:
            hyper_stream = f.read()
        pos, strings, sizes = 0, [], []
        for s, s0 in enumerate(range(0, N_valid, max_batch * 10)):                           # :1326-1336 (any N_valid, Q2)
            nb = bit_hyper_list[s] // 8
            strings.append(hyper_stream[pos:pos + nb])
            sizes.append(min(max_batch * 10, N_valid - s0))
            pos += nb
        hyper_job = pc.latent_codec.decompress_chunks_rows(strings, sizes)
        tr("hyper rANS jobs submitted")

    hyper_decoded = hyper_job()                                                          # [N_valid, H]
    tr("hyper latents decoded (host rANS) and on the device")

    feat_after_Q = torch.zeros(N_valid, D, device=dev)
    grid_scaling_after_Q = torch.zeros(N_valid, 6, device=dev)
    grid_offset_after_Q = torch.zeros(N_valid, K, 3, device=dev)
    already_coded = torch.zeros(N_valid, dtype=torch.bool, device=dev)
    content_pre_gathered = None

    def chunk_lens(name, level, bit_list):
        blob = staged.get(path(f"{name}{level}.b"))                                      # device bytes
        lens = np.asarray(bit_list, dtype=np.int64) // 8
        assert int(lens.sum()) == int(blob.numel())                                      # :1479-1481
        return blob, lens

    # Coder launches last as long as their LONGEST stream (a 1000-anchor feature chunk: 50 000 serial symbols at ~0.4 us each) however
    # many streams they hold, so the fewer the better: one per level for features + scaling — the offsets of a level
    # need nothing but that level's prediction and the masks, so ALL of them go into ONE more launch, beside the last
    # level's (see below): 3 serial launches for 3 levels.
    pending_offsets = []

    def live_slots(orig_, n_, rows_):
        m30 = masks_decoded[orig_].repeat(1, 1, 3).reshape(n_, 3 * K).to(torch.bool)
        live = torch.nonzero(m30.reshape(-1))[:, 0]              # ONE compaction index for the three operands and the fill
        if rows_ is None:                                        # version 2: blocks of `block` live symbols
            return _block_edges(int(live.numel()), block_of(n_, int(live.numel()))), live
        cnt = torch.zeros(n_ + 1, dtype=torch.int64, device=dev)
        cnt[1:] = torch.cumsum(m30.sum(1), 0)
        return cnt[rows_.to(dev)], live

    live_pre = {}
    if version == 2:
        # the offsets' stream edges of EVERY level in one host read, before the first coder launch is queued (a read inside
        # the level loop would drain the previous level's launch and leave the device idle while the next one is built)
        torch.cuda.current_stream().wait_event(masks_ready)
        masks_decoded.record_stream(torch.cuda.current_stream())
        for (level_, _tc, orig_, _h) in plan:
            live_pre[level_] = live_slots(orig_, int(orig_.shape[0]), None)
        tr("offset blocks of all levels on the host")

    def offset_groups(pending):
        groups, fills = [], []
        for (level_, orig_, n_, rows_, mean_o, scale_o, Qo_) in pending:
            if level_ in live_pre:
                off_edges, live = live_pre[level_]
            else:
                off_edges, live = live_slots(orig_, n_, rows_)
                off_edges = off_edges.cpu()
            groups.append((mean_o.reshape(-1).index_select(0, live), scale_o.reshape(-1).index_select(0, live),
                           Qo_.index_select(0, live // (3 * K)), off_edges,
                           min_offsets_d[level_], max_offsets_d[level_],
                           *chunk_lens("offsets", level_, bit_offsets_d[level_]), 1))
            fills.append((orig_, n_, live))
        return groups, fills

    last_level = plan[-1][0] if plan else None
    for (level, to_code, orig, hybrid_anchor) in plan:
        n_l = int(orig.shape[0])
        assert n_l == N_levels_list[level]
        if content_pre_gathered is None:
            feat_in = torch.cat([anchor_decoded[orig], hyper_decoded[orig].float()], dim=1)
        else:
            feat_in = torch.cat([content_pre_gathered, hyper_decoded[orig]], dim=1)
        tr(f"level {level}: input rows assembled")
        (mean_feat, scale_feat, mean_scaling, scale_scaling, mean_offsets, scale_offsets, Qf, Qs, Qo) = \
            _predict(pc, level, feat_in)
        tr(f"level {level}: predicted (enqueued)")
        rows = torch.tensor(_chunk_rows(n_l, max_batch), dtype=torch.int64)
        edges_f, edges_s = ((_block_edges(n_l * D, block_of(n_l, n_l * D)), _block_edges(n_l * 6, block_of(n_l, n_l * 6))) if lanes
                            else (rows * D, rows * 6))
        pending_offsets.append((level, orig, n_l, rows, mean_offsets, scale_offsets, Qo))
        groups = [(mean_feat, scale_feat, Qf, edges_f, min_feat_d[level], max_feat_d[level],
                   *chunk_lens("feat", level, bit_feat_d[level]), D),
                  (mean_scaling, scale_scaling, Qs, edges_s, min_scaling_d[level], max_scaling_d[level],
                   *chunk_lens("scaling", level, bit_scaling_d[level]), 6)]
        fills = []
        if version == 2:
            # version 2: the masks are on the device long before the first level is predicted, so a level's offsets ride in
            # that level's launch (their streams are not longer than the feature streams beside them)
            og, fills = offset_groups(pending_offsets[-1:])
            groups += og
        elif level == last_level:
            fork = torch.cuda.current_stream().record_event()       # everything the offsets need exists before this point
        decoded = codec.gaussian_decode_groups(groups, lanes=lanes)
        tr(f"level {level}: coder launch enqueued")
        feat_dec, scal_dec = decoded[0], decoded[1]
        for (orig_, n_, live), off_vals in zip(fills, decoded[2:]):
            off_dec = torch.zeros(n_ * 3 * K, device=dev)
            off_dec.index_copy_(0, live, off_vals)
            grid_offset_after_Q[orig_] = off_dec.view(n_, K, 3)
        if version == 1 and level == last_level:
            # The offsets of ALL levels as their own launch on a side stream, forked BEFORE the last feature launch:
            # they need the mask stream (a serial host job of ~70 ms that ends about now), the features do not, and a
            # coder launch keeps few SIMDs busy (one wave per stream), so the two launches run side by side instead of
            # the last one waiting for the masks.
            main = torch.cuda.current_stream()
            with torch.cuda.stream(side_stream):
                side_stream.wait_event(fork)
                tr("waiting for the mask stream")
                masks_decoded = torch.from_numpy(mask_job.result()).to(dev).to(torch.float32).view(-1, K, 1)
                tr("mask stream decoded")
                og, fills = offset_groups(pending_offsets)
                for (orig_, n_, live), off_vals in zip(fills, codec.gaussian_decode_groups(og)):
                    off_dec = torch.zeros(n_ * 3 * K, device=dev)
                    off_dec.index_copy_(0, live, off_vals)
                    grid_offset_after_Q[orig_] = off_dec.view(n_, K, 3)
                tr("offsets launch enqueued")
            main.wait_stream(side_stream)

        tr(f"level {level}: offsets placed")
        feat_after_Q[orig] = feat_dec.view(n_l, D)
        grid_scaling_after_Q[orig] = scal_dec.view(n_l, 6)
        tr(f"level {level}: decoded rows placed")
        if level != 0:
            already_coded.index_fill_(0, orig, True)   # (x[idx] = True uploads its scalar through a blocking pageable copy)
            content_pre_gathered = context_rows(pc, anchor_decoded, feat_after_Q, grid_scaling_after_Q, already_coded,
                                                inverse_indices_list, mapping_list, level)
            tr(f"level {level}: context of the next level gathered")
    if masks_decoded is None:                    # no level at all (empty model)
        masks_decoded = torch.from_numpy(mask_job.result()).to(dev).to(torch.float32).view(-1, K, 1)
    if masks_ready is not None:
        torch.cuda.current_stream().wait_event(masks_ready)
        masks_decoded.record_stream(torch.cuda.current_stream())
    tr("levels enqueued")
    torch.cuda.synchronize(); t2 = time.time()
    tr("device done")
    if version == 2:
        codec.decode_status_check(dev)           # (after the synchronize: no extra wait)
    print("decoding time:", t2 - t1)

    z = lambda *s: torch.zeros(*s, device=dev)                                            # :1503-1533
    _hyper, _anchor, _feat = z(N_full, H), z(N_full, 3), z(N_full, D)
    _offset, _scaling, _mask = z(N_full, K, 3), z(N_full, 6), z(N_full, K, 1)
    _anchor[:N_valid] = anchor_decoded
    _hyper[:N_valid] = hyper_decoded
    _feat[:N_valid] = feat_after_Q
    _offset[:N_valid] = grid_offset_after_Q
    _scaling[:N_valid] = grid_scaling_after_Q
    _mask[:N_valid] = masks_decoded
    pc._hyper_latent = nn.Parameter(_hyper)
    pc._anchor_feat = nn.Parameter(_feat)
    pc._offset = nn.Parameter(_offset)
    pc.decoded_version = True
    pc._anchor = nn.Parameter(_anchor)
    pc._scaling = nn.Parameter(_scaling)
    pc._mask = nn.Parameter(_mask)
    if hasattr(pc, "_level_cache"):
        pc._level_cache = None
        pc._anchor_q_cache = None
    pc._anchor_q_cache = None
    return f"\nDecTime {round(t2 - t1, 4)}"
