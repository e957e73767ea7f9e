"""Byte-level digest of a container written by conduct_encoding (a known-answer pin of the
bitstream).  Shared by tools/make_container_golden.py (which writes tests/golden/container_n*.json on the MI355X) and
tests/test_codec_gpu.py::test_container_bytes_match_the_committed_digest (which re-encodes and compares).

The reference's only codec check is a consumption assert (scene/gaussian_model.py:1479-1481); a self-consistent
change of the CDF arithmetic, the chunk order or the coder would leave every round-trip test green and silently orphan
every file written by an earlier build.  This pins sha256 + length of every coded file, the header's content, and
the decoded tensors, for the golden models (tests/golden_inputs.py: N = 3000 seed 2, N = 10000 seed 3).
mlp.pt and the pickled meta.b are pinned by CONTENT (tensor bytes / a canonical JSON of the header list), not by their
file bytes: those are torch.save archives whose byte layout belongs to the installed torch.
"""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np
import torch

CODED = ["anchor.npy", "hyper.b", "masks.b"] + [f"{a}{l}.b" for a in ("feat", "scaling", "offsets") for l in range(3)]


def _sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def build_model(N, seed):
    import golden_inputs as gi
    from contextgs_amd.model import GaussianModel
    pc = GaussianModel(voxel_size=0.01)
    sd = pc.state_dict()
    for k, v in gi.mlp_weights(seed).items():
        sd[k] = torch.from_numpy(v).cuda()
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(N, seed)
    pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
    pc.update_anchor_bound()
    pc.eval()
    return pc


def _canon(o):
    if isinstance(o, dict):
        return {str(k): _canon(v) for k, v in sorted(o.items(), key=lambda kv: str(kv[0]))}
    if isinstance(o, (list, tuple)):
        return [_canon(v) for v in o]
    if isinstance(o, torch.Tensor) or isinstance(o, np.ndarray):
        return _canon(o.tolist())
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (float, np.floating)):
        return float(np.float64(o)).hex()
    return o


def container_digest(N, seed, version, workdir):
    """Encode the golden model (N, seed) as container `version` into workdir, decode it into a second model, and return
    {"files": {name: [sha256, bytes]}, "meta": sha256 of the canonical header, "decoded": {tensor: sha256}}."""
    from contextgs_amd.codec_driver import conduct_encoding
    enc = build_model(N, seed)
    d = os.path.join(str(workdir), f"c_{N}_{seed}_v{version}")
    conduct_encoding(enc, d, container_version=version)
    files = {}
    for f in CODED:
        b = open(os.path.join(d, f), "rb").read()
        files[f] = [_sha(b), len(b)]
    meta = torch.load(os.path.join(d, "meta.b"), map_location="cpu", weights_only=False)
    dec = build_model(N, seed)
    with torch.no_grad():
        dec._anchor_feat.zero_(); dec._offset.zero_(); dec._hyper_latent.zero_(); dec._scaling.zero_()
    dec.conduct_decoding(d)
    decoded = {}
    for name in ("_anchor", "_anchor_feat", "_offset", "_scaling", "_mask", "_hyper_latent"):
        decoded[name] = _sha(getattr(dec, name).detach().cpu().contiguous().numpy().tobytes())
    return {"N": N, "seed": seed, "container_version": version, "files": files,
            "meta": _sha(json.dumps(_canon(meta)).encode()), "decoded": decoded}
