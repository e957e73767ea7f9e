"""On-disk PLY formats of the reference (scene/gaussian_model.py:561-654): header text, packed
float32 records, the TRANSPOSED offset / mask layout, and the plyfile-compatible shim the reference's own
save_ply / load_ply_sparse_gaussian / fetchPly can run on."""
import os
import sys

import numpy as np
import torch

from contextgs_amd import ply_io


def _state(N=300, K=10, D=50, H=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    return dict(anchor=r(N, 3), offset=r(N, K, 3), mask=(r(N, K, 1) > 0).float(), feat=r(N, D), hyper=r(N, H),
                opacity=r(N, 1), scaling=r(N, 6), rotation=r(N, 4))


def test_model_ply_layout_and_roundtrip(tmp_path):
    st = _state()
    p = str(tmp_path / "point_cloud.ply")
    ply_io.save_model_ply(p, **st)
    raw = open(p, "rb").read()
    names = ply_io.model_attribute_names(10, 50, 12)
    assert names[:6] == ["x", "y", "z", "nx", "ny", "nz"] and names[6] == "f_offset_0" and names[36] == "f_mask_0"
    assert names[-11:] == ["opacity"] + [f"scale_{i}" for i in range(6)] + [f"rot_{i}" for i in range(4)]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 300\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    assert raw.startswith(header.encode()) and len(raw) == len(header) + 300 * 4 * len(names)
    body = np.frombuffer(raw[len(header):], dtype="<f4").reshape(300, len(names))
    assert np.array_equal(body[:, :3], st["anchor"].numpy()) and not body[:, 3:6].any()
    # transposed layout (:587-588): f_offset_{c*K + k} = offset[:, k, c]
    assert np.array_equal(body[:, 6 + 1 * 10 + 7], st["offset"][:, 7, 1].numpy())
    assert np.array_equal(body[:, 36 + 4], st["mask"][:, 4, 0].numpy())
    back = ply_io.load_model_ply(p)
    for k, v in st.items():
        assert back[k].dtype == np.float32 and np.array_equal(back[k], v.numpy()), k


def test_reader_handles_ascii_mixed_types_and_comments(tmp_path):
    p = str(tmp_path / "points3D.ply")
    open(p, "w").write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty float y\n"
                       "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n"
                       "0.5 1 -2 255 0 7\n1e-3 2 3 1 2 3\n")
    v = ply_io.read_ply(p)
    assert v["x"].tolist() == [0.5, np.float32(1e-3)] and v["blue"].tolist() == [7, 3] and v["red"].dtype == np.uint8
    # binary with mixed types, the layout storePly writes (scene/dataset_readers.py)
    dt = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")]
    a = np.zeros(5, dtype=dt)
    a["x"] = np.arange(5); a["blue"] = np.arange(5) + 9
    q = str(tmp_path / "in.ply")
    ply_io.write_ply(q, a, comments=["c"])
    b = ply_io.read_ply(q)
    assert b.dtype.names == a.dtype.names and np.array_equal(b["x"], a["x"]) and np.array_equal(b["blue"], a["blue"])


def test_plyfile_shim_runs_the_reference_style_calls(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "contextgs_amd", "dropin"))
    try:
        sys.modules.pop("plyfile", None)
        from plyfile import PlyData, PlyElement
        st = _state(N=50)
        names = ply_io.model_attribute_names(10, 50, 12)
        cols = [st["anchor"], torch.zeros(50, 3), st["offset"].transpose(1, 2).flatten(1), st["mask"].transpose(1, 2).flatten(1),
                st["feat"], st["hyper"], st["opacity"], st["scaling"], st["rotation"]]
        attributes = np.concatenate([c.numpy() for c in cols], axis=1)
        elements = np.empty(50, dtype=[(n, "f4") for n in names])
        elements[:] = list(map(tuple, attributes))                      # exactly what save_ply does (:593-597)
        p = str(tmp_path / "ref_style.ply")
        PlyData([PlyElement.describe(elements, "vertex")]).write(p)
        q = str(tmp_path / "ours.ply")
        ply_io.save_model_ply(q, **st)
        assert open(p, "rb").read() == open(q, "rb").read()
        plydata = PlyData.read(p)                                       # load_ply_sparse_gaussian's accesses (:601-640)
        assert np.array_equal(np.asarray(plydata.elements[0]["x"]), st["anchor"][:, 0].numpy())
        scale_names = sorted([pr.name for pr in plydata.elements[0].properties if pr.name.startswith("scale_")],
                             key=lambda x: int(x.split("_")[-1]))
        assert scale_names == [f"scale_{i}" for i in range(6)] and len(plydata["vertex"]) == 50
    finally:
        sys.path.pop(0)
        sys.modules.pop("plyfile", None)


def test_capture_restore_is_the_reference_checkpoint_tuple(tmp_path):
    """capture() / restore() (scene/gaussian_model.py:222-286): the 19-entry tuple in the reference's order survives
    torch.save / torch.load and rebuilds an identical model (tensors, MLPs, prior, bounds, level scale)."""
    import torch
    from contextgs_amd.model import GaussianModel
    torch.manual_seed(0)
    n, K, D = 40, 10, 50
    a = GaussianModel(device="cpu")
    a.set_state(torch.randn(n, 3), torch.randn(n, K, 3), torch.randn(n, K, 1), torch.randn(n, D), torch.randn(n, D // 4),
                torch.randn(n, 6))
    a.level_scale, a.spatial_lr_scale = 1.75, 3.0
    a.x_bound_min, a.x_bound_max = torch.full((1, 3), -2.0), torch.full((1, 3), 2.5)
    with torch.no_grad():
        for p in a.latent_codec.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    tup = a.capture()
    assert len(tup) == 19 and tup[9] is None               # slot 9 is the optimizer state (no driver attached here)
    torch.save(tup, tmp_path / "chkpnt.pth")
    b = GaussianModel(device="cpu").restore(torch.load(tmp_path / "chkpnt.pth", weights_only=False))
    for name in ("_anchor", "_anchor_feat", "_hyper_latent", "_offset", "_mask", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert getattr(a, name).requires_grad == getattr(b, name).requires_grad, name
    for ma, mb in ((a.mlp_opacity, b.mlp_opacity), (a.mlp_cov, b.mlp_cov), (a.mlp_color, b.mlp_color), (a.mlp_grid, b.mlp_grid),
                   (a.latent_codec, b.latent_codec)):
        for (ka, va), (kb, vb) in zip(ma.state_dict().items(), mb.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka
    assert b.level_scale == 1.75 and b.spatial_lr_scale == 3.0
    assert torch.equal(b.x_bound_min, a.x_bound_min) and torch.equal(b.x_bound_max, a.x_bound_max)


def test_attribute_names_equal_the_reference_list():
    """tests/golden/ply_names.json: the reference's construct_list_of_attributes() for the default shapes
    (generated by tools/make_goldens.py from the reference's own method)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ply_names.json")))
    assert ply_io.model_attribute_names(g["n_offsets"], g["feat_dim"], g["hyper_dim"]) == g["names"]
