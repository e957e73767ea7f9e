"""PLY files of the reference: `save_ply` / `load_ply_sparse_gaussian`
(scene/gaussian_model.py:561-654) and the input clouds of `create_from_pcd` (scene/dataset_readers.py fetchPly /
storePly).  The reference goes through the `plyfile` wheel (not in the mount, not installed); this is the same
on-disk format — header `ply / format binary_little_endian 1.0 / element vertex N / property <type> <name> ... /
end_header` followed by N packed little-endian records — written and parsed with numpy structured arrays.
Scalar properties only (what both files use); ascii and big-endian bodies are read too."""
from __future__ import annotations

import numpy as np

# PLY scalar type names (both spellings) -> numpy codes
_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
          "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
          "double": "f8", "float64": "f8"}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_ply(path: str, vertex: np.ndarray, comments=()) -> None:
    """vertex: 1-D structured array; written as element `vertex`, binary little endian (plyfile's default)."""
    assert vertex.dtype.names, "structured array expected"
    lines = ["ply", "format binary_little_endian 1.0"] + [f"comment {c}" for c in comments]
    lines.append(f"element vertex {vertex.shape[0]}")
    fields = []
    for name in vertex.dtype.names:
        code = vertex.dtype[name].str[1:]                     # strip the byte-order character
        if code not in _NAMES:
            raise ValueError(f"property {name}: unsupported dtype {vertex.dtype[name]}")
        lines.append(f"property {_NAMES[code]} {name}")
        fields.append((name, "<" + code))
    lines.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(vertex.astype(np.dtype(fields), copy=False)).tobytes())


def read_ply(path: str, element: str = "vertex") -> np.ndarray:
    """-> structured array of the named element (scalar properties only)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header has no end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][2].append(None)              # list property: only skippable if never reached
                else:
                    elements[-1][2].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        order = {"binary_little_endian": "<", "binary_big_endian": ">", "ascii": "="}.get(fmt)
        if order is None:
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")
        for name, count, props in elements:
            if any(p is None for p in props):
                raise NotImplementedError(f"{path}: list properties (element {name}) are not supported")
            dt = np.dtype([(n, (order if fmt != "ascii" else "") + c) for n, c in props])
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                arr = np.empty(count, dtype=dt)
                for j, (n, _) in enumerate(props):
                    arr[n] = np.array([r[j] for r in rows], dtype=np.float64).astype(dt[n]) if count else []
            else:
                arr = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
            if name == element:
                return arr.astype(dt.newbyteorder("="), copy=False)
    raise KeyError(f"{path}: no element {element!r}")


# ---- the model's ply (scene/gaussian_model.py:561-654) -------------------------------------------------------
def model_attribute_names(n_offsets: int, feat_dim: int, hyper_dim: int, n_scaling: int = 6, n_rot: int = 4) -> list[str]:
    """construct_list_of_attributes (:561-577)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_offset_{i}" for i in range(n_offsets * 3)]
    names += [f"f_mask_{i}" for i in range(n_offsets)]
    names += [f"f_anchor_feat_{i}" for i in range(feat_dim)]
    names += [f"f_hyper_latent_{i}" for i in range(hyper_dim)]
    names += ["opacity"] + [f"scale_{i}" for i in range(n_scaling)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_model_ply(path: str, anchor, offset, mask, feat, hyper, opacity, scaling, rotation) -> None:
    """save_ply (:579-598): offsets [N,K,3] and masks [N,K,1] are stored TRANSPOSED (transpose(1,2).flatten(1):
    all x offsets, then all y, then all z), normals are zeros, every property is float32."""
    c = lambda t: t.detach().cpu().numpy().astype(np.float32)
    anchor, feat, hyper, opacity, scaling, rotation = map(c, (anchor, feat, hyper, opacity, scaling, rotation))
    off = c(offset.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    msk = c(mask.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    table = np.concatenate((anchor, np.zeros_like(anchor), off, msk, feat, hyper, opacity, scaling, rotation), axis=1)
    names = model_attribute_names(offset.shape[1], feat.shape[1], hyper.shape[1], scaling.shape[1], rotation.shape[1])
    assert table.shape[1] == len(names)
    vertex = np.ascontiguousarray(table, dtype="<f4").view([(n, "<f4") for n in names])[:, 0]     # one record per row
    write_ply(path, vertex)


def load_model_ply(path: str) -> dict:
    """load_ply_sparse_gaussian (:600-654) -> dict of float32 numpy arrays with the model's shapes
    (offset [N,K,3], mask [N,K,1]: the stored [N,3,K] / [N,1,K] layouts transposed back)."""
    v = read_ply(path)

    def group(prefix):
        names = sorted((n for n in v.dtype.names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
        return np.stack([v[n].astype(np.float32) for n in names], axis=1) if names else np.zeros((v.shape[0], 0), np.float32)
    n = v.shape[0]
    return {"anchor": np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32),
            "opacity": v["opacity"].astype(np.float32)[:, None],
            "scaling": group("scale_"), "rotation": group("rot"), "feat": group("f_anchor_feat"),
            "hyper": group("f_hyper_latent"),
            "offset": np.ascontiguousarray(group("f_offset").reshape(n, 3, -1).transpose(0, 2, 1)),
            "mask": np.ascontiguousarray(group("f_mask").reshape(n, 1, -1).transpose(0, 2, 1))}
