"""Shim for the `plyfile` wheel the reference imports (scene/gaussian_model.py:21, scene/dataset_readers.py): the
subset it uses — PlyData.read / PlyData([el]).write, PlyElement.describe, element["name"], element.properties[i].name,
plydata["vertex"], plydata.elements[0] — over contextgs_amd.ply_io (same on-disk format, scalar properties)."""
from contextgs_amd import ply_io as _io


class PlyProperty:
    def __init__(self, name, dtype):
        self.name, self.val_dtype = name, dtype


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data

    @staticmethod
    def describe(data, name, **_):
        return PlyElement(name, data)

    @property
    def properties(self):
        return tuple(PlyProperty(n, self.data.dtype[n].str) for n in self.data.dtype.names)

    @property
    def count(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.data.shape[0]


class PlyData:
    def __init__(self, elements=(), text=False, byte_order="<", comments=()):
        if text:
            raise NotImplementedError("ascii output is not implemented (the reference writes binary)")
        self.elements, self.comments = list(elements), list(comments)

    @staticmethod
    def read(stream):
        return PlyData([PlyElement("vertex", _io.read_ply(stream))])

    def write(self, stream):
        (el,) = self.elements
        _io.write_ply(stream, el.data, self.comments)

    def __getitem__(self, name):
        for el in self.elements:
            if el.name == name:
                return el
        raise KeyError(name)
