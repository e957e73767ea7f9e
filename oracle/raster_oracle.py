"""numpy/ctypes front-end of the CPU rasterizer oracle (oracle/raster_ref.c).

TEST INFRASTRUCTURE ONLY (see the header of raster_ref.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
contextgs_amd/.  PARITY UNPINNED for the rasterizer — the reference's CUDA
extension is not in the mount.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "build")


def build(force: bool = False) -> None:
    """Compile the oracle libraries with gcc (oracle/Makefile)."""
    targets = ["libraster_ref_f32.so", "libraster_ref_f64.so"]
    if not force and all(os.path.exists(os.path.join(BUILD, t)) for t in targets):
        srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".c")]
        newest_src = max(os.path.getmtime(s) for s in srcs)
        oldest_lib = min(os.path.getmtime(os.path.join(BUILD, t)) for t in targets)
        if oldest_lib >= newest_src:
            return
    r = subprocess.run(["make", "-C", HERE, "-B", "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)


def _cfg_type(real):
    class RefCfg(C.Structure):
        _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("tanfovx", real), ("tanfovy", real),
                    ("scale_modifier", real), ("view", real * 16), ("proj", real * 16), ("bg", real * 3)]
    return RefCfg


class RasterOracle:
    def __init__(self, dtype=np.float32):
        build()
        self.dtype = np.dtype(dtype)
        name = "libraster_ref_f32.so" if self.dtype == np.float32 else "libraster_ref_f64.so"
        self.real = C.c_float if self.dtype == np.float32 else C.c_double
        self.lib = C.CDLL(os.path.join(BUILD, name))
        assert self.lib.ref_sizeof_real() == self.dtype.itemsize
        self.Cfg = _cfg_type(self.real)
        self.lib.ref_filter.restype = C.c_int
        self.lib.ref_render.restype = C.c_int

    def _cfg(self, H, W, tanfovx, tanfovy, view, proj, bg, scale_modifier=1.0):
        c = self.Cfg()
        c.H, c.W = int(H), int(W)
        c.tanfovx, c.tanfovy, c.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        v = np.asarray(view, dtype=self.dtype).reshape(16)
        p = np.asarray(proj, dtype=self.dtype).reshape(16)
        b = np.asarray(bg, dtype=self.dtype).reshape(3)
        for i in range(16):
            c.view[i] = float(v[i])
            c.proj[i] = float(p[i])
        for i in range(3):
            c.bg[i] = float(b[i])
        return c

    def _arr(self, a, shape=None):
        a = np.ascontiguousarray(np.asarray(a, dtype=self.dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)

    def visible_filter(self, cam, means3D, scales, rots):
        """radii int32[N] (reference: gaussian_renderer/__init__.py:280-285)."""
        cfg = self._cfg(**cam)
        m, s, r = self._arr(means3D), self._arr(scales), self._arr(rots)
        N = m.shape[0]
        radii = np.zeros(N, dtype=np.int32)
        rc = self.lib.ref_filter(C.byref(cfg), C.c_int64(N), self._p(m), self._p(s), self._p(r), self._p(radii))
        assert rc == 0
        return radii

    def render(self, cam, means3D, colors, opacities, scales, rots, dL_dout=None):
        """Forward (and backward if dL_dout is given).  Returns a dict."""
        cfg = self._cfg(**cam)
        H, W = cfg.H, cfg.W
        m, col = self._arr(means3D), self._arr(colors)
        op = self._arr(opacities).reshape(-1)
        s, r = self._arr(scales), self._arr(rots)
        P = m.shape[0]
        out = np.zeros((3, H, W), dtype=self.dtype)
        radii = np.zeros(P, dtype=np.int32)
        final_T = np.zeros((H, W), dtype=self.dtype)
        wsum = np.zeros((H, W), dtype=self.dtype)
        stats = np.zeros(2, dtype=np.int64)
        res = {"color": out, "radii": radii, "final_T": final_T, "weight_sum": wsum, "stats": stats}
        if dL_dout is None:
            g = None
            grads = [None] * 6
        else:
            g = self._arr(dL_dout).reshape(3, H, W)
            grads = [np.zeros((P, 3), self.dtype), np.zeros((P, 3), self.dtype), np.zeros((P, 3), self.dtype),
                     np.zeros((P,), self.dtype), np.zeros((P, 3), self.dtype), np.zeros((P, 4), self.dtype)]
            for k, a in zip(["dL_dmeans3D", "dL_dmeans2D", "dL_dcolors", "dL_dopacities", "dL_dscales",
                             "dL_drotations"], grads):
                res[k] = a
        rc = self.lib.ref_render(C.byref(cfg), C.c_int64(P), self._p(m), self._p(col), self._p(op), self._p(s),
                                 self._p(r), self._p(out), self._p(radii), self._p(final_T), self._p(wsum),
                                 self._p(g), *[self._p(a) for a in grads], self._p(stats))
        assert rc == 0
        return res
