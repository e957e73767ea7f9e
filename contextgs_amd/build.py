"""Build libcgs_hip.so (the gfx950 HIP library behind include/cgs.h) in-tree.

Plain `hipcc --offload-arch=gfx950` on each translation unit, then one shared
link.  hipcc cross-compiles without a GPU, so this also runs in the CPU-only
authoring container.  The result (contextgs_amd/libcgs_hip.so) is git-ignored
but travels with the gpurun snapshot.

Usage:  python -m contextgs_amd.build [--force] [--verbose]
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT = os.path.join(HERE, "libcgs_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")

ARCH = "gfx950"
# -ffp-contract=off: every fused multiply-add in the kernels is an explicit
# fmaf(), so forward and backward take bit-identical skip decisions and the
# encoder/decoder see bit-identical CDFs.
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}",
            "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]
# tools/ only: extra defines for timing experiments (e.g. CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS -DAG_ABL=1"); the flags are
# part of the build stamp, so an experiment build never masquerades as the product library
CXXFLAGS += os.environ.get("CGS_EXTRA_FLAGS", "").split()


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libcgs_hip.so cannot be built")
    return exe


def sources() -> list[str]:
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers() -> list[str]:
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]


def source_digest() -> str:
    """sha256 over the library's sources and headers (names relative to the package, so the digest is the same in the
    authoring container and on the GPU box); cgs_build_info() carries it."""
    h = hashlib.sha256()
    for p in sorted(sources() + _headers(), key=os.path.basename):
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())
            h.update(f.read())
    return h.hexdigest()


def _digest(paths: list[str]) -> str:
    return hashlib.sha256((source_digest() + " ".join(CXXFLAGS)).encode()).hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    srcs = sources()
    stamp = os.path.join(OBJ_DIR, "stamp")
    digest = _digest(srcs)
    src_digest = source_digest()
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == digest:
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        cmd = [cc, *CXXFLAGS, "-c", src, "-o", obj]
        if os.path.basename(src) == "api.hip":      # cgs_build_info()
            flags = " ".join(f for f in CXXFLAGS if f.startswith(("-O", "-D", "-ffp")))
            cmd += [f'-DCGS_SOURCE_DIGEST="{src_digest}"', f'-DCGS_BUILD_FLAGS="{flags}"']
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", OUT]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return OUT


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(path)
