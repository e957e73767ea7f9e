"""Write the bitstream known-answer fixtures tests/golden/container_n3000.json / container_n10000.json (run on the
MI355X: the coded bytes depend on the device's erf and on the HIP kernels, nothing here reads the reference).

    python tools/make_container_golden.py [outdir]        (default gpurun_out/golden; copy the files to tests/golden/)
"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import contextlib
    from container_digest import container_digest
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(out, exist_ok=True)
    for N, seed in ((3000, 2), (10000, 3)):
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(sys.stderr):
            runs = [[container_digest(N, seed, v, os.path.join(tmp, f"r{r}")) for v in (1, 2)] for r in range(2)]
        assert runs[0] == runs[1], "two encodes of the same model differ: the container is not deterministic"
        path = os.path.join(out, f"container_n{N}.json")
        with open(path, "w") as f:
            json.dump({"generator": "tools/make_container_golden.py (MI355X, gfx950)", "containers": runs[0]}, f, indent=1)
        print(path, {v["container_version"]: sum(b for (_h, b) in v["files"].values()) for v in runs[0]})


if __name__ == "__main__":
    main()
