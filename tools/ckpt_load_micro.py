"""How long the decoder's two header reads take on the GPU box: codec_driver._fast_checkpoint_load vs torch.load (mapped / plain),
alone and while the file-staging threads of a decode are running (GIL / mmap-lock contention)."""
import os, sys, tempfile, shutil, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import codec, codec_driver as cd
from contextgs_amd.synth import make_scene

pc = make_scene(1_000_000, seed=0, requires_grad=False); pc.eval()
d = tempfile.mkdtemp(prefix="cgs_ck_")
try:
    cd.conduct_encoding(pc, d, container_version=2)
    p = os.path.join(d, "mlp.pt")
    loaders = {"fast": lambda: cd._fast_checkpoint_load(p),
               "torch mmap": lambda: torch.load(p, map_location="cpu", weights_only=False, mmap=True),
               "torch plain": lambda: torch.load(p, map_location="cpu", weights_only=False)}
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".b") and f != "meta.b"]
    for staged in (False, True):
        for name, fn in loaders.items():
            ts = []
            for _ in range(5):
                st = codec.StagedFiles(files, pc.x_bound_min.device) if staged else None
                t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
                if st is not None:
                    st.wait_all()
                torch.cuda.synchronize()
            print(f"staging threads {'running' if staged else 'idle   '}  {name:12s} " + " ".join(f"{x:6.2f}" for x in ts) + " ms")
finally:
    shutil.rmtree(d, ignore_errors=True)

# the decoder's own sequence: anchor thread, header, staging threads, checkpoint
import numpy as np
d = tempfile.mkdtemp(prefix="cgs_ck_")
try:
    cd.conduct_encoding(pc, d, container_version=2)
    files = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(".b") and f != "meta.b"]
    for with_anchor in (False, True):
        for rep in range(4):
            t0 = time.perf_counter()
            job = codec.host_pool().submit(lambda: np.load(os.path.join(d, "anchor.npy")).astype(np.int32)) if with_anchor else None
            cd.read_mlp_checkpoint(os.path.join(d, "meta.b")); t1 = time.perf_counter()
            st = codec.StagedFiles(files, pc.x_bound_min.device); t2 = time.perf_counter()
            try:
                cd._fast_checkpoint_load(os.path.join(d, "mlp.pt")); ok = "fast ok"
            except Exception as e:
                ok = "fast FAILED: " + str(e)[:80]
            t3 = time.perf_counter()
            cd.read_mlp_checkpoint(os.path.join(d, "mlp.pt")); t4 = time.perf_counter()
            if job is not None:
                job.result()
            st.wait_all(); torch.cuda.synchronize()
            print(f"sequence anchor_thread={with_anchor}: meta {1e3*(t1-t0):.2f} staging-submit {1e3*(t2-t1):.2f} mlp fast {1e3*(t3-t2):.2f} ({ok}) "
                  f"mlp wrapper again {1e3*(t4-t3):.2f} ms")
finally:
    shutil.rmtree(d, ignore_errors=True)
