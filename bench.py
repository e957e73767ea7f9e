#!/usr/bin/env python
"""bench.py — views/s of the ContextGS render-and-compress hot path on MI355X.

A "step" = one training view through the hot path, exactly what the reference does per
iteration (train.py:158-161,211): prefilter_voxel -> render(training) -> backward of a
fixed linear loss (sum(image * w), w seeded) -> (N>1) gradient all-reduce over RCCL.
Workload = BASELINE.json's metric config: 1 M-anchor synthetic scene, 1920x1080,
step-20000 semantics (context model on all anchors + rate + rasterizer).  The
raster-only phase (step <= 3000 semantics) is timed too and reported as
`value_raster_only`.  Inputs are resident in HBM before the timed region.

One rank per GPU (torchrun); views shard across ranks (weak scaling: one view per
rank per step), parameters replicated, one flattened-gradient all-reduce per step.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--anchors", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--step-semantics", type=int, default=20000, help="training iteration number passed to render()")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-raster-only", action="store_true")
    return ap.parse_args()


def flat_grads(params):
    import torch
    return [p.grad for p in params if p.grad is not None]


def one_step(pc, cam, pipe, bg, w, step_sem, params, dist_on):
    """prefilter -> render -> backward (-> all-reduce). Returns the render dict."""
    import torch
    from contextgs_amd.renderer import prefilter_voxel, render
    for p in params:
        p.grad = None
    vis = prefilter_voxel(cam, pc, pipe, bg)
    pkg = render(cam, pc, pipe, bg, visible_mask=vis, retain_grad=False, step=step_sem)
    loss = (pkg["render"] * w).sum()
    if pkg["bit_per_param"] is not None:
        loss = loss + 0.001 * pkg["bit_per_param"]          # lambda * rate term (train.py:206-209)
    loss.backward()
    if dist_on:
        from contextgs_amd.dist import allreduce_gradients
        allreduce_gradients(params, average=True)             # RCCL over xGMI: one flat bucket per step
    return pkg


def timed(fn, steps, dist_on):
    import torch
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def read_prof():
    from contextgs_amd import _lib
    L = _lib.lib()
    out = {}
    for i in range(L.cgs_prof_count()):
        ms, n = C.c_double(0), C.c_int64(0)
        L.cgs_prof_read(i, C.byref(ms), C.byref(n))
        if n.value:
            out[L.cgs_prof_name(i).decode()] = (ms.value, n.value)
    return out


def main():
    args = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
    from contextgs_amd import _lib
    from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
    L = _lib.lib()

    N, W, H = args.anchors, args.width, args.height
    pc = make_scene(N, seed=0)                  # identical replicas on every rank (same seed)
    pc.train()
    pipe = SynthPipe()
    bg = torch.zeros(3, device="cuda")
    n_views = max(args.views, world)
    cams = [c.to_torch("cuda") for c in orbit_cameras(n_views, W, H)]
    g = torch.Generator(device="cuda").manual_seed(1234)
    w = torch.randn(3, H, W, device="cuda", generator=g) / (H * W)
    params = [p for p in pc.parameters() if p.requires_grad]

    def cam_of(i):
        return cams[(i * world + rank) % n_views]

    full = lambda i: one_step(pc, cam_of(i), pipe, bg, w, args.step_semantics, params, dist_on)
    raster = lambda i: one_step(pc, cam_of(i), pipe, bg, w, 1000, params, dist_on)

    for i in range(args.warmup):
        full(i)
    L.cgs_prof_enable(1)
    dt = timed(full, args.steps, dist_on)
    prof = read_prof()
    L.cgs_prof_enable(0)
    views = args.steps * world
    value = views / dt
    ms_per_step = dt / args.steps * 1e3

    value_raster = None
    if not args.no_raster_only:
        for i in range(max(1, args.warmup // 2)):
            raster(i)
        dt_r = timed(raster, args.steps, dist_on)
        value_raster = views / dt_r

    # ---- workload statistics of one view (rank 0) for the algorithmic-byte accounting ----
    result = None
    if rank == 0:
        from contextgs_amd.rasterizer import raster_stats
        from contextgs_amd.renderer import _raster_settings, prefilter_voxel, render
        cam = cam_of(0)
        vis = prefilter_voxel(cam, pc, pipe, bg)
        pkg = render(cam, pc, pipe, bg, visible_mask=vis, step=1000)
        from contextgs_amd.rasterizer import last_call
        P = int(pkg["radii"].numel())
        R = int(last_call["num_rendered"])
        img_ws = last_call["img_ws"]
        st = raster_stats(_raster_settings(cam, pipe, bg, 1.0), img_ws).cpu().tolist()
        R_eff = int(st[0])
        n_vis = int(vis.sum())
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        tile_bits = max(1, (tiles - 1).bit_length())
        # algorithmic bytes per launch (SURVEY.md §8d / BASELINE.md §4; sort traffic is for OUR two-level
        # sort: 4 passes over P (key+value, histogram read + scatter read/write) and ceil(bits/8) over R)
        alg = {
            "blend_fwd": 40 * R_eff + 20 * H * W,
            "blend_bwd": 76 * R_eff + 20 * H * W,
            "preprocess": 128 * P,
            "preprocess_bwd": (40 + 20 + 52) * P,
            "filter": 44 * N,
            "depth_sort": 4 * 20 * P + 4 * P,
            "tile_sort": ((tile_bits + 7) // 8) * 20 * R,
            "emit_pairs": 20 * P + 8 * R,
            "offsets_scan": 20 * P,
            "ranges": 4 * R + 8 * tiles,
            "expand_fwd": (396 - 200) * n_vis + 56 * P,
            "expand_bwd": 2 * (396 - 200) * n_vis + 56 * P,
        }
        kernels = {}
        for name, (ms, n) in prof.items():
            avg_us = ms / n * 1e3
            k = {"avg_us": round(avg_us, 2), "launches": n, "total_ms": round(ms, 3)}
            if name in alg:
                k["alg_bytes"] = alg[name]
                k["GBps"] = round(alg[name] / (avg_us * 1e-6) / 1e9, 1)
            kernels[name] = k
        dom = max(kernels, key=lambda n_: kernels[n_]["total_ms"]) if kernels else None
        roofline = None
        if dom and "GBps" in kernels[dom]:
            roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(kernels[dom]["GBps"] / HBM_PEAK_GBS, 4), "traffic": None,
                        "alg_bytes_per_launch": kernels[dom]["alg_bytes"], "avg_launch_us": kernels[dom]["avg_us"]}
        lib_ms = sum(k["total_ms"] for k in kernels.values()) / max(1, args.steps)

        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(pc, cam, pipe, bg, w, pkg)

        result = {
            "metric": "views/sec fwd+bwd @1920x1080, 1M anchors", "value": round(value, 3), "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N}-anchor synthetic scene (seed 0), {W}x{H}, prefilter_voxel + "
                                   f"render(training, step={args.step_semantics}: context model + rate + rasterizer)"
                                   f" + backward, 1 view/GPU/step" + (", grad all-reduce (RCCL)" if dist_on else ""),
                       "anchors": N, "image": [W, H], "views_per_step": world, "visible_anchors": n_vis,
                       "gaussians_per_view": P, "tile_pairs_per_view": R, "R_eff": R_eff, "parallelism": f"dp{world}"},
            "value_raster_only": None if value_raster is None else round(value_raster, 3),
            "roofline": roofline, "kernels": kernels, "hip_kernel_ms_per_step": round(lib_ms, 3),
            "cpu_baseline": cpu,
        }
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def cpu_baseline(pc, cam, pipe, bg, w, pkg):
    """The oracle (oracle/raster_ref.c, OpenMP) timed on the host cores on ONE view's
    rasterizer work (fwd+bwd) of the same workload.  The reference has no CPU rasterize
    path, so this is kind="port".  Bounded: if the view has more than
    1.5 M Gaussians a seeded random subset of 1.5 M is used (stated in `sample`)."""
    import numpy as np
    import torch
    from contextgs_amd.renderer import generate_neural_gaussians
    from oracle.raster_oracle import RasterOracle
    with torch.no_grad():
        was_training = pc.get_color_mlp.training
        xyz, color, opacity, scaling, rot, *_ = generate_neural_gaussians(
            cam, pc, None, is_training=True, step=1000)
    P = xyz.shape[0]
    cap = 1_500_000
    if P > cap:
        idx = torch.randperm(P, device=xyz.device, generator=torch.Generator(device=xyz.device).manual_seed(7))[:cap]
        idx = idx.sort().values
        xyz, color, opacity, scaling, rot = xyz[idx], color[idx], opacity[idx], scaling[idx], rot[idx]
    f = lambda t: t.detach().cpu().numpy()
    oracle = RasterOracle(np.float32)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    cd = cam.oracle_dict(bg=f(bg))
    t0 = time.perf_counter()
    oracle.render(cd, f(xyz), f(color), f(opacity), f(scaling), f(rot), dL_dout=f(w))
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"1 view fwd+bwd, rasterizer stages only (R1-R8), {xyz.shape[0]} of {P} Gaussians of the "
                      f"bench view at {cam.image_width}x{cam.image_height}; {dt:.1f} s wall"}


if __name__ == "__main__":
    main()
