"""Wave-iteration count of the blend kernels: current mapping (8x8 quadrant per wave, one Gaussian per iteration)
vs four 4x4 blocks per wave walking their own lists (csrc/raster_debug.hip).

The counting kernels are compiled into -DCGS_EXPERIMENTS builds only (not the product library): build one with
`bash tools/variant_lib.sh occ raster_debug.hip` and run with CGS_LIB_PATH=tools/variants/libcgs_occ.so."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib
from contextgs_amd.rasterizer import last_call
from contextgs_amd.renderer import prefilter_voxel, render
from contextgs_amd.synth import SynthPipe, make_scene, orbit_cameras
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pc = make_scene(N, seed=0); pc.train()
pipe = SynthPipe(); bg = torch.zeros(3, device="cuda")
cam = orbit_cameras(8, 1920, 1080)[0].to_torch("cuda")
vis = prefilter_voxel(cam, pc, pipe, bg)
pkg = render(cam, pc, pipe, bg, visible_mask=vis, step=1000)
L = C.CDLL(_lib.LIB_PATH)
out = torch.zeros(6, dtype=torch.int64, device="cuda")
f = L.cgs_debug_blend_occupancy
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
lc = last_call
rc = f(C.addressof(lc["cfg"].c), lc["P"], lc["bin_R"], lc["geom_ws"].data_ptr(), lc["geom_ws"].numel(),
       lc["bin_ws"].data_ptr(), lc["bin_ws"].numel(), lc["img_ws"].data_ptr(), lc["img_ws"].numel(), out.data_ptr(), None)
torch.cuda.synchronize()
q, b, v, e, o, ph = out.tolist()
print(f"rc={rc} P={lc['P']} R={lc['num_rendered']}: quadrant iterations {q}, block-mapped iterations {b} ({q / max(1, b):.2f}x fewer), 4x4 block visits {v} ({v / max(1, q):.2f} per quadrant visit)")
print(f"4x4 block visits: box test {v}, octagon test {o} ({o / max(1, v):.3f}), exact (some pixel has alpha >= 1/255) {e} ({e / max(1, v):.3f})")
print(f"(pixel, Gaussian) pairs with alpha >= 1/255 up to the tile's last contributor: {ph} = {ph / max(1, v * 16):.3f} of the 16 pixels of a visited block; {ph / max(1, lc['P']):.1f} per Gaussian")

# ---- splat-parallel backward mapping, counted (csrc/raster_debug.hip: blend_splat_occupancy_kernel) ----
out5 = torch.zeros(5, dtype=torch.int64, device="cuda")
f2 = L.cgs_debug_blend_splat_occupancy
f2.restype = C.c_int
f2.argtypes = f.argtypes
rc = f2(C.addressof(lc["cfg"].c), lc["P"], lc["bin_R"], lc["geom_ws"].data_ptr(), lc["geom_ws"].numel(),
        lc["bin_ws"].data_ptr(), lc["bin_ws"].numel(), lc["img_ws"].data_ptr(), lc["img_ws"].numel(), out5.data_ptr(), None)
torch.cuda.synchronize()
pairs, visits, hits, blkv, buckets = out5.tolist()
print(f"splat-parallel mapping rc={rc}: {buckets} 64-entry buckets; (pixel, bucket) pairs below the pixel's last contributor {pairs}, "
      f"with >= 1 hit {visits} ({visits / max(1, pairs):.3f}); lane hits {hits} = {hits / max(1, visits):.2f} of 64 lanes per visited pair "
      f"({hits / max(1, 64 * visits):.3f} lane utilisation); (4x4 block, bucket) pairs with a hit {blkv}")
ROW_INSTR, SPLAT_INSTR, SPLAT_SKIP = 84, 70, 16
print(f"wave instructions per view: row mapping {b} iterations x {ROW_INSTR} = {b * ROW_INSTR / 1e6:.0f} M;  splat-parallel "
      f"{visits} x {SPLAT_INSTR} + {pairs - visits} x {SPLAT_SKIP} = {(visits * SPLAT_INSTR + (pairs - visits) * SPLAT_SKIP) / 1e6:.0f} M "
      f"({(visits * SPLAT_INSTR + (pairs - visits) * SPLAT_SKIP) / max(1, b * ROW_INSTR):.2f}x)")

# ---- four 4x4 blocks per wave (shipped) against eight 4x2 half-blocks per wave, both with the octagon test and the per-group bound ----
out4 = torch.zeros(5, dtype=torch.int64, device="cuda")
f3 = L.cgs_debug_blend_group_occupancy
f3.restype = C.c_int
f3.argtypes = f.argtypes
rc = f3(C.addressof(lc["cfg"].c), lc["P"], lc["bin_R"], lc["geom_ws"].data_ptr(), lc["geom_ws"].numel(),
        lc["bin_ws"].data_ptr(), lc["bin_ws"].numel(), lc["img_ws"].data_ptr(), lc["img_ws"].numel(), out4.data_ptr(), None)
torch.cuda.synchronize()
i16, i32, v16, v32, i16f = out4.tolist()
print(f"group mapping rc={rc}: wave iterations with four 4x4 blocks per wave {i16} ({v16} block visits), with eight 4x2 half-blocks "
      f"per wave {i32} ({v32} half-block visits): {i32 / max(1, i16):.3f}x the iterations; four 4x4 blocks whose rows do not wait "
      f"for each other at segment boundaries {i16f} ({i16f / max(1, i16):.3f}x)")
