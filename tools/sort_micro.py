"""Depth-sort micro-benchmark (GPU box): cgs_sort_depth_keys (27-bit keys, three 9-bit passes) against cgs_sort_pairs_u32 on the
full float bits (four 8-bit passes) at the headline view's size, HIP events around 50 calls each; checks the orders agree.
usage: python tools/sort_micro.py [n]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from contextgs_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_798_070
L = _lib.lib()
gen = torch.Generator(device="cuda").manual_seed(1)
z = torch.exp(torch.empty(n, device="cuda").uniform_(math.log(1.0), math.log(9.0), generator=gen))      # an orbit view's depth range
keys = z.view(torch.int32).clone()
keys[torch.rand(n, device="cuda", generator=gen) < 0.03] = -1
ko, vo, kt, vt, vo2 = (torch.empty_like(keys) for _ in range(5))
scratch = torch.empty(L.cgs_sort_scratch_bytes(n), dtype=torch.uint8, device="cuda")
flag = torch.zeros(4, dtype=torch.int32, device="cuda")
st = _lib.current_stream()


def depth():
    _lib.check(L.cgs_sort_depth_keys(_lib.ptr(keys), _lib.ptr(ko), _lib.ptr(vo), _lib.ptr(kt), _lib.ptr(vt), n, _lib.ptr(scratch),
                                     scratch.numel(), _lib.ptr(flag), 5, st), "sort_depth_keys")


def full():
    _lib.check(L.cgs_sort_pairs_u32(_lib.ptr(keys), None, _lib.ptr(ko), _lib.ptr(vo2), _lib.ptr(kt), _lib.ptr(vt), n, 0, 32,
                                    _lib.ptr(scratch), scratch.numel(), st), "sort")


for name, f in (("depth27 (3 x 9 bit)", depth), ("full32 (4 x 8 bit)", full)):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        f()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per sort of {n} keys")
live = int((keys != -1).sum())
print("orders agree on the live keys:", bool(torch.equal(vo[:live], vo2[:live])), "overflow flag", int(flag[0]))
