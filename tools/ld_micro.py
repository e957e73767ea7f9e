"""Does the row stride of the level-MLP input matter?  cgs_mlp2_forward (71 -> 100 -> 3, the step-size head that runs on every
anchor of a level) on the same values with ldx = 71 (284-byte rows, as rowcat writes them) and ldx = 80 (320-byte rows: every
16-column chunk of a row is one aligned 64-byte sector).  python tools/ld_micro.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from contextgs_amd import _lib
L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
dev = "cuda"
for (i, h, o) in ((71, 100, 3), (15, 100, 3), (71, 100, 175)):
    nn = n if o == 3 else n // 6
    W1 = torch.randn(h, i, device=dev); b1 = torch.randn(h, device=dev); W2 = torch.randn(o, h, device=dev); b2 = torch.randn(o, device=dev)
    x = torch.randn(nn, i, device=dev)
    ref = None
    for ld in (i, (i + 15) // 16 * 16):
        xb = torch.zeros(nn, ld, device=dev); xb[:, :i] = x
        for ldy in ((o,) if o == 3 else (o, (o + 15) // 16 * 16)):
            y = torch.empty(nn, ldy, device=dev)
            f = lambda: _lib.check(L.cgs_mlp2_forward(i, h, o, 0, _lib.ptr(xb), ld, _lib.ptr(W1), _lib.ptr(b1), _lib.ptr(W2), _lib.ptr(b2),
                                                      _lib.ptr(y), ldy, None, nn, _lib.current_stream()), "fwd")
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            out = y[:, :o].clone()
            if ref is None: ref = out
            print(f"{i}->{h}->{o} n={nn}: ldx {ld:3d} ldy {ldy:3d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   same values: {bool(torch.equal(out, ref))}")
