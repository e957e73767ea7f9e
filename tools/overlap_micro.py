"""Do two of the step's backward kernels overlap productively on two HIP streams?  (Decides whether moving the weight
gradients to a side stream can pay.)  A = anchor-MLP backward incl. weight gradients at 1 M rows, B = the step-size
context MLP backward (recompute kernel + weight gradients) at 807 k rows.  python tools/overlap_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
from contextgs_amd import mlp

dev = "cuda"
torch.manual_seed(0)
mk = lambda i, h, o, act=None: nn.Sequential(nn.Linear(i, h), nn.ReLU(True), nn.Linear(h, o), *([act] if act else [])).to(dev)
mo, mc, mv = mk(54, 50, 10, nn.Tanh()), mk(54, 50, 30, nn.Sigmoid()), mk(54, 50, 70)
grid = mk(71, 100, 175)
xa = torch.randn(1_000_000, 54, device=dev, requires_grad=True)
xb = torch.randn(807_417, 71, device=dev, requires_grad=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd_a():
    yo, yc, yv = mlp.anchor_mlp3(xa, mo, mc, mv)
    return yo.sum() + yc.sum() + yv.sum()


def fwd_b():
    return mlp.mlp2_weights(xb, grid[0].weight, grid[0].bias, grid[2].weight[172:], grid[2].bias[172:]).sum()


def run(mode, iters=20):
    la, lb = [], []
    for _ in range(iters):
        la.append(fwd_a()); lb.append(fwd_b())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if mode == "serial":
        for a, b in zip(la, lb):
            a.backward(); b.backward()
    else:
        # backward kernels run on the stream the forward ran on: rebuild the graphs on the two streams
        la, lb = [], []
        for _ in range(iters):
            with torch.cuda.stream(s1):
                la.append(fwd_a())
            with torch.cuda.stream(s2):
                lb.append(fwd_b())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, b in zip(la, lb):
            with torch.cuda.stream(s1):
                a.backward()
            with torch.cuda.stream(s2):
                b.backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def only(which, iters=20):
    ls = [fwd_a() if which == "a" else fwd_b() for _ in range(iters)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for l in ls:
        l.backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


for _ in range(2):
    print(f"A alone {only('a'):.3f} ms, B alone {only('b'):.3f} ms, serial A+B {run('serial'):.3f} ms, two streams {run('streams'):.3f} ms")
