"""Multi-process (gloo, world_size=2, CPU) coverage of the data-parallel helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        torch.manual_seed(0)                                  # identical replicas
        lin = torch.nn.Linear(4, 3)
        extra = torch.nn.Parameter(torch.zeros(5))            # gets a gradient on rank 1 only
        frozen = torch.nn.Parameter(torch.ones(2), requires_grad=False)
        big1 = torch.nn.Parameter(torch.zeros(10))            # "per-anchor" size class, gradient on rank 1 only
        cd.BIG_TENSOR = 8                                     # lin.weight (12) and big1 (10) go in place, the rest in the bucket
        params = list(lin.parameters()) + [extra, frozen, big1]
        # each rank "renders" a different view
        views = [cd.view_for(step, 8) for step in range(4)]
        x = torch.full((2, 4), float(rank + 1))
        loss = lin(x).sum()
        if rank == 1:
            loss = loss + (extra * torch.arange(5.0)).sum() + (big1 * torch.arange(10.0)).sum()
        loss.backward()
        n = cd.allreduce_gradients(params, average=True)
        stats = [torch.full((3, 1), float(rank + 1)), torch.tensor([rank], dtype=torch.int32)]
        cd.allreduce_stats(stats)
        p2 = torch.nn.Parameter(torch.full((3,), float(rank)))
        cd.broadcast_parameters([p2], src=1)
        chunks = cd.shard([bytes([i]) for i in range(7)])
        merged = cd.gather_bytes(chunks, dst=0)
        # the sharded decoder's exchange step and the encoder's byte gather
        rows = cd.all_gather_rows(torch.arange(3 + 2 * rank, dtype=torch.float32) + 10 * rank, [3, 5])
        objs = cd.gather_objects(("r", rank), dst=0)
        word = cd.broadcast_object("from0" if rank == 0 else None)
        with cd.local_only():
            alone = (cd.world(), cd.rank(), cd.all_gather_rows(torch.ones(2), [2]).tolist())
        q.put((rank, views, n, lin.weight.grad.clone(), extra.grad.clone(), stats[0].clone(), int(stats[1]),
               p2.data.clone(), merged, rows, objs, word, alone, big1.grad.clone()))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gradient_sync_and_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # views: step s -> ranks take views 2s and 2s+1
    assert res[0][1] == [0, 2, 4, 6] and res[1][1] == [1, 3, 5, 7]
    assert res[0][2] == res[1][2] == 4 * 3 + 3 + 5 + 10
    # d/dW of sum(W x + b) over 2 rows of constant x = 2 * x ; averaged over x=1 and x=2 -> 3
    for r in (0, 1):
        assert torch.allclose(res[r][3], torch.full((3, 4), 3.0))
        assert torch.allclose(res[r][4], torch.arange(5.0) / 2)        # only rank 1 had it; average over 2
        assert torch.equal(res[r][5], torch.full((3, 1), 3.0)) and res[r][6] == 1
        assert torch.equal(res[r][7], torch.full((3,), 1.0))
    assert res[0][8] == [bytes([i]) for i in range(7)] and res[1][8] is None
    for r in (0, 1):
        assert res[r][9].tolist() == [0, 1, 2, 10, 11, 12, 13, 14] and res[r][11] == "from0"
        assert res[r][12] == (1, 0, [1.0, 1.0])
        assert torch.allclose(res[r][13], torch.arange(10.0) / 2)      # in-place path, None on rank 0
    assert res[0][10] == [("r", 0), ("r", 1)] and res[1][10] is None


def test_stream_blocks_are_contiguous_balanced_and_cover_everything():
    from contextgs_amd import dist as mgpu
    import numpy as np
    rng = np.random.default_rng(0)
    for w in (1, 2, 3, 8):
        for S in (0, 1, 5, 100):
            lens = rng.integers(0, 50, S)
            edges = np.concatenate([[0], np.cumsum(lens)])
            b = mgpu.stream_blocks(edges, w)
            assert len(b) == w + 1 and b[0] == 0 and b[-1] == S and all(x <= y for x, y in zip(b, b[1:]))
            if S >= 20:
                sizes = [edges[b[i + 1]] - edges[b[i]] for i in range(w)]
                assert max(sizes) - min(sizes) <= 2 * 50
    with mgpu.local_only():
        assert mgpu.world() == 1 and mgpu.rank() == 0


def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        cd.BIG_TENSOR = 8
        torch.manual_seed(0)                                  # identical replicas
        a = torch.nn.Parameter(torch.randn(40, 3))            # "per-anchor" tensors: in place, hook-driven
        b = torch.nn.Parameter(torch.randn(40, 2))
        c = torch.nn.Parameter(torch.randn(40, 1))            # no gradient on rank 0
        lin = torch.nn.Linear(3, 2)                           # small: flat bucket
        params = [a, b, c] + list(lin.parameters())
        touched = {}

        def rows_of(p):                                       # rows this rank's "view" produced gradients for
            return touched.get(id(p))

        sync = cd.GradientSync(params, average=True, sparse=rows_of, sparse_below=0.6)
        out = []
        for step in range(3):                                 # step 0 uses the parameter order, later steps rank 0's order
            for p in params:
                p.grad = None
            torch.manual_seed(100 + 10 * step + rank)         # different "views": RNG streams diverge between ranks
            vis = torch.zeros(40, dtype=torch.bool)
            vis[torch.randperm(40)[:8 + 4 * rank]] = True     # each rank sees a few rows -> union below 60 %
            touched[id(a)] = vis
            x = torch.randn(40, 3)
            # b is used BEFORE a in the graph so that gradients become final in the order a, b (reverse of use) or not:
            loss = (lin(a * vis[:, None].float()) * x[:, :2]).sum() + (b * float(rank + 1)).sum()
            if rank == 1:
                loss = loss + (c * torch.arange(40.0)[:, None]).sum()
            # reference: plain dense average of the local gradients; torch.autograd.grad does not touch .grad, so the
            # hooks (which start reducing in place DURING loss.backward()) do not see this pass
            local = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
            loss.backward()
            nbytes = sync.finish()
            ref = []
            for g, p in zip(local, params):
                g = torch.zeros_like(p) if g is None else g.clone()
                dist.all_reduce(g)
                ref.append(g / world)
            out.append(([p.grad.tolist() for p in params], [t.tolist() for t in ref], nbytes, list(sync.order)))
        # same random draw on both ranks although their generators have diverged
        shared = cd.shared_rand_like(torch.empty(5))
        own = torch.rand(5)
        sync.close()
        q.put((rank, out, shared.tolist(), own.tolist()))             # plain lists: no tensor fd passing at exit
    finally:
        dist.destroy_process_group()


def test_gradient_sync_hooks_sparse_rows_and_shared_rng():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for grads, ref, nbytes, order in res[r][1]:
            for g, e in zip(grads, ref):
                assert torch.allclose(torch.tensor(g), torch.tensor(e), atol=1e-6), (g, e)
            assert nbytes > 0 and sorted(order) == [0, 1, 2]
    for step in range(3):                                     # replicas hold identical reduced gradients
        for g0, g1 in zip(res[0][1][step][0], res[1][1][step][0]):
            assert g0 == g1
    # the sparse path moved fewer bytes than a dense reduction of `a` would have (40 x 3 floats = 480 B of the total)
    dense_total = (40 * 3 + 40 * 2 + 40 * 1) * 4 + (6 + 2) * 4
    assert res[0][1][0][2] < dense_total
    assert res[0][2] == res[1][2] and res[0][3] != res[1][3]


def _nograd_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        cd.BIG_TENSOR = 8
        torch.manual_seed(0)
        a = torch.nn.Parameter(torch.randn(20, 2))            # big, always differentiated
        late = torch.nn.Parameter(torch.randn(20, 1))         # big, no gradient anywhere before step 2 (then rank 1 only):
        never = torch.nn.Parameter(torch.randn(3))            # the schedule of _hyper_latent / mlp_grid before iteration 10000
        params = [a, late, never]
        opt = torch.optim.Adam(params, lr=0.1)
        sync = cd.GradientSync(params, average=True)
        trace = []
        for step in range(4):
            opt.zero_grad(set_to_none=True)
            loss = (a * float(rank + 1)).sum()
            if step >= 2 and rank == 1:
                loss = loss + (late * 3.0).sum()
            loss.backward()
            nbytes = sync.finish()
            trace.append((late.grad is None, never.grad is None, nbytes,
                          None if late.grad is None else late.grad.flatten().tolist()))
            opt.step()
        sync.close()
        # the non-hook variant applies the same rule
        for p in params:
            p.grad = None
        (a * 1.0).sum().backward()
        cd.allreduce_gradients(params)
        plain = (late.grad is None, never.grad is None, a.grad.flatten().tolist())
        st = lambda p: int(opt.state[p]["step"]) if p in opt.state and "step" in opt.state[p] else 0
        q.put((rank, trace, (st(a), st(late), st(never)), late.detach().flatten().tolist(), plain))
    finally:
        dist.destroy_process_group()


def test_parameters_without_a_gradient_on_any_rank_stay_without_one():
    """under world > 1 a parameter no rank differentiated must keep .grad = None, so that Adam's step counter
    (bias correction) matches single-GPU training; zeros are contributed only where some other rank has a gradient."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nograd_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        trace, steps, late, plain = res[r][1], res[r][2], res[r][3], res[r][4]
        assert [t[0] for t in trace] == [True, True, False, False] and all(t[1] for t in trace)
        assert trace[2][3] == [1.5] * 20                        # rank 1's 3.0 averaged with rank 0's zeros
        assert trace[1][2] < trace[2][2]                         # no zero all-reduce of `late` while nobody has a gradient
        assert steps == (4, 2, 0)                                # Adam stepped `late` twice, `never` not at all
        assert plain[0] and plain[1] and plain[2] == [1.0] * 40
    assert res[0][3] == res[1][3]                                # replicas identical
    # world-1 reference of the same schedule: two Adam steps on a constant gradient 1.5 move every entry by 2 * lr
    torch.manual_seed(0)
    torch.randn(20, 2)
    late0 = torch.randn(20, 1).flatten()
    assert torch.allclose(torch.tensor(res[0][3]), late0 - 0.2, atol=1e-5)
