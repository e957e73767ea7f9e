"""Multi-process (gloo, world_size=2, CPU) coverage of the data-parallel helpers."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp



def _by_value(obj):
    """tensors -> numpy arrays (pickled by value) before a result goes into the mp queue: a torch tensor travels as a
    file descriptor that the parent fetches from the CHILD, which fails when the child has already exited."""
    if isinstance(obj, torch.Tensor):
        return ("__tensor__", obj.detach().cpu().numpy().copy())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_by_value(o) for o in obj)
    if isinstance(obj, dict):                 # (a dict's tensors went by descriptor until round 6: the race the docstring names)
        return {k: _by_value(v) for k, v in obj.items()}
    return obj


def _from_value(obj):
    if isinstance(obj, tuple) and len(obj) == 2 and isinstance(obj[0], str) and obj[0] == "__tensor__":
        return torch.from_numpy(obj[1])
    if isinstance(obj, (list, tuple)):
        return type(obj)(_from_value(o) for o in obj)
    if isinstance(obj, dict):
        return {k: _from_value(v) for k, v in obj.items()}
    return obj


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
        q.put(_by_value((rank, views, n, lin.weight.grad.clone(), extra.grad.clone(), stats[0].clone(), int(stats[1]),
                         p2.data.clone(), merged, rows, objs, word, alone, big1.grad.clone())))
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
        r = _from_value(q.get(timeout=120))
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
        q.put(_by_value((rank, out, shared.tolist(), own.tolist())))  # by value: no tensor fd passing at exit
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
        r = _from_value(q.get(timeout=120))
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
        q.put(_by_value((rank, trace, (st(a), st(late), st(never)), late.detach().flatten().tolist(), plain)))
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
        r = _from_value(q.get(timeout=120))
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


class _DeferringLinear(torch.autograd.Function):
    """Stand-in for the MLP nodes of contextgs_amd/mlp.py on CPU: y = x W^T with the data gradient returned at once and
    the weight gradient left to mlp's end-of-backward queue when deferral is on."""

    @staticmethod
    def forward(ctx, x, W, log):
        ctx.save_for_backward(x, W.detach())
        ctx.W, ctx.log = W, log
        return x @ W.detach().t()

    @staticmethod
    def backward(ctx, g):
        from contextgs_amd import mlp
        x, Wd = ctx.saved_tensors
        dx = g @ Wd
        if not mlp._can_defer((ctx.W,)):
            return dx, g.t() @ x, None
        W, log = ctx.W, ctx.log

        def job():
            log.append("wgrad")
            mlp._accumulate((W,), (g.t() @ x,))
        mlp._defer(job)
        return dx, None, None


def _defer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        from contextgs_amd import mlp
        torch.manual_seed(0)
        W = torch.nn.Parameter(torch.randn(3, 4))             # "MLP weight": small bucket, gradient deferred
        big = torch.nn.Parameter(torch.randn(16, 4))          # "per-anchor" tensors: reduced in place from the hooks
        big2 = torch.nn.Parameter(torch.randn(16, 4))
        cd.BIG_TENSOR = 32
        assert not mlp._Deferred.on
        sync = cd.GradientSync([W, big, big2], average=True)
        assert mlp._Deferred.on                               # world > 1: GradientSync switched the deferral on
        log = []
        orig_issue = sync._issue
        sync._issue = lambda p: (log.append("issue"), orig_issue(p))[1]
        results = []
        for step in range(2):                                 # second step: W is used by TWO nodes and .grad accumulates
            x = big * float(rank + 1) + (big2 if step else 0.0)
            y = _DeferringLinear.apply(x, W, log)
            if step:
                y = y + _DeferringLinear.apply(big2 * 2.0, W, log)
            y.sum().backward()
            assert not mlp._Deferred.queue and not mlp._Deferred.armed and not mlp._Deferred.staged
            sync.finish()
            results.append([t.grad.clone() for t in (W, big, big2) if t.grad is not None])
            if step == 0:
                for t in (big, big2):
                    t.grad = None                              # W.grad is kept: the next backward must ADD to it
        sync.close()
        assert not mlp._Deferred.on and not mlp._Deferred.before_flush
        q.put(_by_value((rank, log, results)))
    finally:
        dist.destroy_process_group()


def test_deferred_weight_gradients_follow_the_per_anchor_collectives():
    """dist.GradientSync + mlp.defer_weight_gradients on two gloo ranks with a CPU stand-in for the MLP nodes: every
    per-anchor collective is issued BEFORE the first deferred weight-gradient job runs, a parameter used by two nodes gets
    the sum of both, an existing .grad is added to, and the reduced values are what an inline backward + all-reduce gives."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_defer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted((_from_value(q.get(timeout=120)) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, log, results in out:
        # step 0: both per-anchor tensors are predicted active (big2 without a gradient anywhere: zeros, dropped again in
        # finish()) and are issued BEFORE the deferred weight gradient runs; step 1: big is issued first, then the two queued
        # weight gradients, and big2 — predicted inactive after step 0 — only after finish()'s has-grad mask
        assert log == ["issue", "issue", "wgrad", "issue", "wgrad", "wgrad", "issue"], log
    # reference: the same two steps in one process per "rank", inline gradients, averaged by hand
    torch.manual_seed(0)
    W = torch.randn(3, 4); big = torch.randn(16, 4); big2 = torch.randn(16, 4)
    ref = []
    for step in range(2):
        gW, gb, gb2 = [], [], []
        for rank in range(world):
            Wp, bp, b2p = (t.clone().requires_grad_(True) for t in (W, big, big2))
            x = bp * float(rank + 1) + (b2p if step else 0.0)
            y = x @ Wp.t()
            if step:
                y = y + (b2p * 2.0) @ Wp.t()
            y.sum().backward()
            gW.append(Wp.grad); gb.append(bp.grad); gb2.append(b2p.grad if b2p.grad is not None else torch.zeros_like(b2p))
        ref.append((sum(gW) / world, sum(gb) / world, sum(gb2) / world))
    for rank, log, results in out:
        w0, b0 = results[0][0], results[0][1]
        torch.testing.assert_close(w0, ref[0][0]); torch.testing.assert_close(b0, ref[0][1])
        # step 1: W.grad carried over from step 0 (already averaged) + this step's average
        torch.testing.assert_close(results[1][0], ref[0][0] + ref[1][0])
        torch.testing.assert_close(results[1][1], ref[1][1]); torch.testing.assert_close(results[1][2], ref[1][2])


def _stats_worker(rank, world, port, q):
    """Two densification rounds of the statistics reduction (densify.reduce_statistics) on a stand-in model: per-rank
    accumulations, the resets adjust_anchor applies above its thresholds, carry-over below them."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import densify

        class PC:
            pass
        pc = PC()
        n = 12
        for name in densify._STATS:
            setattr(pc, name, torch.zeros(n, 1))
        out = []
        for rnd in range(3):
            g = torch.Generator().manual_seed(100 * rnd + rank)           # what THIS rank's views add in this round
            for name in densify._STATS:
                getattr(pc, name).add_(torch.randint(0, 5, (n, 1), generator=g).float())
            densify.reduce_statistics(pc)
            densify.reduce_statistics(pc)                                  # idempotent
            out.append([getattr(pc, name).clone() for name in densify._STATS])
            # adjust_anchor's resets: entries above a threshold are zeroed, the rest carry over (densify.py, :894-896)
            for name in densify._STATS:
                t = getattr(pc, name)
                t[t > 6] = 0
            pc._stats_base = [getattr(pc, name).clone() for name in densify._STATS]
        q.put(_by_value((rank, out)))
    finally:
        dist.destroy_process_group()


def test_statistics_carry_over_is_counted_once_across_densification_rounds():
    """an in-place all-reduce of the statistics buffers every round counts the carry-over of un-reset entries
    once per rank.  reduce_statistics sums only what the ranks added since they last agreed: three rounds on two ranks
    equal ONE process that sees both ranks' views."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = _from_value(q.get(timeout=120))
        res[r[0]] = r[1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the single-process run over the union of the views
    n = 12
    bufs = [torch.zeros(n, 1) for _ in range(4)]
    for rnd in range(3):
        for rank in range(world):
            g = torch.Generator().manual_seed(100 * rnd + rank)
            for t in bufs:
                t.add_(torch.randint(0, 5, (n, 1), generator=g).float())
        for k, t in enumerate(bufs):
            assert torch.equal(res[0][rnd][k], t) and torch.equal(res[1][rnd][k], t), (rnd, k)
        for t in bufs:
            t[t > 6] = 0
    assert any(float(t.sum()) > 0 for t in bufs)          # some entries did carry over


def _auto_rows_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        cd.BIG_TENSOR = 8
        torch.manual_seed(0)
        n = 64
        a = torch.nn.Parameter(torch.randn(n, 3))             # three "per-anchor" tensors sharing the view's visible rows
        b = torch.nn.Parameter(torch.randn(n, 2))
        c = torch.nn.Parameter(torch.randn(n, 5))
        lin = torch.nn.Linear(3, 2)
        params = [a, b, c] + list(lin.parameters())
        sync = cd.GradientSync(params, average=True)          # sparse="auto" is the default
        out = []
        for step, dense_phase in enumerate((False, False, True)):
            for p in params:
                p.grad = None
            vis = torch.zeros(n, dtype=torch.bool)
            vis[(n // 4) * rank:(n // 4) * (rank + 1)] = True          # DISJOINT quarters: the union is half of the rows
            # what renderer.generate_neural_gaussians does: note the rows, or None when the context model touches every anchor
            cd.note_touched_rows(None if dense_phase else vis, int(vis.sum()))
            m = torch.ones(n, 1) if dense_phase else vis[:, None].float()
            loss = (lin(a * m)).sum() * float(rank + 1) + (b * m).sum() + (c * m * float(step + 1)).sum()
            local = torch.autograd.grad(loss, params, retain_graph=True, allow_unused=True)
            loss.backward()
            nbytes = sync.finish()
            ref = []
            for g, p in zip(local, params):
                g = torch.zeros_like(p) if g is None else g.clone()
                dist.all_reduce(g)
                ref.append(g / world)
            out.append(([p.grad.tolist() for p in params], [t.tolist() for t in ref], nbytes))
        rep = cd.diagnostics(n * 10 * 4, 64, reps=2)
        exp = sync.exposure_report()
        sync.close()
        q.put(_by_value((rank, out, rep, exp)))
    finally:
        dist.destroy_process_group()


def test_touched_rows_are_the_default_and_the_group_reports_itself():
    """(b) with views that see disjoint sets of anchors the per-anchor gradients travel as the UNION of the
    touched rows by default (one mask exchange per step shared by the tensors of the view), and dense again in the phase in
    which the context model differentiates every anchor; (a) diagnostics() reports ranks and measured all-reduce bandwidths."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_auto_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = _from_value(q.get(timeout=120))
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 64
    dense_big = n * (3 + 2 + 5) * 4
    small = (6 + 2) * 4
    for r in (0, 1):
        for step, (grads, ref, nbytes) in enumerate(res[r][1]):
            for g, e in zip(grads, ref):
                assert torch.allclose(torch.tensor(g), torch.tensor(e), atol=1e-6), (step, g, e)
        # sparse steps: ONE n-byte mask + half of the rows of each tensor (+ the small bucket); dense step: everything
        assert res[r][1][0][2] == n + dense_big // 2 + small, res[r][1][0][2]
        assert res[r][1][1][2] == n + dense_big // 2 + small
        assert res[r][1][2][2] == dense_big + small
        rep = res[r][2]
        assert rep["world"] == 2 and rep["backend"] == "gloo" and [x["rank"] for x in rep["ranks"]] == [0, 1]
        for tag in ("per_anchor_payload", "small_bucket"):
            e = rep["all_reduce"][tag]
            assert e["bytes"] > 0 and e["median_ms"] > 0 and e["busbw_GBps"] > 0
        assert res[r][3]["steps"] == 3 and res[r][3]["mean_ms"] >= 0
    for step in range(3):
        for g0, g1 in zip(res[0][1][step][0], res[1][1][step][0]):
            assert g0 == g1


def test_a_new_sync_closes_the_one_it_replaces_and_leaves_an_unrelated_one_alone():
    """ A GradientSync over (some of) the same parameters replaces the open one — the documented pattern after
    adjust_anchor, where the per-anchor Parameters are new objects and the MLP Parameters are not; a sync over OTHER
    parameters (a second model) stays open next to it and the constructor warns instead of silently removing its hooks."""
    import warnings
    import torch
    from contextgs_amd import dist as cd
    big = lambda: torch.nn.Parameter(torch.zeros(cd.BIG_TENSOR // 4 + 4, 4))
    w = torch.nn.Parameter(torch.zeros(3, 3))
    a1, a2 = big(), big()
    s1 = cd.GradientSync([w, a1])
    assert s1._handles
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s2 = cd.GradientSync([w, a2])                    # same model after optimizer surgery: a1 is gone, w is shared
    assert not s1._handles and s2._handles
    other = [torch.nn.Parameter(torch.zeros(2, 2)), big()]
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        s3 = cd.GradientSync(other)
    assert s2._handles and s3._handles, "an unrelated sync must keep its hooks"
    assert any("still open" in str(r.message) for r in rec)
    s2.close(); s3.close()


def _dense_outside_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        cd.BIG_TENSOR = 8
        torch.manual_seed(0)
        n = 64
        feat = torch.nn.Parameter(torch.randn(n, 3))          # gradient confined to the view's rows
        mask = torch.nn.Parameter(torch.randn(n, 2))          # + a dense regulariser on EVERY row (train.py:209, 3000 < step <= 10000)
        params = [feat, mask]
        sync = cd.GradientSync(params, average=False)         # SUM: outside rows must become world x g, not g
        out = []
        for step in range(3):
            for p in params:
                p.grad = None
            vis = torch.zeros(n, dtype=torch.bool)
            vis[(n // 4) * rank:(n // 4) * (rank + 1)] = True
            cd.note_touched_rows(vis, int(vis.sum()))
            m = vis[:, None].float()
            loss = (feat * m).sum() * float(rank + 1) + (mask * m).sum() * float(step + 1) + 0.5 * torch.sigmoid(mask).mean()
            local = torch.autograd.grad(loss, params, retain_graph=True)
            loss.backward()
            nbytes = sync.finish()
            ref = []
            for g in local:
                g = g.clone()
                dist.all_reduce(g)
                ref.append(g)
            out.append(([p.grad.clone() for p in params], ref, nbytes, sorted(len(sync._dense_only) for _ in (0,))))
        sync.close()
        q.put(_by_value((rank, out)))
    finally:
        dist.destroy_process_group()


def test_a_dense_regulariser_outside_the_noted_rows_is_reduced_too():
    """sparse="auto" exchanges only the union of the rows the renderer noted; a loss term outside the
    renderer (the mask regulariser of train.py:209) puts gradient on every row of `_mask`.  Under SUM those rows must come out
    as the sum over ranks — detected on the step it happens (repaired by one dense collective), dense from then on."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dense_outside_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = _from_value(q.get(timeout=120))
        res[r[0]] = r[1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 64
    for r in (0, 1):
        for step, (grads, ref, nbytes, n_dense) in enumerate(res[r]):
            for g, e in zip(grads, ref):
                assert torch.allclose(g, e, atol=1e-6), (r, step, (g - e).abs().max())
            assert n_dense == [1]                    # `mask` left the touched-rows path on the first step, `feat` never does
        # step 0: mask exchanged compactly, found out, repaired densely; later steps: feat compact (half the rows), mask dense
        assert res[r][1][2] == n + n * 3 * 4 // 2 + n * 2 * 4, res[r][1][2]
        assert res[r][0][2] > res[r][1][2]


def test_broadcast_invalidates_the_entropy_bottleneck_tables():
    """update(force=True) reuses its tables while the parameters are at the version they were built from;
    dist.broadcast_parameters writes the parameters without that version moving on its own."""
    from contextgs_amd import entropy_bottleneck as eb
    g0 = eb._TABLE_GENERATION[0]
    eb.invalidate_tables()
    assert eb._TABLE_GENERATION[0] == g0 + 1
    import inspect
    from contextgs_amd import dist as cd
    assert "invalidate_tables" in inspect.getsource(cd.broadcast_parameters)
    p = torch.nn.Parameter(torch.zeros(3))
    v0 = p._version
    with torch.no_grad():
        p.detach().add_(0)
    assert p._version > v0                           # the alias shares the counter: what broadcast_parameters relies on


def _bucket_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from contextgs_amd import dist as cd
        cd.BIG_TENSOR = 8
        torch.manual_seed(0)
        a = torch.nn.Parameter(torch.randn(64, 3))
        lin = torch.nn.Linear(3, 2)
        params = [a] + list(lin.parameters())
        out = {}
        for mode in ("per_tensor", "bucket"):
            sync = cd.GradientSync(params, average=True, big_mode=mode, sparse=None)
            for p in params:
                p.grad = None
            (lin(a).sum() * float(rank + 1)).backward()
            nbytes = sync.finish()
            out[mode] = ([p.grad.clone() for p in params], nbytes)
            sync.close()
        rep = cd.diagnostics(64 * 3 * 4, 32, reps=2)
        q.put(_by_value((rank, out, cd.choose_big_mode(rep, 1))))
    finally:
        dist.destroy_process_group()


def test_bucket_mode_reduces_the_same_gradients_and_the_choice_is_computed_from_measurements():
    """GradientSync(big_mode="bucket") (one flat collective for the per-anchor tensors) gives the gradients of
    the per-tensor mode; choose_big_mode() turns diagnostics()' measured latencies into the choice bench.py prints."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = _from_value(q.get(timeout=120))
        res[r[0]] = r
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        g_pt, g_b = res[r][1]["per_tensor"][0], res[r][1]["bucket"][0]
        for x, y in zip(g_pt, g_b):
            assert torch.allclose(x, y, atol=1e-6)
        assert res[r][1]["per_tensor"][1] == res[r][1]["bucket"][1] == (64 * 3 + 6 + 2) * 4
        c = res[r][2]
        assert c["mode"] in ("per_tensor", "bucket") and c["est_per_tensor_ms"] > 0 and c["est_bucket_ms"] > 0
    from contextgs_amd import dist as cd
    fake = {"all_reduce": {"per_anchor_payload": {"bytes": 444_000_000, "median_ms": 2.0}, "small_bucket": {"bytes": 4000, "median_ms": 0.02}}}
    assert cd.choose_big_mode(fake, 6)["mode"] == "per_tensor"           # 5 x 20 us of latency < two passes over 444 MB
    fake["all_reduce"]["small_bucket"]["median_ms"] = 0.2
    assert cd.choose_big_mode(fake, 6)["mode"] == "bucket"
    assert cd.choose_big_mode(None)["mode"] == "per_tensor"
