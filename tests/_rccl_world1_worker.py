"""Worker of tests/test_dist_train_gpu.py::test_rccl_backend_initialises_and_runs_our_collectives_on_one_rank: the `nccl`
(= RCCL) backend at world size 1 on cuda:0 — communicator init plus every collective flavour contextgs_amd/dist.py
issues (in-place AVG / SUM all-reduce with async_op on fp32, MAX on int32 and uint8 masks, broadcast, padded all_gather,
object broadcast / gather), so that the driver's first multi-GPU run is not also the first time RCCL is touched."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from contextgs_amd import dist as cd
    g = torch.randn(1 << 22, device="cuda")
    ref = g.clone()
    work = dist.all_reduce(g, op=dist.ReduceOp.AVG, async_op=True)
    flat = torch.randn(100_000, device="cuda")
    ref_flat = flat.clone()
    work2 = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
    work.wait(); work2.wait()
    assert torch.equal(g, ref) and torch.equal(flat, ref_flat)
    m = torch.tensor([0, 1, 0, 1], dtype=torch.int32, device="cuda")
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    assert m.tolist() == [0, 1, 0, 1]
    b = (torch.rand(1000, device="cuda") < 0.5).to(torch.uint8)
    keep = b.clone()
    dist.all_reduce(b, op=dist.ReduceOp.MAX)
    assert torch.equal(b, keep)
    u = torch.rand(4096, device="cuda")
    dist.broadcast(u, src=0)
    out = [torch.empty(16, device="cuda")]
    dist.all_gather(out, torch.arange(16.0, device="cuda"))
    assert out[0].tolist() == list(range(16))
    box = ["order"]
    dist.broadcast_object_list(box, src=0)
    got = [None]
    dist.gather_object(("r", 0), got, dst=0)
    assert box == ["order"] and got == [("r", 0)]
    # the module's helpers on this backend (world 1: they must not issue anything)
    p = torch.nn.Parameter(torch.randn(8, device="cuda"))
    p.grad = torch.ones(8, device="cuda")
    assert cd.allreduce_gradients([p]) == 0 and cd.world() == 1 and cd.rank() == 0
    dist.barrier()
    torch.cuda.synchronize()
    print("rccl world-1 ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
