"""Worker of tests/test_dist_codec_gpu.py: world_size 2 on ONE GPU (gloo rendezvous, both ranks on cuda:0).
Sharded conduct_encoding / conduct_decoding (contextgs_amd.dist: contiguous stream blocks per rank, byte gather to
rank 0, all-gather of decoded values between levels) must produce byte-identical files and bit-identical decoded
parameters to the single-process path (dist.local_only)."""
import filecmp
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import golden_inputs as gi                                   # noqa: E402
from contextgs_amd import dist as mgpu                       # noqa: E402
from contextgs_amd.model import GaussianModel                # noqa: E402


def build(N, seed):
    pc = GaussianModel(voxel_size=0.01)
    sd = pc.state_dict()
    for k, v in gi.mlp_weights(seed).items():
        sd[k] = torch.from_numpy(v).cuda()
    pc.load_state_dict(sd, strict=False)
    st = gi.anchor_state(N, seed)
    pc.set_state(st["anchor"], st["offset"], st["mask"], st["feat"], st["hyper"], st["scaling"])
    pc.update_anchor_bound()
    pc.eval()
    return pc


def scrambled(N, seed):
    pc = build(N, seed)
    with torch.no_grad():
        pc._anchor_feat.zero_(); pc._offset.zero_(); pc._hyper_latent.zero_(); pc._scaling.zero_()
        for p in pc.mlp_grid.parameters():
            p.zero_()
    return pc


def main():
    out, N, seed = sys.argv[1], int(sys.argv[2]), 5
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    r = dist.get_rank()
    assert mgpu.world() == 2
    d_multi, d_single = os.path.join(out, "multi"), os.path.join(out, f"single{r}")
    enc = build(N, seed)
    info = enc.conduct_encoding(d_multi)                      # sharded; every rank gets rank 0's summary
    assert "EncTime" in info
    with mgpu.local_only():
        assert mgpu.world() == 1
        enc.conduct_encoding(d_single)
    names = ["anchor.npy", "hyper.b", "masks.b"] + [f"{a}{l}.b" for a in ("feat", "scaling", "offsets") for l in range(3)]
    match, mismatch, errors = filecmp.cmpfiles(d_multi, d_single, names, shallow=False)
    assert not mismatch and not errors, (mismatch, errors)
    from container_digest import _canon            # header lists hold numpy arrays in container version 2
    meta_m, meta_s = (_canon(torch.load(os.path.join(d, "meta.b"), weights_only=False)) for d in (d_multi, d_single))
    assert meta_m == meta_s
    assert (len(meta_m) == 15) == (os.environ.get("CGS_CONTAINER_VERSION") == "2")
    dist.barrier()

    dec_m = scrambled(N, seed)
    dec_m.conduct_decoding(d_multi)                           # sharded: all-gather of decoded values between levels
    dec_s = scrambled(N, seed)
    with mgpu.local_only():
        dec_s.conduct_decoding(d_single)
    for name in ("_anchor", "_anchor_feat", "_offset", "_scaling", "_mask", "_hyper_latent"):
        assert torch.equal(getattr(dec_m, name), getattr(dec_s, name)), name
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {r}: sharded codec == single-process codec")


if __name__ == "__main__":
    main()
