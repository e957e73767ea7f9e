#!/usr/bin/env python
"""fp64 yardsticks + the direct extract_context_feat fixture, from the REFERENCE's own Python on CPU
(harness of tools/make_goldens.py; authoring container only):

  tests/golden/train64_{n3000,n10000}.npz  the training-mode fixtures of make_goldens.golden_training re-run with every
      parameter and activation in fp64 (same inputs, same fp32 noise): gradients of every parameter, loss, rate terms.
  tests/golden/context_feat.npz            scene/gaussian_model.py:1711-1724 `extract_context_feat` called directly on the
      n3000 model's level division for levels 1 and 2 (row b3 was only pinned through the level outputs before).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg


def golden_context_feat():
    import golden_inputs as gi
    from scene import gaussian_model as gm
    N, seed = 3000, 2
    pc = mg.build_reference_model(N, seed)
    out = {}
    with torch.no_grad():
        anchor = pc.get_anchor
        pc.level_scale = gm.find_divide_scale(pc, anchor, pc.target_ratio, pc.level_num)
        mask_anchor = torch.ones(N, dtype=torch.bool)
        _, inverse_indices_list, mapping_list, _ = gm.divide_levels(pc, anchor, mask_anchor)
        rng = np.random.default_rng(41)
        feat = torch.from_numpy(rng.normal(size=(N, gi.D)).astype(np.float32))
        scal = torch.from_numpy(rng.normal(size=(N, 6)).astype(np.float32))
        out["level_scale"] = np.asarray(pc.level_scale, dtype=np.float64)
        out["feat"], out["scaling"], out["anchor_q"] = mg.npy(feat), mg.npy(scal), mg.npy(anchor)
        for i in range(1, pc.level_num):
            for frac in (0.0, 0.3):
                coded = torch.from_numpy(rng.random(N) < frac)
                got = gm.extract_context_feat(anchor, feat, scal, coded, inverse_indices_list, mapping_list, i)
                out[f"coded_l{i}_{int(frac * 10)}"] = mg.npy(coded)
                out[f"ctx_l{i}_{int(frac * 10)}"] = mg.npy(got)
    np.savez_compressed(os.path.join(mg.OUT, "context_feat.npz"), **out)


def main():
    mg.install_stubs()
    mg.patch_cuda()
    sys.path.insert(0, mg.REF)
    torch.manual_seed(0)
    with mg.CudaToCpu():
        golden_context_feat()
        mg.golden_training(3000, 2, "n3000", 1, double=True)
        mg.golden_training(10000, 4, "n10000", 5, double=True)
    for f in sorted(os.listdir(mg.OUT)):
        if f.startswith(("train64", "context_feat")):
            print(f, os.path.getsize(os.path.join(mg.OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
