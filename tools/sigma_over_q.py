"""sigma/Q of the bench scene's predictions and the coder launch time per attribute (each group coded / decoded alone)
at the coarsest level: what a symbol costs on the serial chain of one wave."""
import sys, torch
sys.path.insert(0, '/root/repo')
from contextgs_amd.synth import make_scene
from contextgs_amd import codec_driver as cd
from contextgs_amd.context_model import level_plan, find_divide_scale
from contextgs_amd.encodings import Quantize_anchor
pc = make_scene(1_000_000, seed=0); pc.eval()
with torch.no_grad():
    pc.latent_codec.update(force=True)
    m = pc.get_mask_anchor
    _anchor, qa = Quantize_anchor.apply(pc._anchor[m], pc.x_bound_min, pc.x_bound_max)
    hyper = pc.latent_codec.quantize(pc._hyper_latent[m], "symbols", means=pc.latent_codec._get_medians().permute(1, 2, 0)[0])
    if pc.level_scale is None:
        pc.level_scale = find_divide_scale(pc, _anchor, pc.target_ratio, pc.level_num)
    plan, inv, mp = level_plan(pc, _anchor, None)
    level, to_code, orig, hy = plan[0]
    feat_in = torch.cat([_anchor[orig], hyper[orig].float()], dim=1)
    (mf, sf, ms, ss, mo, so, Qf, Qs, Qo) = cd._predict(pc, level, feat_in)
    for name, s_, q_ in (("feat", sf, Qf), ("scaling", ss, Qs), ("offsets", so, Qo)):
        r = (s_ / q_.reshape(-1, 1)).abs().reshape(-1)
        qs = torch.quantile(r[:1000000].float(), torch.tensor([0.1, 0.5, 0.9, 0.99], device=r.device))
        print(name, "sigma/Q quantiles 10/50/90/99 %:", [round(v, 2) for v in qs.tolist()])
    x = pc._offset[m][orig].reshape(len(orig), -1)
    print("offset |x-mean|/Q median:", ((x - mo).abs() / Qo.reshape(-1, 1)).median().item(), " range of round(x/Q):", (x / Qo.reshape(-1,1)).round().min().item(), (x / Qo.reshape(-1,1)).round().max().item())

# per-attribute coder launch time at the finest level (each group coded / decoded alone)
import time
from contextgs_amd import codec
from contextgs_amd.encodings import STE_multistep
with torch.no_grad():
    n_l, K, D = len(orig), pc.n_offsets, pc.feat_dim
    rows = torch.tensor(cd._chunk_rows(n_l), dtype=torch.int64)
    feat_q = STE_multistep.apply(pc._anchor_feat[m][orig], Qf.unsqueeze(1))
    scal_q = STE_multistep.apply(pc.get_scaling[m][orig], Qs.unsqueeze(1))
    off_q = STE_multistep.apply(pc._offset[m][orig].reshape(n_l, 3 * K), Qo.unsqueeze(1))
    for name, (xq, mu, sg, q, off, qd) in (("feat", (feat_q, mf, sf, Qf, rows * D, D)), ("scaling", (scal_q, ms, ss, Qs, rows * 6, 6)),
                                           ("offsets(all)", (off_q, mo, so, Qo, rows * 3 * K, 3 * K))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        (blob, lens, mn, mx), = codec.gaussian_encode_groups([(xq, mu, sg, q, off, qd)])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out, = codec.gaussian_decode_groups([(mu, sg, q, off, mn, mx, blob, lens, qd)])
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ok = torch.equal(out.view_as(xq), xq)
        nsym = int(off[1] - off[0])
        print(f"{name:14s} {len(lens)} streams of {nsym} symbols: encode {1e3*(t1-t0):.1f} ms, decode {1e3*(t2-t1):.1f} ms "
              f"({1e9*(t2-t1)/nsym:.0f} ns per symbol of the longest stream), bytes {int(lens.sum())}, alphabet {int((mx-mn).max())+2}, exact {ok}")
