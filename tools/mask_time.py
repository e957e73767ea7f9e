import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from contextgs_amd.synth import make_scene
from contextgs_amd import codec
pc = make_scene(1_000_000, seed=0); pc.eval()
m = pc.get_mask[pc.get_mask_anchor]
p = (m.sum() / m.numel()).item()
sym = torch.floor(((m * 2 - 1).view(-1) + 1) / 2).to(torch.int16).cpu().numpy()
print("symbols", sym.size, "p", p)
for _ in range(3):
    t = time.perf_counter(); b = codec.bernoulli_encode_host(sym, p); t1 = time.perf_counter()
    s = codec.bernoulli_decode_host(b, sym.size, p); t2 = time.perf_counter()
    print(f"encode {1e3*(t1-t):.1f} ms decode {1e3*(t2-t1):.1f} ms bytes {len(b)}")
