import sys, time, torch, tempfile, os
sys.path.insert(0, '.')
from contextgs_amd.synth import make_scene
from contextgs_amd import codec_driver, codec
import contextgs_amd.encodings as enc
pc = make_scene(1_000_000, seed=0, requires_grad=False); pc.eval()
d = tempfile.mkdtemp()
# instrument
def timed(mod, name):
    f = getattr(mod, name)
    acc = {'t': 0.0, 'n': 0}
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a, **k); torch.cuda.synchronize(); acc['t'] += time.perf_counter() - t0; acc['n'] += 1; return r
    setattr(mod, name, w); return acc
a1 = timed(codec, 'gaussian_encode_streams'); a2 = timed(codec_driver, 'encoder'); a3 = timed(codec, 'rans_encode_channels')
a4 = timed(codec, 'gaussian_decode_streams'); a5 = timed(codec_driver, 'decoder'); a6 = timed(codec, 'rans_decode_channels')
a7 = timed(codec_driver, 'level_plan')
for it in range(2):
    for a in (a1,a2,a3,a4,a5,a6,a7): a['t']=0; a['n']=0
    torch.cuda.synchronize(); t0=time.perf_counter(); pc.conduct_encoding(d); torch.cuda.synchronize(); t1=time.perf_counter()
    dec = make_scene(1_000_000, seed=0, requires_grad=False); dec.eval()
    torch.cuda.synchronize(); t2=time.perf_counter(); dec.conduct_decoding(d); torch.cuda.synchronize(); t3=time.perf_counter()
    print(f"enc {t1-t0:.3f}s dec {t3-t2:.3f}s | gauss_enc {a1['t']:.3f} ({a1['n']}) masks_enc {a2['t']:.3f} rans_enc {a3['t']:.3f} ({a3['n']}) | gauss_dec {a4['t']:.3f} masks_dec {a5['t']:.3f} rans_dec {a6['t']:.3f} | level_plan {a7['t']:.3f}")
