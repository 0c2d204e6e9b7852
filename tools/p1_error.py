"""Default-mode error of config 3's patch against the oracle over a full second: max |gpu - ref| / max(|ref|, 1) per render flag.
usage: python tools/p1_error.py [voices] [flags...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
from oracle import oracle as O
S = srack_pkg.load(); O.build()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = 48000
det, cut = S.p1_voice_params(V)
if os.environ.get("P1_ERR_WIDE"):  # the whole range of cutoffs and resonances below the exact-mode threshold
    rng = np.random.default_rng(5)
    cut = rng.uniform(0.0, 0.9, V).astype(np.float32)
res = np.random.default_rng(6).uniform(0.0, 0.89, V).astype(np.float32) if os.environ.get("P1_ERR_WIDE") else None
o = O.OraclePatch(48000, 1024, 2)
ids = S.build_p1(o)
ov = [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)] + ([(ids["vcf"], S.VCF_RES, res)] if res is not None else [])
ref, _ = o.render_batch(V, T, ov, threads=8)
r64 = ref.astype(np.float64)
for flags in [int(a) for a in sys.argv[2:]] or [0, 2]:
    p = S.Patch(48000, 1024, 2)
    S.build_p1(p)
    p.configure_voices(V)
    for m, f, vals in ov: p.set_voice_field(m, f, vals)
    fr = p.render_channels(T, flags).astype(np.float64)
    err = np.abs(fr - r64) / np.maximum(np.abs(r64), 1.0)
    print(f"flags {flags}: max {err.max():.3e}  mean {err.mean():.3e}  rms {np.sqrt((err ** 2).mean()):.3e}  worst voice {int(err.max(axis=(0, 1)).argmax())}  {p.info()[:90]}")
