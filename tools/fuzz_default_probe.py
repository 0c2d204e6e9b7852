"""Where do default-mode renders of random patches leave the 1e-5 band?  (diagnostic: error growth over time)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
from oracle import oracle as O
from tests import fuzz_patches as fz
S = srack_pkg.load()
NAMES = {0: "out", 1: "osc", 2: "vcf", 3: "adsr", 4: "vca", 5: "mix", 6: "math", 7: "grid", 8: "pat", 10: "smp"}
for seed in [int(a) for a in sys.argv[1:]] or [23, 102, 6, 94]:
    B, build, overrides = fz.random_patch(seed)
    V, T = 67, 1300
    o = O.OraclePatch(48000, B, 2); ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    p = S.Patch(48000, B, 2); build(p); p.configure_voices(V)
    for m, f, vals in ov: p.set_voice_field(m, f, vals)
    fr = p.render_channels(T, 0)
    err = np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
    per_t = err.max(axis=(0, 2))
    first6 = int(np.argmax(per_t > 1e-6)) if (per_t > 1e-6).any() else -1
    first5 = int(np.argmax(per_t > 1e-5)) if (per_t > 1e-5).any() else -1
    voices_bad = int((err.max(axis=(0, 1)) > 1e-5).sum())
    types = [NAMES[p.module_type(m)] for m in range(p.num_modules())]
    print(f"seed {seed} B={B} modules {types}")
    print(f"   first t with err>1e-6: {first6}, >1e-5: {first5}; voices affected {voices_bad}/{V}; err at t=0..4 {per_t[:5]}; plan {p.plan()} delayed {p.delayed_edges()}")
    if first5 >= 0:
        v = int(np.argmax(err.max(axis=(0, 1)))); c = int(np.argmax(err[:, :, v].max(axis=1)))
        t = int(np.argmax(err[c, :, v] > 1e-5))
        print(f"   worst voice {v} ch {c}: first bad t={t}: gpu {fr[c, t-2:t+3, v]} ref {ref[c, t-2:t+3, v]}")
