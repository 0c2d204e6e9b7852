"""Config 4 (P2 FM with feedback) over a full second: error of the default and exact modes against the oracle (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
from oracle import oracle as O
S = srack_pkg.load()
V, T = 64, 48000
beta, index = S.p2_voice_params(V)
for B in (1, 1024):
    o = O.OraclePatch(48000, B, 2); ids = S.build_p2(o)
    ref, _ = o.render_batch(V, T, [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)], threads=8)
    for flags in (0, 2, 1):
        p = S.Patch(48000, B, 2); S.build_p2(p); p.configure_voices(V)
        p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta); p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
        fr = p.render_channels(T, flags)
        err = np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
        per_t = err.max(axis=(0, 2))
        marks = [int(np.argmax(per_t > th)) if (per_t > th).any() else -1 for th in (1e-6, 1e-5, 1e-4)]
        print(f"B={B} flags={flags}: max err {err.max():.3e}; err at 0.25 s {per_t[:12000].max():.3e}, 0.5 s {per_t[:24000].max():.3e}; first t > 1e-6/1e-5/1e-4: {marks}; {p.info()[-40:]}")
