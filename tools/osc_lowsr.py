import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, srack_pkg
from oracle import oracle as O
S = srack_pkg.load(); O.build()
for sr in (1000, 2000, 8000):
    V, T = 64, 3000
    val = np.linspace(-3, 1.2, V).astype(np.float32)
    def build(g):
        osc, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_OUTPUT)
        g.connect(osc, S.OSC_OUT_SAW, out, 0); g.connect(osc, S.OSC_OUT_SQUARE, out, 1)
        return osc
    o = O.OraclePatch(sr, 64, 2); osc = build(o)
    ref, _ = o.render_batch(V, T, [(osc, S.OSC_VAL, val)], threads=8)
    for flags in (0, 2, 1):
        p = S.Patch(sr, 64, 2); build(p); p.configure_voices(V); p.set_voice_field(osc, S.OSC_VAL, val)
        fr = p.render_channels(T, flags)
        err = np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
        ev = err.max(axis=(0, 1))
        badv = np.flatnonzero(ev > 1e-5)
        delta = 440.0 * 2.0 ** val.astype(np.float64) / sr
        print(f"sr {sr} flags {flags}: max err {err.max():.3e}; voices outside: {len(badv)}; their delta range {delta[badv].min() if len(badv) else 0:.3f}..{delta[badv].max() if len(badv) else 0:.3f}; saw err {err[0].max():.2e} square err {err[1].max():.2e}  {p.info()[-60:]}")
