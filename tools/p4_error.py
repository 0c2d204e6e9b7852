import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, srack_pkg
from oracle import oracle as O
S = srack_pkg.load(); O.build()
V, T = 256, 12000
depth, expo = S.p4_voice_params(V)
o = O.OraclePatch(48000, 1024, 2); ids = S.build_p4(o)
ref, _ = o.render_batch(V, T, [(ids["depth"], S.MATH_CONSTANT, depth), (ids["shaper"], S.NONLIN_CONSTANT, expo)], threads=8)
for flags in (32, 0):
    p = S.Patch(48000, 1024, 2); S.build_p4(p); p.configure_voices(V)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth); p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
    got = p.render_channels(T, flags)
    for c in range(2):
        err = np.abs(got[c].astype(np.float64) - ref[c]) / np.maximum(np.abs(ref[c]), 1.0)
        print("flags", flags, "channel", c, "max rel err %.3e" % err.max(), "differing bits %.4f" % (got[c].view(np.uint32) != ref[c].view(np.uint32)).mean(), p.info()[-40:])
