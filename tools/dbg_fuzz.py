"""Debug helper: one fuzz seed under two render modes, first differences (python tools/dbg_fuzz.py SEED FLAGS_A FLAGS_B [noise])."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
S = srack_pkg.load()
from tests.fuzz_patches import random_patch
seed, fa, fb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
noise = len(sys.argv) > 4
B, build, overrides = random_patch(seed, noise)
V, T = (67, 1300) if B < 1024 else (131, 2300)
vals = None
outs = []
for flags in (fa, fb):
    p = S.Patch(48000, B, 2)
    ids = build(p)
    p.configure_voices(V)
    if vals is None:
        vals = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    for m, f, v in vals:
        p.set_voice_field(m, f, v)
    outs.append(p.render_channels(T, flags))
    print(flags, p.info())
a, b = outs
d = a.view(np.uint32) != b.view(np.uint32)
print("differ:", d.sum(), "of", d.size, "per channel", d.sum(axis=(1, 2)))
idx = np.argwhere(d)[:10]
for c, t, v in idx:
    print(c, t, v, a[c, t, v], b[c, t, v], hex(a.view(np.uint32)[c, t, v]), hex(b.view(np.uint32)[c, t, v]))
