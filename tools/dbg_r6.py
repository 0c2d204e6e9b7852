import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tools')
import numpy as np, srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load(); O.build()
for seed, noise in ((900146, False), (900023, True)):
    B, build, overrides = random_patch(seed, noise)
    V, T = 64, 3000
    o = O.OraclePatch(48000, B, 2)
    ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    p = S.Patch(48000, B, 2); build(p); p.configure_voices(V)
    for m, f, vals in ov: p.set_voice_field(m, f, vals)
    fr = p.render_channels(T, 0)
    e = np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
    print(seed, p.info())
    print(' per channel max', e.max(axis=(1, 2)))
    c = int(np.argmax(e.max(axis=(1, 2))))
    pv = e[c].max(axis=0)
    order = np.argsort(-pv)[:6]
    for v in order:
        t = int(np.argmax(e[c][:, v]))
        print('  voice', v, 'max', pv[v], 'at t', t, 'ov', [(m, f, float(vals[v])) for m, f, vals in ov], 'ref', ref[c, max(t-2,0):t+3, v], 'gpu', fr[c, max(t-2,0):t+3, v], 'n differing', int((e[c][:, v] > 0).sum()))
