import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
S = srack_pkg.load(); O.build()
for seed in [int(x) for x in sys.argv[1:]]:
    r = np.random.default_rng((seed, 0xF6))
    sr = int(r.choice([8000, 22050, 44100, 48000, 65535]))
    B = 1 if r.random() < 0.4 else int(r.choice([256, 512, 1000, 1024, int(r.integers(256, 1025))]))
    V = int(r.choice([1, 31, 32, 33, 64, 70, 130, 200]))
    blocks = int(r.integers(5, 10))
    T1 = blocks * max(B, 1) if B > 1 else int(r.integers(2000, 6000))
    if B > 1 and T1 < 4096: T1 = ((4096 + B - 1) // B) * B
    if r.random() < 0.3: T1 += int(r.integers(1, 64))
    T2 = int(r.choice([0, 100, 1024, 4096, 5000]))
    bet = r.uniform(0, 0.7, V).astype(np.float32); idx = r.uniform(0, 2.5, V).astype(np.float32)
    if r.random() < 0.2: bet[int(r.integers(0, V))] = 3.0e4
    if r.random() < 0.2: idx[int(r.integers(0, V))] = float(r.uniform(3, 12))
    vm = (r.uniform(-2, 2, V) if r.random() < 0.5 else np.zeros(V)).astype(np.float32)
    vcr = r.uniform(-2, 2, V).astype(np.float32)
    pm, pc = r.uniform(0, 1, V), r.uniform(0, 1, V)
    o = O.OraclePatch(sr, B, 2); ids = S.build_p2(o)
    ov = [(ids["mul_fb"], S.MATH_CONSTANT, bet), (ids["mul_idx"], S.MATH_CONSTANT, idx), (ids["osc_m"], S.OSC_VAL, vm), (ids["osc_c"], S.OSC_VAL, vcr), (ids["osc_m"], S.OSC_POS, pm), (ids["osc_c"], S.OSC_POS, pc)]
    ref, _ = o.render_batch(V, T1, ov, threads=8)
    for flags in (0, 2):
        p = S.Patch(sr, B, 2); S.build_p2(p); p.configure_voices(V)
        for m, f, vals in ov: p.set_voice_field(m, f, vals)
        a = p.render_channels(T1, flags)[0]
        mism = np.isnan(a) != np.isnan(ref[0])
        vs = np.where(mism.any(axis=0))[0]
        print(seed, 'flags', flags, 'sr', sr, 'B', B, 'V', V, 'T1', T1, p.info().split('kernel=')[-1], 'voices with NaN mismatch', vs[:10], 'wild beta voices', np.where(bet > 100)[0], 'big idx', np.where(idx > 2.9)[0])
        for v in vs[:3]:
            t = int(np.argmax(mism[:, v]))
            print('   voice', v, 'first mismatch t', t, 'ref', ref[0][max(0,t-2):t+3, v], 'gpu', a[max(0,t-2):t+3, v], 'beta', bet[v], 'idx', idx[v], 'vm', vm[v], 'vc', vcr[v], 'gpu nan count', int(np.isnan(a[:, v]).sum()), 'ref nan count', int(np.isnan(ref[0][:, v]).sum()))
