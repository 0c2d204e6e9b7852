"""Sample-player soak: random waves (length 0 ... 5000), wave / audio sample rates, gate clocks and pitch CVs (constant, sine LFO, noise);
exact modes bit for bit, default modes: fraction of samples outside 1e-5 (a read index that truncates the other way is a different
sample).  usage: <first> <last>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
S = srack_pkg.load()
O.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, slips, n, t0 = [], [], 0, time.time()
for seed in range(lo, hi):
    r = np.random.default_rng((seed, 0x5D))
    sr = int(r.choice([8000, 22050, 44100, 48000, 65535])); B = int(r.choice([1, 16, 64, 1024])); V = int(r.choice([1, 64, 70])); T = int(r.choice([900, 4000]))
    wave = r.uniform(-1, 1, int(r.choice([0, 1, 2, 100, 1000, 5000]))).astype(np.float32); wsr = float(r.choice([8000.0, 44100.0, 96000.0]))
    clock = float(np.float32(r.uniform(-7, -1))); cvk = int(r.integers(0, 4)); lfo = float(np.float32(r.uniform(-8, -3))); depth = float(np.float32(r.uniform(0, 1.5)))
    expo0 = float(np.float32(r.uniform(0.3, 3)))
    def build(g):
        clk, src, k, smp, nl, out = (g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_NOISE if cvk == 3 else S.MOD_OSCILLATOR), g.add_module(S.MOD_MATH),
                                     g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_NONLINEAR), g.add_module(S.MOD_OUTPUT))
        g.set_field(clk, S.OSC_VAL, clock); g.connect(clk, S.OSC_OUT_SQUARE, smp, 0)
        g.set_field(k, S.MATH_OPERATION, S.MATH_MULTIPLY); g.set_field(k, S.MATH_CONSTANT, depth)
        if cvk == 1: g.connect(k, 0, smp, 1)                                   # constant 0 * depth
        if cvk in (2, 3):
            if cvk == 2: g.set_field(src, S.OSC_VAL, lfo)
            g.connect(src, 0, k, 0); g.connect(k, 0, smp, 1)                    # sine LFO or noise, scaled
        g.set_wave(smp, wave, wsr)
        g.connect(smp, 0, nl, 0); g.set_field(nl, S.NONLIN_CONSTANT, expo0)
        g.connect(nl, 0, out, 0); g.connect(smp, 0, out, 1)
        g.set_noise_seed(seed, 3)
        return dict(k=k, nl=nl)
    dep = r.uniform(0, 1.5, V).astype(np.float32); ex = r.uniform(0.3, 3, V).astype(np.float32)
    o = O.OraclePatch(sr, B, 2); ids = build(o)
    ov = [(ids["k"], S.MATH_CONSTANT, dep), (ids["nl"], S.NONLIN_CONSTANT, ex)]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    for flags in (1, 3, 0, 2):
        p = S.Patch(sr, B, 2); build(p); p.configure_voices(V)
        for m, f, vals in ov: p.set_voice_field(m, f, vals)
        fr = p.render_channels(T, flags)
        n += 1
        raw_same = ((fr[1].view(np.uint32) == ref[1].view(np.uint32)) | (np.isnan(fr[1]) & np.isnan(ref[1])))
        err = np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)
        if flags & 1:
            if not raw_same.all() or np.nanmax(err[0]) > 1e-5: bad.append((seed, flags, sr, B, V, T, len(wave), wsr, cvk, float(1 - raw_same.mean()), float(np.nanmax(err))))
        else:
            slips.append(float((err > 1e-5).mean()))
            if slips[-1] > 0.01: bad.append((seed, flags, sr, B, V, T, len(wave), wsr, cvk, float(1 - raw_same.mean()), float(np.nanmax(err))))
print(f"sample-player soak, seeds {lo}..{hi - 1}: {n} renders, {len(bad)} fail, default-mode samples outside 1e-5: mean {np.mean(slips):.2e} max {np.max(slips):.2e}, {time.time() - t0:.0f} s")
for b in bad[:20]: print("  ", b)
