"""Reverb soak: random FreeverbModule parameters, sample rates, buffer sizes and input wirings, long renders; exact modes bit for bit,
default modes within 1e-5.  usage: <first> <last>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
S = srack_pkg.load()
O.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, n, t0 = [], 0, time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng((seed, 0xFE))
    sr = int(rng.choice([1000, 8000, 22050, 44100, 48000, 65535]))
    B = int(rng.choice([1, 7, 64, 1024]))
    V, T = int(rng.choice([1, 20, 70])), int(rng.choice([3000, 12000]))
    params = [(S.FREEVERB_DAMPENING, rng.uniform(0, 2)), (S.FREEVERB_WET, rng.uniform(0, 1)), (S.FREEVERB_WIDTH, rng.uniform(0, 1)),
              (S.FREEVERB_ROOM_SIZE, rng.uniform(0, 1)), (S.FREEVERB_DRY, rng.uniform(0, 1)), (S.FREEVERB_FREEZE, int(rng.random() < 0.25))]
    wiring = int(rng.integers(0, 4))  # which inputs are fed, and by what
    det = rng.uniform(-3, 1, V).astype(np.float32)
    def build(g):
        osc, nz, fv, fv2, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_NOISE), g.add_module(S.MOD_FREEVERB), g.add_module(S.MOD_FREEVERB), g.add_module(S.MOD_OUTPUT)
        if wiring in (0, 2): g.connect(osc, S.OSC_OUT_SAW, fv, 0)
        if wiring in (1, 2): g.connect(nz, 0, fv, 1)
        if wiring == 3:
            g.connect(osc, S.OSC_OUT_SQUARE, fv, 0); g.connect(osc, S.OSC_OUT_SAW, fv, 1)
        g.connect(fv, 0, fv2, 1); g.connect(fv, 1, fv2, 0)   # a second reverb in series, channels crossed
        g.connect(fv2, 0, out, 0); g.connect(fv, 1, out, 1)
        for f, v in params: g.set_field(fv, f, v)
        g.set_field(fv2, S.FREEVERB_ROOM_SIZE, 0.2)
        g.set_noise_seed(seed, 100)
        return osc
    o = O.OraclePatch(sr, B, 2)
    osc = build(o)
    ref, _ = o.render_batch(V, T, [(osc, S.OSC_VAL, det)], threads=8)
    for flags in (1, 3, 0, 2):
        p = S.Patch(sr, B, 2)
        build(p)
        p.configure_voices(V)
        p.set_voice_field(osc, S.OSC_VAL, det)
        fr = p.render_channels(T, flags)
        n += 1
        if flags & 1:
            ok = ((fr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(fr) & np.isnan(ref))).all()
        else:
            ok = bool((np.abs(fr.astype(np.float64) - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0)).all())
        if not ok: bad.append((seed, flags, sr, B, V, T, wiring, float((np.abs(fr.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)).max())))
print(f"reverb soak, seeds {lo}..{hi - 1}: {n} renders, {len(bad)} fail, {time.time() - t0:.0f} s")
for b in bad[:20]: print("  ", b)
