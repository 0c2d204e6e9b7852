"""Soak of the differential fuzzer over the CONFIGURATION space the pinned tests fix: sample rate, buffer size, channel count, voice
count and render length are drawn per seed (exact modes, bit for bit against the oracle; SOAK_DEFAULT=1: the default modes at the 1e-5 bar).  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
O.build()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, t0 = [], 0, time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng((seed, 0xC0F))
    _, build, overrides = random_patch(seed, noise)
    B = int(rng.choice([1, 2, 5, 17, 31, 32, 33, 100, 256, 333, 1024, 2048, 4096]))
    sr = int(rng.choice([8000, 22050, 44100, 48000, 65535]))
    C = int(rng.choice([2, 3, 5, 8]))
    V = int(rng.choice([1, 2, 63, 64, 65, 128, 129, 200]))
    T = int(rng.choice([1, 2, 31, 32, 33, 500, 1023, 1024, 1025, 3000, 4095, 4096, 4097, 5000]))
    os.environ["SRACK_WANT_WAVES"] = "1" if seed % 2 else "0"
    o = O.OraclePatch(sr, B, C)
    ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    DEFAULT = bool(os.environ.get("SOAK_DEFAULT"))  # the default modes over the same configurations, at the 1e-5 bar
    for flags in ((0, 2, 4) if DEFAULT else (1, 3, 7, 11)):
        p = S.Patch(sr, B, C)
        build(p)
        p.configure_voices(V)
        for m, f, vals in ov:
            p.set_voice_field(m, f, vals)
        fr = p.render_channels(T, flags)
        n += 1
        same = (fr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(fr) & np.isnan(ref))
        if DEFAULT and fr.shape == ref.shape:
            r64, f64 = ref.astype(np.float64), fr.astype(np.float64)
            fin = np.isfinite(r64) & np.isfinite(f64)
            with np.errstate(invalid="ignore"):
                same = np.where(fin, np.abs(f64 - r64) <= 1e-5 * np.maximum(np.abs(r64), 1.0), (np.isnan(fr) & np.isnan(ref)) | (fr == ref))
        if fr.shape != ref.shape or not same.all():
            bad.append((seed, flags, (sr, B, C, V, T), float(1 - same.mean())))
print(f"configs, seeds {lo}..{hi - 1} noise={noise}: {n} renders, {len(bad)} {'outside the 1e-5 band' if os.environ.get('SOAK_DEFAULT') else 'not bit-identical'}, {time.time() - t0:.0f} s")
for b in bad[:40]:
    print("  seed %d flags %d (sr, B, C, V, T) = %s: %.5f of the samples differ" % b)
