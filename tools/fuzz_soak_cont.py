"""Soak of the continuity property past the pinned seeds: render(T) == render(a) ++ render(b) ++ render(c), bit for bit, in
default and exact modes, whatever kernels / tiles / chunks / control units the patch got.  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, t0 = [], 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    rng = np.random.default_rng(1000 + seed)
    V, T = 70, 2600 if seed % 5 else 9000   # (every fifth patch crosses 4096-sample launch borders)
    cuts = sorted(int(c) for c in rng.choice(np.arange(1, T), size=2, replace=False))
    values = [(m, f, fn(V)) for m, f, fn in overrides]
    for flags in ((34, 38, 35) if os.environ.get("FUZZ_SPECIAL") else (0, 2, 4, 1)):
        if flags & 32:
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(V)
            for m, f, vals in values:
                p.set_voice_field(ids[m], f, vals)
            try:
                p.kernel_source(flags)
            except S.SrackError:
                continue
        outs = []
        for parts in ([T], [cuts[0], cuts[1] - cuts[0], T - cuts[1]]):
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(V)
            for m, f, vals in values:
                p.set_voice_field(ids[m], f, vals)
            outs.append(np.concatenate([p.render_channels(k, flags) for k in parts], axis=1))
        n += 1
        same = (outs[0].view(np.uint32) == outs[1].view(np.uint32)) | (np.isnan(outs[0]) & np.isnan(outs[1]))
        if not same.all():
            bad.append((seed, flags, cuts, float(1 - same.mean())))
print(f"continuity, seeds {lo}..{hi - 1} noise={noise}: {n} comparisons, {len(bad)} differ, {time.time() - t0:.0f} s")
for b in bad[:30]:
    print("  seed %d flags %d cuts %s: %.5f of the samples differ" % b)
