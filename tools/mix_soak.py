"""Soak of the planes' shared mix tile (wave.hip.h: emit_put_rows / emit_rows_flush) over random patches: the mix of every channel
through the specialised kernel, default modes, bit for bit against the same kernel generated with a tile per plane (SRACK_TILE_PER_PLANE=1,
read when a kernel is generated) and within 1e-5 sum|terms| of the f64 sum of the frames.  usage: <first> <last>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad, n, shared, t0 = [], 0, 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed)
    rng = np.random.default_rng(9000 + seed)
    V = int(rng.choice([70, 64, 130]))
    T = int(rng.choice([1003, 257, 2048 + 17]))
    values = [(m, f, fn(V)) for m, f, fn in overrides]
    for flags in (34, 38):
        outs = []
        for per_plane in (False, True):
            if per_plane:
                os.environ["SRACK_TILE_PER_PLANE"] = "1"
            else:
                os.environ.pop("SRACK_TILE_PER_PLANE", None)
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(V)
            for m, f, vals in values:
                p.set_voice_field(ids[m], f, vals)
            try:
                src = p.kernel_source(flags)
            except S.SrackError:
                outs = None
                break
            if not per_plane:
                shared += "emit_rows_flush" in src
            outs.append(p.render(T, frames=True, mix=True, flags=flags) + (p.planes()[1],))
        os.environ.pop("SRACK_TILE_PER_PLANE", None)
        if outs is None:
            continue
        (fr, mx, cp), (fr1, mx1, _) = outs
        n += 1
        same = ((mx.view(np.uint32) == mx1.view(np.uint32)) | (np.isnan(mx) & np.isnan(mx1))).all() and ((fr.view(np.uint32) == fr1.view(np.uint32)) | (np.isnan(fr) & np.isnan(fr1))).all()
        ok = True
        for c, pl in enumerate(cp):
            if pl < 0:
                continue
            f64 = fr[pl].astype(np.float64)
            fin = np.isfinite(f64).all(axis=1) & (np.abs(f64).sum(axis=1) < 3.0e38)  # (an exploding patch: finite frames whose f32 sum is not — seed 370)
            own, scale = f64.sum(axis=1), np.abs(f64).sum(axis=1)
            ok = ok and bool((np.abs(mx[c][fin] - own[fin]) <= 1e-5 * np.maximum(scale[fin], 1.0)).all())
        if not (same and ok):
            bad.append((seed, flags, V, T, same, ok))
print(f"mix soak, seeds {lo}..{hi - 1}: {n} comparisons ({shared} kernels with a shared tile), {len(bad)} bad, {time.time() - t0:.0f} s")
for b in bad[:20]:
    print("  seed %d flags %d V %d T %d: identical to tile-per-plane %s, within tolerance of the f64 sums %s" % b)
