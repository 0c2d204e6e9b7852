"""Debug helper: default-mode error of given fuzz seeds, per render flag, with the program description (python tools/dbg_default.py [noise] seed...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load(); O.build()
NOISE = len(sys.argv) > 1 and sys.argv[1] == "noise"
for seed in [int(a) for a in sys.argv[1 + NOISE:]]:
    B, build, overrides = random_patch(seed, NOISE)
    V, T = (67, 1300) if B < 1024 else (131, 2300)
    if os.environ.get("SOAK_VT"):
        V, T = (int(x) for x in os.environ["SOAK_VT"].split(","))
    o = O.OraclePatch(48000, B, 2)
    ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    r64 = ref.astype(np.float64)
    for flags in ([int(x) for x in os.environ["DBG_FLAGS"].split(",")] if os.environ.get("DBG_FLAGS") else (0, 1, 18)):
        p = S.Patch(48000, B, 2)
        build(p)
        p.configure_voices(V)
        for m, f, vals in ov: p.set_voice_field(m, f, vals)
        fr = p.render_channels(T, flags)
        ok = np.isfinite(r64) & np.isfinite(fr)
        err = np.abs(fr.astype(np.float64)[ok] - r64[ok]) / np.maximum(np.abs(r64[ok]), 1.0)
        bad = np.argwhere((np.abs(fr.astype(np.float64) - r64) / np.maximum(np.abs(r64), 1.0)) > 1e-5)
        print(f"seed {seed} flags {flags}: max {err.max():.2e} outside {float((err>1e-5).mean()):.5f} first bad (c,t,v) {bad[0] if len(bad) else None} voices {len(set(bad[:,2])) if len(bad) else 0}")
        if os.environ.get("DBG_VOICES") and len(bad):
            for v in sorted(set(bad[:, 2]))[:8]:
                b = bad[bad[:, 2] == v]
                print(f"      voice {v}: {len(b)} bad, channels {sorted(set(b[:,0]))}, first t {b[:,1].min()} last t {b[:,1].max()}; ref/gpu around the first: {ref[b[0][0], max(b[0][1]-2,0):b[0][1]+3, v]} {fr[b[0][0], max(b[0][1]-2,0):b[0][1]+3, v]}")
    print("   ", p.info()[:200])
    print("    types", [p.module_type(m) for m in range(p.num_modules())], "B", B)
