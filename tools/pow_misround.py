"""tools/pow_misround.py SEED... — where does the host libm's pow(2, e) differ from the correctly rounded 2^e along a fuzz patch's render?
(CPU only: the oracle taps every connected pitch CV, voice by voice; mpmath decides the correct rounding.)  The exact render mode
evaluates 2^cv correctly rounded (modules.hip.h, exp2_cr); the reference evaluates `2.0_f64.powf(e)` with the host's libm, which is within
0.52 ulp, i.e. NOT always the correctly rounded double.  A voice whose render meets such an argument has an increment one ulp off the
device's; in a patch that iterates its phases (a loop through a pitch or sync input) that last bit grows into different samples."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, mpmath
from oracle import oracle as O
from tests.fuzz_patches import random_patch, W
mpmath.mp.prec = 200


def exp2_correct(e):
    return float(mpmath.power(2, mpmath.mpf(e)))   # mpf(e) is exact; the conversion to float rounds to nearest


def misrounded_samples(seed, noise=False, voices=None):
    """-> {voice: [(oscillator module, first sample with a misrounded 2^e, e)]} over the fuzz test's render of this seed"""
    B, build, overrides = random_patch(seed, noise)
    V, T = (67, 1300) if B < 1024 else (131, 2300)
    O.build()
    base = O.OraclePatch(48000, B, 2)
    ids = build(base)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    n = base.num_modules()
    # oscillators with a connected pitch CV: the product's graph accessors give the wiring (the oracle has none)
    import srack_pkg
    S = srack_pkg.load()
    p = S.Patch(48000, B, 2)
    build(p)
    oscs = [(m, p.get_input(m, 0)) for m in range(n) if p.module_type(m) == S.MOD_OSCILLATOR and p.get_input(m, 0) is not None]
    out = {}
    for v in (range(V) if voices is None else voices):
        for m, (src, port) in oscs:
            g = O.OraclePatch(48000, B, 2)
            build(g)
            for mod, f, vals in ov:
                g.set_field(mod, f, float(vals[v]))
            val = np.float64(np.float32(g.get_field(m, W.OSC_VAL)))
            _, cv = g.render(T, tap=(src, port))
            e = cv.astype(np.float64) + val
            uniq, first = np.unique(e, return_index=True)
            for x, t in sorted(zip(uniq, first), key=lambda q: q[1]):
                if not np.isfinite(x) or abs(x) > 1000:
                    continue
                if math.pow(2.0, float(x)) != exp2_correct(float(x)):
                    out.setdefault(v, []).append((m, int(t), float(x)))
                    break
    return out, V, T


if __name__ == "__main__":
    for seed in [int(a) for a in sys.argv[1:]]:
        res, V, T = misrounded_samples(seed)
        print(f"seed {seed}: {len(res)} of {V} voices meet an argument where the libm's pow(2, e) is not the correctly rounded 2^e within {T} samples")
        for v, hits in sorted(res.items()):
            print("   voice", v, " ".join(f"[osc {m}: sample {t}, e = {e!r}]" for m, t, e in hits))
