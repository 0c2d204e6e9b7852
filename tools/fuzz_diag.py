"""Why is a fuzzed patch not bit-identical in exact mode?  usage: python tools/fuzz_diag.py [noise] seed..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
O.build()
args = sys.argv[1:]
noise = args and args[0] == "noise"
if noise: args = args[1:]
NAMES = {0: "OUT", 1: "OSC", 2: "VCF", 3: "ADSR", 4: "VCA", 5: "MIX", 6: "MATH", 7: "GRID", 8: "PAT", 9: "NL", 10: "SMP", 11: "NOISE", 12: "VERB"}
for seed in map(int, args):
    B, build, overrides = random_patch(seed, noise)
    V, T = (67, 1300) if B < 1024 else (131, 2300)
    o = O.OraclePatch(48000, B, 2)
    ids = build(o)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    p = S.Patch(48000, B, 2)
    build(p)
    p.configure_voices(V)
    for m, f, vals in ov: p.set_voice_field(m, f, vals)
    fr = p.render_channels(T, 3)
    d = np.abs(fr.astype(np.float64) - ref.astype(np.float64))
    rel = d / np.maximum(np.abs(ref), 1.0)
    neq = fr.view(np.uint32) != ref.view(np.uint32)
    first = int(np.argwhere(neq.any(axis=(0, 2)))[0]) if neq.any() else -1
    voices = np.flatnonzero(neq.any(axis=(0, 1)))
    n = p.num_modules() if hasattr(p, "num_modules") else len(ids) + 1
    types = [NAMES[p.module_type(m)] for m in range(n)]
    wires = []
    for m in range(n):
        for k in range(p.get_num_inputs(m)):
            src = p.get_input(m, k)
            if src is not None: wires.append(f"{types[src[0]]}{src[0]}.{src[1]}->{types[m]}{m}.{k}")
    print(f"seed {seed} B={B}: {neq.mean():.5f} differ, max abs {d.max():.3e}, max rel {rel.max():.3e}, first sample {first}, voices {len(voices)}/{V} (e.g. {voices[:5]}), nonfinite ref {(~np.isfinite(ref)).sum()}")
    print("   plan", [f"{types[m]}{m}" for m in p.plan()], "overrides", [(types[m] + str(m), f) for m, f, _ in ov])
    print("   wires", " ".join(wires))
    if (~np.isfinite(ref)).any():
        nan_r, nan_g = np.isnan(ref), np.isnan(fr)
        both = ~nan_r & ~nan_g
        print(f"   NaN masks equal: {bool((nan_r == nan_g).all())}; non-NaN samples differing: {int((fr.view(np.uint32)[both] != ref.view(np.uint32)[both]).sum())} of {int(both.sum())};"
              f" NaN patterns ref {sorted(set(hex(x) for x in ref.view(np.uint32)[nan_r][:2000]))[:4]} gpu {sorted(set(hex(x) for x in fr.view(np.uint32)[nan_g][:2000]))[:4]}")
        idx = np.argwhere(both & (fr.view(np.uint32) != ref.view(np.uint32)))
        idx = idx[np.argsort(idx[:, 1], kind="stable")][:6]
        for c_, t_, v_ in idx:
            print(f"      ch {c_} sample {t_} voice {v_}: gpu {fr[c_, t_, v_]!r} ref {ref[c_, t_, v_]!r}; ref around {ref[c_, max(t_-2,0):t_+2, v_]}")
