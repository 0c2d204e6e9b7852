"""Soak of srack_patch_keep_state on random patches: render(T) against render(a), edit, render(b), edit, render(c) where the edit
re-flattens the patch but changes nothing audible.  Exact modes must agree bit for bit (module state, feedback rings, reverb lines and
the sample counter all carried); default modes are reported (a fixed-point phase passes through a double).  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, soft, n, t0 = [], 0, 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    rng = np.random.default_rng(5000 + seed)
    V, T = 70, 2600
    cuts = sorted(int(c) for c in rng.choice(np.arange(1, T), size=2, replace=False))
    values = [(m, f, fn(V)) for m, f, fn in overrides]
    for flags in (1, 3, 0, 2):
        outs = []
        for edit in (False, True):
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(V)
            for m, f, vals in values:
                p.set_voice_field(ids[m], f, vals)
            if not edit:
                outs.append(p.render_channels(T, flags))
                continue
            p.keep_state(True)
            osc = next((m for m in range(p.num_modules()) if p.module_type(m) == S.MOD_OSCILLATOR), None)
            if osc is None: break
            parts = []
            # FUZZ_SPECIAL=1 (first patch family only: no reverbs): specialised kernels alternate with interpreter / fused ones
            exact_modes = (35, 3, 39, 7, 43, 1) if os.environ.get("FUZZ_SPECIAL") else (1, 3, 5, 7, 9, 11)
            for k in (cuts[0], cuts[1] - cuts[0], T - cuts[1]):
                # exact modes: each part in ANOTHER exact mode (fused / interpreter / per-voice / one control unit): a flags change is an edit too
                parts.append(p.render_channels(k, exact_modes[(seed + len(parts) + flags) % 6] if flags & 1 else flags))
                kind = (seed + len(parts)) % 3
                if kind == 0:
                    p.set_field(osc, S.OSC_ANTIALIASING, p.get_field(osc, S.OSC_ANTIALIASING))   # same value: a re-flatten and nothing else
                elif kind == 1:
                    p.add_module(S.MOD_MATH)                                                     # a module nothing reads
                else:                                                                            # a wire pulled and put back
                    wired = [(m, k, p.get_input(m, k)) for m in range(p.num_modules()) for k in range(p.get_num_inputs(m)) if p.get_input(m, k) is not None]
                    m, k, src = wired[(seed * 7) % len(wired)]
                    p.disconnect(m, k)
                    p.connect(src[0], src[1], m, k)
            outs.append(np.concatenate(parts, axis=1))
        if len(outs) < 2: continue
        n += 1
        same = (outs[0].view(np.uint32) == outs[1].view(np.uint32)) | (np.isnan(outs[0]) & np.isnan(outs[1]))
        if flags & 1:
            if not same.all(): bad.append((seed, flags, cuts, float(1 - same.mean()), sorted(set(p.module_type(m) for m in range(p.num_modules()))), B))
        else:
            err = np.abs(outs[0].astype(np.float64) - outs[1]) / np.maximum(np.abs(outs[1]), 1.0)
            if np.nanmax(err) > 1e-5: soft += 1
print(f"keep_state, seeds {lo}..{hi - 1} noise={noise}: {n} comparisons, {len(bad)} exact-mode differences, {soft} default-mode renders beyond 1e-5, {time.time() - t0:.0f} s")
for b in [x for x in bad if x[1] == 1][:40]:
    print("  seed %d flags %d cuts %s: %.5f of the samples differ; module types %s B=%d" % b)
