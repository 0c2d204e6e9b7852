"""Soak of tick sessions (render.hip, TickSession) over random patches: a render driven in equal blocks — with a state read-back and a
change of block length on the way — against the same render in one call, bit for bit, through the specialised kernels (the ones that
carry the control program's units, i.e. the ones that open a session).  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, sessions, t0 = [], 0, 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    rng = np.random.default_rng(7000 + seed)
    V = 70
    L = int(rng.choice([64, 200, 256, 1000, 1024]))
    L2 = int(rng.choice([32, 96, 512]))
    script = [L] * int(rng.integers(3, 9)) + ["read"] + [L] * 2 + [L2] * 3 + [L] * 2
    T = sum(x for x in script if x != "read")
    values = [(m, f, fn(V)) for m, f, fn in overrides]
    for flags in (34, 38, 35):
        def fresh():
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(V)
            for m, f, vals in values:
                p.set_voice_field(ids[m], f, vals)
            return p, ids
        p, ids = fresh()
        try:
            p.kernel_source(flags)
        except S.SrackError:
            continue
        whole = p.render_channels(T, flags)
        p, ids = fresh()
        parts = []
        for x in script:
            if x == "read":
                for m in range(p.num_modules()):
                    if p.module_type(m) == S.MOD_OSCILLATOR:
                        p.get_voice_field(m, S.OSC_POS)
            else:
                parts.append(p.render_channels(x, flags))
        info = p.info()
        sessions += ("tracks=" in info and "kernel=render_specialized" in info)
        got = np.concatenate(parts, axis=1)
        n += 1
        same = (got.view(np.uint32) == whole.view(np.uint32)) | (np.isnan(got) & np.isnan(whole))
        if not same.all():
            bad.append((seed, flags, L, float(1 - same.mean()), info[-60:]))
print(f"tick sessions, seeds {lo}..{hi - 1} noise={noise}: {n} comparisons ({sessions} with a control program on the voice launches), {len(bad)} differ, {time.time() - t0:.0f} s")
for b in bad[:30]:
    print("  seed %d flags %d L %d: %.5f of the samples differ  (%s)" % b)
