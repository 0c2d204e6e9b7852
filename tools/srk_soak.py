"""Rack-file soak: random patch -> save_srk -> load_srk (module order reversed, ui.rs:654-660) -> render on the GPU, against the
oracle rebuilt from what the LOADED patch reports through the graph API (types, fields, steps, waves, wiring).  Exact modes, bit for
bit.  Per-voice overrides are re-applied to the loaded patch by module id.  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
O.build()
NF = {0: 0, 1: 4, 2: 13, 3: 10, 4: 1, 5: 4, 6: 2, 7: 7, 8: 4, 9: 1, 10: 6, 11: 0, 12: 6}   # fields per module type
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, t0 = [], 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    V, T = 40, 1300 if B < 1024 else 2300
    p = S.Patch(48000, B, 2)
    ids = build(p)
    names = {p.module_id(m): m for m in range(p.num_modules())}
    q = S.Patch.load_srk(p.save_srk(), 48000, B, 2)
    if noise: q.set_noise_seed(seed * 7919 + 1, 1000 * seed)      # (the seed is not part of the file)
    new_of = {names[q.module_id(m)]: m for m in range(q.num_modules())}   # index in p -> index in q
    # mirror q into an oracle patch
    o = O.OraclePatch(48000, B, 2)
    for m in range(q.num_modules()):
        t = q.module_type(m)
        assert o.add_module(t) == m
        for f in range(NF[t]):
            o.set_field(m, f, q.get_field(m, f))
        if t in (S.MOD_GRID_SEQUENCER, S.MOD_PATTERN_SEQUENCER):
            for ch in ([0] if t == S.MOD_GRID_SEQUENCER else range(8)):
                for i in range(64):
                    st, val = q.get_step(m, ch, i)
                    if st: o.set_step(m, ch, i, st, val)
        if t == S.MOD_SAMPLE:
            w, rate = q.get_wave(m)
            newflag = q.get_field(m, S.SAMPLE_WAVE_NEW)
            if len(w) or rate: o.set_wave(m, w, rate)
            o.set_field(m, S.SAMPLE_WAVE_NEW, newflag)
    for m in range(q.num_modules()):
        for k in range(q.get_num_inputs(m)):
            src = q.get_input(m, k)
            if src is not None: o.connect(src[0], src[1], m, k)
    if noise: o.set_noise_seed(seed * 7919 + 1, 1000 * seed)
    ov = [(new_of[ids[m]], f, fn(V)) for m, f, fn in overrides]
    ref, _ = o.render_batch(V, T, ov, threads=8)
    q.configure_voices(V)
    for m, f, vals in ov: q.set_voice_field(m, f, vals)
    assert q.plan() == o.plan(), seed
    for flags in (1, 3):
        fr = q.render_channels(T, flags)
        n += 1
        same = (fr.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(fr) & np.isnan(ref))
        if not same.all(): bad.append((seed, flags, float(1 - same.mean())))
        if flags == 1: q = S.Patch.load_srk(p.save_srk(), 48000, B, 2); q.configure_voices(V); [q.set_voice_field(m, f, vals) for m, f, vals in ov]; (q.set_noise_seed(seed * 7919 + 1, 1000 * seed) if noise else None)
print(f"rack-file soak, seeds {lo}..{hi - 1} noise={noise}: {n} renders, {len(bad)} differ, {time.time() - t0:.0f} s")
for b in bad[:30]: print("  seed %d flags %d: %.5f of the samples differ" % b)
