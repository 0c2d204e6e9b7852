"""State read-back soak: after a render, every state field of every module, read per voice with srack_voices_get_field, against the
state the oracle's patch objects hold after the same number of ticks (exact modes; f64 phases bit for bit).  usage: <first> <last> [noise]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import srack_pkg
from oracle import oracle as O
from tests.fuzz_patches import random_patch
S = srack_pkg.load()
O.build()
STATE = {S.MOD_OSCILLATOR: [2, 3], S.MOD_MOOG_FILTER: list(range(3, 13)), S.MOD_ADSR: [4, 5, 6, 7, 9], S.MOD_GRID_SEQUENCER: [3, 4, 5, 6],
         S.MOD_PATTERN_SEQUENCER: [1, 2, 3], S.MOD_SAMPLE: [3, 4, 5]}
lo, hi = int(sys.argv[1]), int(sys.argv[2])
noise = len(sys.argv) > 3
bad, n, skipped, t0 = [], 0, 0, time.time()
for seed in range(lo, hi):
    B, build, overrides = random_patch(seed, noise)
    V = 9
    T = B * max(1, (700 + seed % 900) // B)          # a whole number of blocks: the oracle ticks block by block
    ovs = [(m, f, fn(V)) for m, f, fn in overrides]
    want = {}
    for v in (0, 4, 8):
        o = O.OraclePatch(48000, B, 2)
        ids = build(o)
        o.set_noise_seed(*( (seed * 7919 + 1, 1000 * seed + v) if noise else (0, v)))
        for m, f, vals in ovs: o.set_field(ids[m], f, float(vals[v]))
        o.render(T)
        want[v] = o
    for flags in (1, 3, 5):
        p = S.Patch(48000, B, 2)
        ids = build(p)
        p.configure_voices(V)
        p.keep_state(flags != 5)     # with keep_state every planned module is evaluated, as the reference's execute() does
        for m, f, vals in ovs: p.set_voice_field(ids[m], f, vals)
        init = {(m, f): p.get_voice_field(m, f) for m in range(p.num_modules()) for f in STATE.get(p.module_type(m), [])}
        p.render_channels(T, flags)
        n += 1
        for m in range(p.num_modules()):
            fields = STATE.get(p.module_type(m), [])
            got = {f: p.get_voice_field(m, f) for f in fields}
            # a module that cannot influence any output is not evaluated here (dead-code elimination): its state stays as stored,
            # while the reference ticks everything its planner reaches
            if fields and all((got[f].view(np.uint64) == init[(m, f)].view(np.uint64)).all() for f in fields):
                if any(want[v].get_field(m, f) != init[(m, f)][v] for v in want for f in fields): skipped += 1
                continue
            for f in fields:
                for v, o in want.items():
                    w = o.get_field(m, f)
                    same = (np.float64(got[f][v]).view(np.uint64) == np.float64(w).view(np.uint64)) or (np.isnan(got[f][v]) and np.isnan(w))
                    if not same: bad.append((seed, flags, m, p.module_type(m), f, v, float(got[f][v]), float(w)))
print(f"state soak, seeds {lo}..{hi - 1} noise={noise}: {n} renders, {len(bad)} state values differ ({skipped} unevaluated modules left out), {time.time() - t0:.0f} s")
for b in bad[:30]: print("  seed %d flags %d module %d (type %d) field %d voice %d: gpu %r oracle %r" % b)
