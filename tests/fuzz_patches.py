"""Random patches for differential tests (tests/test_gpu_fuzz.py: GPU vs oracle; tests/test_oracle.py: C oracle vs NumPy twin)."""
import numpy as np

import srack_pkg

W = srack_pkg.load_workloads()

OSC, VCF, ADSR, VCA, MIX, MATH, GRID, PAT, NONLIN, SMP, NOISE, VERB = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
N_IN = {OSC: 2, VCF: 2, ADSR: 1, VCA: 2, MIX: 4, MATH: 2, GRID: 2, PAT: 2, NONLIN: 2, SMP: 2, NOISE: 0, VERB: 2}
import os
# (FUZZ_SINE=1: tools/fuzz_soak.py also draws the oscillators' sine port)
OUT_PORTS = {OSC: [0, 1, 2] if os.environ.get("FUZZ_SINE") else [1, 2], VCF: [0, 1, 2], ADSR: [0], VCA: [0], MIX: [0], MATH: [0], NONLIN: [0], GRID: [0, 1, 2], PAT: [0, 3, 7, 8], SMP: [0], NOISE: [0], VERB: [0, 1]}


def random_patch(seed, noise=False):
    """-> (B, build(g) -> None, overrides [(module, field, values per voice fn)]).  noise=True: a fifth of the modules are
    NoiseModules and a tenth FreeverbModules (a separate family of patches: the seeds of the first family keep their meaning)."""
    rng = np.random.default_rng(seed if not noise else (seed, 0x4e6f))
    B = int(rng.choice([1, 3, 16, 64, 1024]))
    n = int(rng.integers(4, 11))
    types = [OSC, OSC] + [int(rng.choice([OSC, VCF, ADSR, VCA, MIX, MATH, GRID, PAT, SMP], p=[.2, .15, .12, .13, .1, .15, .05, .05, .05])) for _ in range(n - 2)]
    rng.shuffle(types)
    if noise:
        types = [NOISE if u < 0.2 else VERB if u < 0.3 else t for t, u in zip(types, rng.random(n))]
        if NOISE not in types:
            types[int(rng.integers(0, n))] = NOISE
    fields, steps, conns, waves = [], [], [], []
    for m, t in enumerate(types):
        if t == OSC:
            fields.append((m, W.OSC_VAL, float(np.float32(rng.uniform(-6, 3)))))
            if rng.random() < 0.2:
                fields.append((m, W.OSC_ANTIALIASING, 0))
        elif t == VCF:
            fields += [(m, W.VCF_FREQ, float(np.float32(rng.uniform(0.02, 0.8)))), (m, W.VCF_RES, float(np.float32(rng.uniform(0, 1)))),
                       (m, W.VCF_EXP_AMT, float(np.float32(rng.uniform(0, 1))))]
        elif t == ADSR:
            fields += [(m, f, float(np.float32(v))) for f, v in zip((W.ADSR_A_SEC, W.ADSR_D_SEC, W.ADSR_S_VAL, W.ADSR_R_SEC),
                                                                    (rng.choice([0.0, 0.001, 0.004]), rng.uniform(0.001, 0.01), rng.uniform(0, 1), rng.uniform(0.001, 0.01)))]
        elif t == VCA:
            fields.append((m, W.VCA_NEGATIVE, int(rng.random() < 0.3)))
        elif t == MIX:
            fields += [(m, W.MIX_GAIN0 + k, float(np.float32(rng.uniform(0, 1.2)))) for k in range(4)]
        elif t == MATH:
            fields += [(m, W.MATH_CONSTANT, float(np.float32(rng.uniform(-1, 1)))), (m, W.MATH_OPERATION, int(rng.integers(0, 3)))]
        elif t == VERB:
            fields += [(m, f, float(rng.uniform(lo, hi))) for f, lo, hi in ((W.FREEVERB_DAMPENING, 0, 2), (W.FREEVERB_WET, 0, 1), (W.FREEVERB_WIDTH, 0, 1),
                                                                            (W.FREEVERB_ROOM_SIZE, 0, 1), (W.FREEVERB_DRY, 0, 1))]
            fields.append((m, W.FREEVERB_FREEZE, int(rng.random() < 0.2)))
        elif t == SMP:
            waves.append((m, rng.uniform(-1, 1, int(rng.integers(1, 400))).astype(np.float32), float(rng.choice([8000.0, 44100.0, 48000.0, 96000.0]))))
        elif t in (GRID, PAT):
            length = int(rng.integers(1, 9))
            fields.append((m, W.GRIDSEQ_LENGTH if t == GRID else W.PATSEQ_LENGTH, length))
            for i in range(length):
                for ch in ([0] if t == GRID else [0, 3, 7]):
                    steps.append((m, ch, i, int(rng.integers(0, 3)), int(rng.integers(0, 25))))
    for m, t in enumerate(types):  # wire most inputs to a random output of a random OTHER module (self-loops are rejected)
        for k in range(N_IN[t]):
            if rng.random() < 0.75:
                src = int(rng.integers(0, n - 1))
                src += src >= m
                conns.append((src, int(rng.choice(OUT_PORTS[types[src]])), m, k))
    out_src = [int(rng.integers(0, n)) for _ in range(2)]
    out_conns = [(s, int(rng.choice(OUT_PORTS[types[s]])), c) for c, s in enumerate(out_src) if rng.random() < 0.9]
    if not out_conns:
        out_conns = [(out_src[0], OUT_PORTS[types[out_src[0]]][0], 0)]
    out_pos = int(rng.integers(0, n + 1))  # where the OutputModule sits in all_modules: the planner cares
    if os.environ.get("FUZZ_NONLIN"):  # (tools/cpu_soak.py: half of the MathModules become NonLinearModules — same arity, so the rest of the patch keeps its draw)
        r3 = np.random.default_rng((seed, 0x9))
        for m, t in enumerate(types):
            if t == MATH and r3.random() < 0.5:
                types[m] = NONLIN
                fields = [x for x in fields if x[0] != m] + [(m, W.NONLIN_CONSTANT, float(np.float32(r3.choice([0.5, 2.0, 3.0, r3.uniform(0.3, 3.0)]))))]

    def build(g):
        ids = []
        for i, t in enumerate(types):
            if i == out_pos:
                build.out = g.add_module(0)
            ids.append(g.add_module(t))
        if out_pos == n:
            build.out = g.add_module(0)
        shift = lambda m: ids[m]
        for m, f, v in fields:
            g.set_field(shift(m), f, v)
        for m, ch, i, st, val in steps:
            g.set_step(shift(m), ch, i, st, val)
        for m, wave, rate in waves:
            g.set_wave(shift(m), wave, rate)
        for s, sp, k, kp in conns:
            g.connect(shift(s), sp, shift(k), kp)
        for s, sp, c in out_conns:
            g.connect(shift(s), sp, build.out, c)
        if noise:
            g.set_noise_seed(seed * 7919 + 1, 1000 * seed)
        return ids

    overrides = []
    for m, t in enumerate(types):
        if rng.random() < 0.45:
            if t == OSC:
                overrides.append((m, W.OSC_VAL, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(-5, 2, V).astype(np.float32)))
            elif t == VCF:
                overrides.append((m, W.VCF_FREQ, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(0.03, 0.7, V).astype(np.float32)))
            elif t == MATH:
                overrides.append((m, W.MATH_CONSTANT, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(-1, 1, V).astype(np.float32)))
            elif t == NONLIN:
                overrides.append((m, W.NONLIN_CONSTANT, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(0.3, 3.0, V).astype(np.float32)))
            elif t == MIX:
                overrides.append((m, W.MIX_GAIN0 + 1, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(0, 1, V).astype(np.float32)))
            elif t == ADSR:
                overrides.append((m, W.ADSR_S_VAL, lambda V, r=np.random.default_rng(seed * 131 + m): r.uniform(0, 1, V).astype(np.float32)))
    if os.environ.get("FUZZ_MORE_OV"):  # (tools/fuzz_soak.py: per-voice overrides of further parameters and of initial STATE)
        r2 = np.random.default_rng((seed, 0x0F))
        more = {OSC: [(W.OSC_POS, 0.0, 1.0)], VCF: [(W.VCF_RES, 0.0, 1.0), (W.VCF_EXP_AMT, 0.0, 1.0), (W.VCF_ST_B2, -1.0, 1.0)],
                ADSR: [(W.ADSR_A_SEC, 0.0, 0.004), (W.ADSR_D_SEC, 0.0005, 0.01), (W.ADSR_R_SEC, 0.0005, 0.01), (W.ADSR_PHASE, 0.0, 1.0)],
                MIX: [(W.MIX_GAIN0, 0.0, 1.0), (W.MIX_GAIN3, -1.0, 1.0)]}
        for m, t in enumerate(types):
            for f, lo_, hi_ in more.get(t, []):
                if r2.random() < 0.35:
                    gen = np.random.default_rng((seed, m, f))
                    if f == W.OSC_POS:
                        overrides.append((m, f, lambda V, g=gen: g.uniform(0.0, 1.0, V)))  # f64 state
                    else:
                        overrides.append((m, f, lambda V, g=gen, a=lo_, b=hi_: g.uniform(a, b, V).astype(np.float32)))
    return B, build, overrides
