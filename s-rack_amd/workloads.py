"""The BASELINE.json patches and per-voice parameter sets, written against the graph API only.

`build_p1` / `build_p2` take any object with add_module / connect / set_field (the product's
`Patch`, or the test oracle's `OraclePatch`) and wire the patch of SURVEY §8(d) in the same
`all_modules` order, so the planner sees identical lists on both sides.
"""
import numpy as np

# numeric vocabulary of include/srack_hip.h
(MOD_OUTPUT, MOD_OSCILLATOR, MOD_MOOG_FILTER, MOD_ADSR, MOD_VCA, MOD_MONO_MIXER, MOD_MATH, MOD_GRID_SEQUENCER,
 MOD_PATTERN_SEQUENCER, MOD_NONLINEAR, MOD_SAMPLE, MOD_NOISE, MOD_FREEVERB) = range(13)
OSC_VAL, OSC_ANTIALIASING, OSC_POS, OSC_SYNC_LAST = range(4)
(VCF_FREQ, VCF_RES, VCF_EXP_AMT, VCF_ST_F, VCF_ST_P, VCF_ST_Q, VCF_ST_B0, VCF_ST_B1, VCF_ST_B2, VCF_ST_B3,
 VCF_ST_B4, VCF_ST_FREQ, VCF_ST_RES) = range(13)
(ADSR_A_SEC, ADSR_D_SEC, ADSR_S_VAL, ADSR_R_SEC, ADSR_PHASE, ADSR_MODE, ADSR_R_VAL, ADSR_FROM_A_VAL,
 ADSR_SAMPLE_RATE, ADSR_GATE_LAST) = range(10)
VCA_NEGATIVE = 0
MIX_GAIN0, MIX_GAIN1, MIX_GAIN2, MIX_GAIN3 = range(4)
MATH_CONSTANT, MATH_OPERATION = range(2)
MATH_ADD, MATH_SUBTRACT, MATH_MULTIPLY = range(3)
(GRIDSEQ_STEPS_PER_OCTAVE, GRIDSEQ_OCTAVES, GRIDSEQ_LENGTH, GRIDSEQ_CURRENT_STEP, GRIDSEQ_STEP_LAST, GRIDSEQ_SYNC_LAST,
 GRIDSEQ_LAST) = range(7)
PATSEQ_LENGTH, PATSEQ_CURRENT_STEP, PATSEQ_STEP_LAST, PATSEQ_SYNC_LAST = range(4)
NONLIN_CONSTANT = 0
FREEVERB_DAMPENING, FREEVERB_FREEZE, FREEVERB_WET, FREEVERB_WIDTH, FREEVERB_ROOM_SIZE, FREEVERB_DRY = range(6)
SAMPLE_SAMPLE_RATE, SAMPLE_WAVE_SAMPLE_RATE, SAMPLE_WAVE_NEW, SAMPLE_POS, SAMPLE_PLAYING, SAMPLE_GATE_LAST = range(6)
STEP_NONE, STEP_ON, STEP_HOLD = range(3)
GRIDSEQ_OUT_CV, GRIDSEQ_OUT_GATE, GRIDSEQ_OUT_SYNC = range(3)
PATSEQ_OUT_SYNC = 8
OSC_OUT_SINE, OSC_OUT_SQUARE, OSC_OUT_SAW = range(3)
VCF_OUT_LOWPASS, VCF_OUT_BANDPASS, VCF_OUT_HIGHPASS = range(3)

SEED = 0x5EED5EED


def build_p1(g, adsr="default", lfo_val=-8.0):
    """Patch P1 (configs 1/2/3/5): saw VCO -> ladder VCF -> VCA, ADSR gated by an LFO square.

    List order: [0] OSC_A, [1] OSC_LFO, [2] VCF, [3] ADSR, [4] VCA, [5] OUTPUT.
    adsr="default": a 0.0 / d 0.5 / s 0.25 / r 0.5 (adsr.rs:39-42; a_sec = 0 => inf increment);
    adsr="finite":  a 0.01 / d 0.1 / s 0.5 / r 0.2.
    """
    osc_a = g.add_module(MOD_OSCILLATOR)
    osc_lfo = g.add_module(MOD_OSCILLATOR)
    vcf = g.add_module(MOD_MOOG_FILTER)
    adsr_m = g.add_module(MOD_ADSR)
    vca = g.add_module(MOD_VCA)
    out = g.add_module(MOD_OUTPUT)
    g.set_field(osc_a, OSC_VAL, 0.0)
    g.set_field(osc_lfo, OSC_VAL, lfo_val)
    g.set_field(vcf, VCF_FREQ, 0.2)
    g.set_field(vcf, VCF_RES, 0.5)
    g.set_field(vcf, VCF_EXP_AMT, 0.5)
    if adsr == "finite":
        g.set_field(adsr_m, ADSR_A_SEC, 0.01)
        g.set_field(adsr_m, ADSR_D_SEC, 0.1)
        g.set_field(adsr_m, ADSR_S_VAL, 0.5)
        g.set_field(adsr_m, ADSR_R_SEC, 0.2)
    g.connect(osc_a, OSC_OUT_SAW, vcf, 0)
    g.connect(osc_lfo, OSC_OUT_SQUARE, adsr_m, 0)
    g.connect(vcf, VCF_OUT_LOWPASS, vca, 0)
    g.connect(adsr_m, 0, vca, 1)
    g.connect(vca, 0, out, 0)
    g.connect(vca, 0, out, 1)
    return dict(osc_a=osc_a, osc_lfo=osc_lfo, vcf=vcf, adsr=adsr_m, vca=vca, out=out)


def build_p2(g, beta=0.3, index=1.0):
    """Patch P2 (config 4): 2-op FM with a feedback edge.

    List order: [0] OSC_M, [1] MUL_FB (x beta), [2] MUL_IDX (x index), [3] OSC_C, [4] OUTPUT.
    OSC_M.sine -> MUL_FB -> OSC_M.cv is a cycle; the planner delays OSC_M.sine -> MUL_FB by
    buffer_size samples (MUL_FB runs first on the previous block's sine).
    """
    osc_m = g.add_module(MOD_OSCILLATOR)
    mul_fb = g.add_module(MOD_MATH)
    mul_idx = g.add_module(MOD_MATH)
    osc_c = g.add_module(MOD_OSCILLATOR)
    out = g.add_module(MOD_OUTPUT)
    for m, c in ((mul_fb, beta), (mul_idx, index)):
        g.set_field(m, MATH_OPERATION, MATH_MULTIPLY)
        g.set_field(m, MATH_CONSTANT, c)
    g.connect(osc_m, OSC_OUT_SINE, mul_fb, 0)
    g.connect(mul_fb, 0, osc_m, 0)
    g.connect(osc_m, OSC_OUT_SINE, mul_idx, 0)
    g.connect(mul_idx, 0, osc_c, 0)
    g.connect(osc_c, OSC_OUT_SINE, out, 0)
    g.connect(osc_c, OSC_OUT_SINE, out, 1)
    return dict(osc_m=osc_m, mul_fb=mul_fb, mul_idx=mul_idx, osc_c=osc_c, out=out)


def build_p3(g, clock_val=-4.0, length=8):
    """Patch P3 (scope table (f) rank 1): the screenshot's shape — an LFO clocks a grid sequencer whose CV plays the
    VCO and whose gate fires the amplitude ADSR; a pattern sequencer on the same clock gates a second ADSR that
    sweeps the filter.  A per-voice transpose is a Math(Add) after the sequencer CV.

    List order: [0] CLOCK (LFO), [1] GRID, [2] PATTERN, [3] TRANSPOSE (Add), [4] OSC, [5] ADSR_AMP, [6] ADSR_FLT,
    [7] VCF, [8] VCA, [9] OUTPUT.
    """
    clock = g.add_module(MOD_OSCILLATOR)
    grid = g.add_module(MOD_GRID_SEQUENCER)
    pat = g.add_module(MOD_PATTERN_SEQUENCER)
    transpose = g.add_module(MOD_MATH)
    osc = g.add_module(MOD_OSCILLATOR)
    adsr_amp = g.add_module(MOD_ADSR)
    adsr_flt = g.add_module(MOD_ADSR)
    vcf = g.add_module(MOD_MOOG_FILTER)
    vca = g.add_module(MOD_VCA)
    out = g.add_module(MOD_OUTPUT)
    g.set_field(clock, OSC_VAL, clock_val)
    g.set_field(grid, GRIDSEQ_LENGTH, length)
    g.set_field(pat, PATSEQ_LENGTH, length)
    notes = [0, 3, 7, 12, 10, 7, 3, 5]
    for i in range(length):
        if i % 4 == 3:
            continue  # a rest: the CV holds the previous note, the gate stays low
        g.set_step(grid, 0, i, STEP_HOLD if i % 4 == 1 else STEP_ON, notes[i % len(notes)])
        g.set_step(pat, 1, i, STEP_ON if i % 2 == 0 else STEP_NONE)
        g.set_step(pat, 5, i, STEP_HOLD if i == 2 else STEP_NONE)
    for m, vals in ((adsr_amp, (0.002, 0.02, 0.6, 0.01)), (adsr_flt, (0.001, 0.03, 0.3, 0.02))):
        for f, v in zip((ADSR_A_SEC, ADSR_D_SEC, ADSR_S_VAL, ADSR_R_SEC), vals):
            g.set_field(m, f, v)
    g.set_field(transpose, MATH_OPERATION, MATH_ADD)
    g.set_field(transpose, MATH_CONSTANT, -1.0)
    g.set_field(vcf, VCF_FREQ, 0.15)
    g.set_field(vcf, VCF_RES, 0.6)
    g.set_field(vcf, VCF_EXP_AMT, 0.4)
    g.connect(clock, OSC_OUT_SQUARE, grid, 0)
    g.connect(clock, OSC_OUT_SQUARE, pat, 0)
    g.connect(grid, GRIDSEQ_OUT_SYNC, pat, 1)         # the grid's sync output re-syncs the pattern
    g.connect(grid, GRIDSEQ_OUT_CV, transpose, 0)
    g.connect(transpose, 0, osc, 0)
    g.connect(grid, GRIDSEQ_OUT_GATE, adsr_amp, 0)
    g.connect(pat, 1, adsr_flt, 0)
    g.connect(osc, OSC_OUT_SAW, vcf, 0)
    g.connect(adsr_flt, 0, vcf, 1)
    g.connect(vcf, VCF_OUT_LOWPASS, vca, 0)
    g.connect(adsr_amp, 0, vca, 1)
    g.connect(vca, 0, out, 0)
    g.connect(pat, 5, out, 1)                         # a raw gate on the second channel
    return dict(clock=clock, grid=grid, pat=pat, transpose=transpose, osc=osc, adsr_amp=adsr_amp, adsr_flt=adsr_flt, vcf=vcf,
                vca=vca, out=out)


def p4_wave(n=1500):
    """A deterministic stand-in for a loaded .wav (first channel as f32, WaveBox::load sample.rs:31-69): a decaying
    two-partial tone with a little hash noise, so neighbouring samples differ enough for an index slip to show."""
    i = np.arange(n, dtype=np.float64)
    with np.errstate(over="ignore"):
        noise = (splitmix64(np.arange(n, dtype=np.uint64) ^ np.uint64(0xABCDEF)) >> np.uint64(40)).astype(np.float64) / 16777216.0 - 0.5
    w = np.exp(-i / (0.35 * n)) * (0.7 * np.sin(2 * np.pi * i * 0.031) + 0.25 * np.sin(2 * np.pi * i * 0.173)) + 0.05 * noise
    return w.astype(np.float32)


def build_p4(g, wave=None, wave_rate=44100.0, clock_val=-3.0):
    """Patch P4 (scope table (f) rank 4): a clocked sample player with vibrato through the sign-preserving waveshaper.

    List order: [0] CLOCK (square -> Sample.gate), [1] LFO (sine), [2] DEPTH (Multiply: vibrato depth), [3] SAMPLE,
    [4] SHAPER (NonLinear, exponent = its constant), [5] OUTPUT (ch0 <- shaper, ch1 <- raw sample).
    """
    clock = g.add_module(MOD_OSCILLATOR)
    lfo = g.add_module(MOD_OSCILLATOR)
    depth = g.add_module(MOD_MATH)
    smp = g.add_module(MOD_SAMPLE)
    shaper = g.add_module(MOD_NONLINEAR)
    out = g.add_module(MOD_OUTPUT)
    g.set_field(clock, OSC_VAL, clock_val)
    g.set_field(lfo, OSC_VAL, -5.0)
    g.set_field(depth, MATH_OPERATION, MATH_MULTIPLY)
    g.set_field(depth, MATH_CONSTANT, 0.5)
    g.set_field(shaper, NONLIN_CONSTANT, 0.75)
    g.set_wave(smp, p4_wave() if wave is None else wave, wave_rate)
    g.connect(clock, OSC_OUT_SQUARE, smp, 0)
    g.connect(lfo, OSC_OUT_SINE, depth, 0)
    g.connect(depth, 0, smp, 1)
    g.connect(smp, 0, shaper, 0)
    g.connect(shaper, 0, out, 0)
    g.connect(smp, 0, out, 1)
    return dict(clock=clock, lfo=lfo, depth=depth, smp=smp, shaper=shaper, out=out)


def p4_voice_params(n_voices, seed=SEED, first_voice=0):
    """per-voice vibrato depth = U(0, 1) octaves and shaper exponent = U(0.5, 2.0) (the UI slider's range, math.rs:317)."""
    u0 = voice_uniform(n_voices, 0, seed, first_voice)
    u1 = voice_uniform(n_voices, 1, seed, first_voice)
    return u0.astype(np.float32), (np.float32(0.5) + u1 * np.float32(1.5)).astype(np.float32)


def splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def voice_uniform(n_voices, k, seed=SEED, first_voice=0):
    """Counter-based U[0,1) f32 per voice: top 24 bits of splitmix64(seed ^ (voice*2+k)).

    Depends only on the GLOBAL voice index, so a rank that owns voices [v0, v1) draws the same
    numbers the single-GPU run draws for them.
    """
    with np.errstate(over="ignore"):
        v = np.arange(first_voice, first_voice + n_voices, dtype=np.uint64)
        bits = splitmix64(np.uint64(seed) ^ (v * np.uint64(2) + np.uint64(k)))
    return ((bits >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)


def p1_voice_params(n_voices, seed=SEED, first_voice=0):
    """cfg3/cfg5 per-voice randomisation: OSC_A.val = U(-1/24, +1/24) octaves, VCF.freq = U(0.05, 0.6).

    Returned as f32, computed in f32 so host, oracle and device hold identical bits.
    """
    u0 = voice_uniform(n_voices, 0, seed, first_voice)
    u1 = voice_uniform(n_voices, 1, seed, first_voice)
    detune = (u0 * np.float32(2.0) - np.float32(1.0)) * np.float32(1.0 / 24.0)
    cutoff = np.float32(0.05) + u1 * np.float32(0.55)
    return detune.astype(np.float32), cutoff.astype(np.float32)


def p1_poly_voice_params(n_voices, seed=SEED, first_voice=0):
    """cfg3_poly: config 3's draw (detune, cutoff) plus a per-voice gate LFO rate and per-voice envelope times, so that NOTHING of P1 is
    voice-invariant (every voice has its own gate pattern and envelope: real polyphony).  OSC_LFO.val = -8 + U(0, 3) octaves
    (1.72 ... 13.75 Hz: one to thirteen notes in the second), ADSR a_sec = U(0.002, 0.05), d_sec = U(0.05, 0.3), s_val = U(0.2, 0.8),
    r_sec = U(0.05, 0.4).  Draw k >= 2 comes from the same counter-based generator under seed + 0x1000 * (k // 2), so that no two draws
    of neighbouring voices coincide.  -> dict name -> f32[n_voices]."""
    u = [voice_uniform(n_voices, k % 2, seed + 0x1000 * (k // 2), first_voice) for k in range(7)]
    f = np.float32
    detune, cutoff = p1_voice_params(n_voices, seed, first_voice)
    return dict(detune=detune, cutoff=cutoff,
                lfo_val=(f(-8.0) + u[2] * f(3.0)).astype(f),
                a_sec=(f(0.002) + u[3] * f(0.048)).astype(f), d_sec=(f(0.05) + u[4] * f(0.25)).astype(f),
                s_val=(f(0.2) + u[5] * f(0.6)).astype(f), r_sec=(f(0.05) + u[6] * f(0.35)).astype(f))


def p1_poly_overrides(ids, pv):
    """the (module, field, values) list of p1_poly_voice_params' draw, for Patch.set_voice_field / OraclePatch.render_batch"""
    return [(ids["osc_a"], OSC_VAL, pv["detune"]), (ids["vcf"], VCF_FREQ, pv["cutoff"]), (ids["osc_lfo"], OSC_VAL, pv["lfo_val"]),
            (ids["adsr"], ADSR_A_SEC, pv["a_sec"]), (ids["adsr"], ADSR_D_SEC, pv["d_sec"]), (ids["adsr"], ADSR_S_VAL, pv["s_val"]),
            (ids["adsr"], ADSR_R_SEC, pv["r_sec"])]


def p2_voice_params(n_voices, seed=SEED, first_voice=0):
    """cfg4 per-voice randomisation: feedback beta = U(0.1, 0.4), index = U(0.5, 1.5)."""
    u0 = voice_uniform(n_voices, 0, seed, first_voice)
    u1 = voice_uniform(n_voices, 1, seed, first_voice)
    return (np.float32(0.1) + u0 * np.float32(0.3)).astype(np.float32), (np.float32(0.5) + u1).astype(np.float32)


BENCH_WORKLOADS = ("cfg3", "cfg3_poly", "cfg2", "cfg4", "cfg4_b1024", "p3", "p4")


def bench_workload(name, n_voices, first_voice=0, seed=SEED):
    """The workloads `bench.py` times, by name, as data: -> (buffer_size, build(g) -> ids, overrides(ids) -> [(module, field, f32[n_voices])]).
    `build` takes any object with the graph API (the product's Patch, the test oracle's OraclePatch); the per-voice draws are keyed by the
    GLOBAL voice index first_voice + v, so a shard is the same voices whatever the rank count."""
    f = np.float32
    if name in ("cfg3", "cfg2", "cfg3_poly"):
        def overrides(ids):
            if name == "cfg2":
                return []
            if name == "cfg3_poly":
                return p1_poly_overrides(ids, p1_poly_voice_params(n_voices, seed, first_voice))
            det, cut = p1_voice_params(n_voices, seed, first_voice)
            return [(ids["osc_a"], OSC_VAL, det), (ids["vcf"], VCF_FREQ, cut)]
        return 1024, build_p1, overrides
    if name in ("cfg4", "cfg4_b1024"):
        def overrides(ids):
            beta, index = p2_voice_params(n_voices, seed, first_voice)
            return [(ids["mul_fb"], MATH_CONSTANT, beta), (ids["mul_idx"], MATH_CONSTANT, index)]
        return (1 if name == "cfg4" else 1024), build_p2, overrides
    if name == "p3":
        def overrides(ids):
            u0, u1 = voice_uniform(n_voices, 0, seed, first_voice), voice_uniform(n_voices, 1, seed, first_voice)
            return [(ids["transpose"], MATH_CONSTANT, (u0 * f(2.5) - f(2.0)).astype(f)), (ids["vcf"], VCF_FREQ, (f(0.05) + u1 * f(0.35)).astype(f))]
        return 1024, build_p3, overrides
    if name == "p4":
        def overrides(ids):
            depth, expo = p4_voice_params(n_voices, seed, first_voice)
            return [(ids["depth"], MATH_CONSTANT, depth), (ids["shaper"], NONLIN_CONSTANT, expo)]
        return 1024, build_p4, overrides
    raise ValueError(f"unknown workload {name!r}")
