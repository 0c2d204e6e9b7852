"""csrc/approx.cpp on its own (CPU only): the default mode's approximations as an error budget — one case per structure the bound has to
get right (what were soak-derived rules in flatten.cpp 2b until round 5, each named after the fuzz seed that had found it), the numbers it
rests on against tools/ladder_calib.c's emulation of the two ladders, and the benchmarked workloads (none may lose a fast form).
The analysis is compiled by g++ without HIP (tests/cpp/approx_probe.cpp + csrc/graph.cpp + csrc/approx.cpp) and fed patches on stdin."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

import srack_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = srack_pkg.load_workloads()
CSRC = os.path.join(ROOT, "s-rack_amd", "csrc")
BUDGET, HORIZON, EVENT = 5e-6, 2.88e7, 1e9   # approx.hpp / approx.cpp
EPS_BLEP = 3.2e-7                             # approx.cpp kEpsBlep (tools/blep_calib.py)


def _build(binary, sources, extra=()):
    if not os.path.exists(binary) or os.path.getmtime(binary) < max(os.path.getmtime(s) for s in sources):
        tmp = f"{binary}.{os.getpid()}.tmp"   # (pytest-xdist workers build side by side: each into its own file, then an atomic rename)
        subprocess.run(["g++" if sources[0].endswith("pp") else "gcc", "-O1", "-o", tmp] + list(extra) + [s for s in sources if not s.endswith(".hpp") and not s.endswith(".h")] + ["-lm"], check=True)
        os.replace(tmp, binary)
    return binary


@pytest.fixture(scope="module")
def probe():
    srcs = [os.path.join(ROOT, "tests", "cpp", "approx_probe.cpp"), os.path.join(CSRC, "graph.cpp"), os.path.join(CSRC, "approx.cpp"),
            os.path.join(CSRC, "approx.hpp"), os.path.join(CSRC, "graph.hpp"), os.path.join(CSRC, "flatten.hpp"), os.path.join(ROOT, "include", "srack_hip.h")]
    return _build(os.path.join(ROOT, "tests", "cpp", "approx_probe"), srcs, ["-std=c++17", "-Wall"])


class Rec:
    """The graph API of Patch / OraclePatch, recorded as the probe's input lines."""

    def __init__(self, sample_rate=48000, buffer_size=1024, channels=2):
        self.lines, self.n = [f"cfg {sample_rate} {buffer_size} {channels}"], 0

    def add_module(self, t):
        self.lines.append(f"mod {t}")
        self.n += 1
        return self.n - 1

    def set_field(self, m, f, v):
        self.lines.append(f"field {m} {f} {float(v)!r}")

    def set_step(self, m, ch, i, st, val=0):
        self.lines.append(f"step {m} {ch} {i} {st} {val}")

    def set_wave(self, m, wave, rate):
        self.lines.append(f"wave {m} {len(wave)} " + " ".join(repr(float(x)) for x in wave))

    def connect(self, a, ap, b, bp):
        self.lines.append(f"conn {a} {ap} {b} {bp}")

    def override(self, m, f, values):
        self.lines.append(f"ov {m} {f} {len(values)} " + " ".join(repr(float(x)) for x in values))

    def run(self, probe, exact=False):
        r = subprocess.run([probe], input="\n".join(self.lines + (["exact"] if exact else [])) + "\n", capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stderr
        return json.loads(r.stdout)


OSC, VCF, ADSR, VCA, MIX, MATH, GRID, PAT, NONLIN, SMP, NOISE = (W.MOD_OSCILLATOR, W.MOD_MOOG_FILTER, W.MOD_ADSR, W.MOD_VCA, W.MOD_MONO_MIXER, W.MOD_MATH,
                                                               W.MOD_GRID_SEQUENCER, W.MOD_PATTERN_SEQUENCER, W.MOD_NONLINEAR, W.MOD_SAMPLE, W.MOD_NOISE)
SINE, SQUARE, SAW = 0, 1, 2


def chain(*specs):
    """modules by type, then wires (a, port, b, port); the last module added is the OutputModule.  -> Rec, ids"""
    g = Rec()
    return g, [g.add_module(t) for t in specs] + [g.add_module(W.MOD_OUTPUT)]


# ---- the straight way to the output: every cheap form taken, the bound is the sum of their contributions -------------------------------
def test_straight_chain_takes_every_form_and_says_its_bound(probe):
    g, (osc, vcf, out) = chain(OSC, VCF)
    g.connect(osc, SAW, vcf, 0)
    g.connect(vcf, 0, out, 0)
    r = g.run(probe)
    assert r["exact_blep"][osc] == 0 and r["literal"][vcf] == 0 and r["saw_fixed"][osc] == 1 and not r["exact_patch"]
    l1 = r["gain"][osc][SAW]                       # the lowpass's L1 norm at the default cutoff 0.2 / resonance 0.5 (tools/ladder_calib.c l1: 1.54)
    assert 1.5 < l1 < 1.6
    dt = 440.0 / 48000.0                             # the fixed-point phase's window term: 2^-31 over the smallest increment (5.1e-8 at 440 Hz)
    assert r["bound"] == pytest.approx(1.5e-6 + EPS_BLEP * l1 + (3.2e-12 + 2.0 ** -31 / dt) * l1, rel=1e-6) and r["bound"] < BUDGET
    assert g.run(probe, exact=True)["saw_fixed"][osc] == 0   # exact mode: no form on offer at all


def test_ladder_l1_norms_against_known_values(probe):
    """resonance 0: the lowpass's impulse response is positive and sums to its DC gain 1; the highpass = input - lowpass: 2 at low cutoffs."""
    g, (osc, vcf, out) = chain(OSC, VCF)
    g.set_field(vcf, W.VCF_RES, 0.0)
    g.set_field(vcf, W.VCF_FREQ, 0.05)
    g.connect(osc, SAW, vcf, 0)
    g.connect(vcf, 2, out, 0)
    g.connect(vcf, 0, out, 1)
    r = g.run(probe)
    assert r["gain"][vcf] == [1.0, 0.0, 1.0]
    assert r["gain"][osc][SAW] == pytest.approx(2.0, abs=0.01)   # max over the two channels: the highpass's


# ---- what integrates: a pitch input -------------------------------------------------------------------------------------------------------
def test_a_producer_into_a_pitch_is_exact_by_the_horizon(probe):
    g, (lfo, osc, out) = chain(OSC, OSC)
    g.set_field(lfo, W.OSC_VAL, -6.0)
    g.connect(lfo, SAW, osc, 0)
    g.connect(osc, SINE, out, 0)
    r = g.run(probe)
    # d sine / d cv = horizon x ln2 x delta x 2 pi with delta = 440 x 2^(0 + |saw| <= 1) / 48000
    delta = 440.0 * 2.0 / 48000.0
    assert r["gain"][lfo][SAW] == pytest.approx(HORIZON * math.log(2) * delta * 2 * math.pi, rel=1e-6)
    assert r["exact_blep"][lfo] == 1 and r["saw_fixed"][lfo] == 0 and not r["exact_patch"]
    assert r["sine_loose"][osc] == 1     # the carrier's own sine goes straight out


def test_a_sine_into_a_pitch_keeps_its_f64_form_and_the_patch_its_flavour(probe):
    g, (mod, car, out) = chain(OSC, OSC)
    g.connect(mod, SINE, car, 0)
    g.connect(car, SINE, out, 0)
    r = g.run(probe)
    assert r["sine_loose"] [mod] == 0 and r["sine_loose"][car] == 1 and not r["exact_patch"]


# ---- what thresholds: event inputs -----------------------------------------------------------------------------------------------------------
def test_event_inputs_and_the_square_that_arrives_unchanged(probe):
    # a square wired straight to a gate (P1's gate LFO): nothing to pay — square_sign_safe makes its edges the reference's
    g, (lfo, env, out) = chain(OSC, ADSR)
    g.connect(lfo, SQUARE, env, 0)
    g.connect(env, 0, out, 0)
    r = g.run(probe)
    assert r["exact_blep"][lfo] == 0 and r["gain"][lfo][SQUARE] == pytest.approx(EVENT)     # (the wire's gain is an event's all the same)
    # ... handed on by a sequencer's gate output: the same
    g, (lfo, seq, env, out) = chain(OSC, GRID, ADSR)
    g.set_step(seq, 0, 0, W.STEP_ON, 3)
    g.connect(lfo, SQUARE, seq, 0)
    g.connect(seq, W.GRIDSEQ_OUT_GATE, env, 0)
    g.connect(env, 0, out, 0)
    assert g.run(probe)["exact_blep"][lfo] == 0
    # ... through as many sequencers in a row as anyone wires, the last one's gate also HEARD (a VCA): past the fourth the hand-on is no longer
    # followed and that gate output counts with its ordinary gain — until round 5 it counted 0, readers and all
    g, ids = chain(OSC, *([GRID] * 6), ADSR, VCA)
    lfo, seqs, env, vca, out = ids[0], ids[1:7], ids[7], ids[8], ids[9]
    for q in seqs:
        g.set_step(q, 0, 0, W.STEP_ON, 3)
    g.connect(lfo, SQUARE, seqs[0], 0)
    for a_, b_ in zip(seqs, seqs[1:]):
        g.connect(a_, W.GRIDSEQ_OUT_GATE, b_, 0)
    g.connect(seqs[-1], W.GRIDSEQ_OUT_GATE, env, 0)
    g.connect(seqs[-1], W.GRIDSEQ_OUT_GATE, vca, 0)
    g.connect(env, 0, vca, 1)
    g.connect(vca, 0, out, 0)
    assert g.run(probe)["exact_blep"][lfo] == 1
    g, ids = chain(OSC, *([GRID] * 3), ADSR, VCA)   # three in a row: followed to the end — the gate costs nothing, the VCA hears 3.2e-7
    lfo, seqs, env, vca, out = ids[0], ids[1:4], ids[4], ids[5], ids[6]
    for q in seqs:
        g.set_step(q, 0, 0, W.STEP_ON, 3)
    g.connect(lfo, SQUARE, seqs[0], 0)
    for a_, b_ in zip(seqs, seqs[1:]):
        g.connect(a_, W.GRIDSEQ_OUT_GATE, b_, 0)
    g.connect(seqs[-1], W.GRIDSEQ_OUT_GATE, env, 0)
    g.connect(seqs[-1], W.GRIDSEQ_OUT_GATE, vca, 0)
    g.connect(env, 0, vca, 1)
    g.connect(vca, 0, out, 0)
    assert g.run(probe)["exact_blep"][lfo] == 0
    # ... after arithmetic it is a value like any other (round 4's seed 2691: a bandpass into a gate): exact PolyBLEP
    g, (lfo, gain, env, out) = chain(OSC, MATH, ADSR)
    g.set_field(gain, W.MATH_OPERATION, W.MATH_MULTIPLY)
    g.set_field(gain, W.MATH_CONSTANT, 0.5)
    g.connect(lfo, SQUARE, gain, 0)
    g.connect(gain, 0, env, 0)
    g.connect(env, 0, out, 0)
    assert g.run(probe)["exact_blep"][lfo] == 1
    # a saw into a gate; a filter into a sync: both pay
    g, (osc, vcf, slave, out) = chain(OSC, VCF, OSC)
    g.connect(osc, SAW, vcf, 0)
    g.connect(vcf, 1, slave, 1)
    g.connect(slave, SAW, out, 0)
    r = g.run(probe)
    assert r["exact_blep"][osc] == 1 and r["literal"][vcf] == 1 and r["exact_blep"][slave] == 0 and r["saw_fixed"][osc] == 0


def test_a_fixed_point_phase_is_for_audio_pitches(probe):
    """fosc_saw takes t = pos / dt inside the PolyBLEP windows from the phase's upper 32 bits: an error of up to 2^-31 / dt — 5e-8 at 440 Hz, 1.0e-6
    at 17 Hz (what the GPU rendered for the fuzzer's seed 900146 in round 6, against a bound that said 2.4e-7), 2.6e-5 for a 0.86 Hz LFO.  The
    form's epsilon is that over the smallest increment any voice has; an LFO keeps its f64 phase."""
    for val, per_voice, fixed in ((0.0, None, 1), (-9.0, None, 0), (0.0, [0.0, -2.0, -9.0], 0), (-3.0, None, 1)):
        g, (osc, out) = chain(OSC)
        g.set_field(osc, W.OSC_VAL, val)
        if per_voice is not None:
            g.override(osc, W.OSC_VAL, per_voice)
        g.connect(osc, SAW, out, 0)
        r = g.run(probe)
        lowest = min(per_voice) if per_voice is not None else val
        assert r["saw_fixed"][osc] == fixed and r["exact_blep"][osc] == 0, (val, per_voice, r["bound"])
        assert r["bound"] == pytest.approx(EPS_BLEP + (3.2e-12 + 2.0 ** -31 / (440.0 * 2.0 ** lowest / 48000.0) if fixed else 0.0), rel=1e-6)


def test_a_fixed_point_phase_is_denied_in_front_of_an_event_but_not_of_a_vca(probe):
    g, (osc, env, lfo, vca, out) = chain(OSC, ADSR, OSC, VCA)
    g.connect(lfo, SQUARE, env, 0)
    g.connect(env, 0, vca, 0)
    g.connect(osc, SAW, vca, 1)        # the saw opens the VCA: `cv > 0.0` there is continuous in the output (audio * cv), not an event
    g.connect(vca, 0, out, 0)
    assert g.run(probe)["saw_fixed"][osc] == 1
    g, (osc, env, out) = chain(OSC, ADSR)
    g.connect(osc, SAW, env, 0)
    g.connect(env, 0, out, 0)
    r = g.run(probe)
    assert r["saw_fixed"][osc] == 0 and r["exact_blep"][osc] == 1


# ---- a filter's cutoff ---------------------------------------------------------------------------------------------------------------------------
def test_producers_into_a_cutoff_pay_by_the_filters_sensitivity(probe):
    """Round 4's seed 10901: a saw through a highpass into a second filter's audio AND cutoff (amount 0.92, resonance 0.76), 2.6e-5 in the
    default forms.  The bound: epsilon x the highpass's L1 norm x 3 / cutoff x the second filter's norm x amount — far over; the same saw
    into a cutoff with a small amount at a high cutoff stays fast: a bound, not a reachability rule."""
    def patch(amount, freq, res):
        g, (osc, a, b, out) = chain(OSC, VCF, VCF)
        g.set_field(b, W.VCF_EXP_AMT, amount)
        g.set_field(b, W.VCF_FREQ, freq)
        g.set_field(b, W.VCF_RES, res)
        g.connect(osc, SAW, a, 0)
        g.connect(a, 2, b, 0)
        g.connect(a, 2, b, 1)
        g.connect(b, 1, out, 0)
        return g, osc, a, b
    g, osc, a, b = patch(0.92, 0.2, 0.76)
    r = g.run(probe)
    assert r["exact_blep"][osc] == 1 and r["literal"][a] == 1
    g, (lfo, osc, vcf, out) = chain(OSC, OSC, VCF)
    g.set_field(lfo, W.OSC_VAL, -7.0)      # 3.4 Hz: an LFO (a sine at audio rate on a cutoff is outside the contracted ladder's calibration: literal)
    g.set_field(vcf, W.VCF_EXP_AMT, 0.01)
    g.set_field(vcf, W.VCF_FREQ, 0.6)
    g.connect(osc, SAW, vcf, 0)
    g.connect(lfo, SINE, vcf, 1)
    g.connect(vcf, 0, out, 0)
    r = g.run(probe)
    assert r["sine_loose"][lfo] == 1 and r["literal"][vcf] == 0 and r["bound"] < BUDGET
    g.lines = [l for l in g.lines if not l.startswith(f"field {lfo} ")]   # the same at 440 Hz
    assert g.run(probe)["literal"][vcf] == 1


def test_a_cutoff_that_jumps(probe):
    """A cutoff that moves at audio rate — a square, a saw, a sine above LFO rate, noise, another filter's output — makes the ladder a
    time-varying system: tools/ladder_calib.c has the contracted form at 5e-5 under a square and the literal one's response to an input
    disturbance at 3.5 x its static norm (a unit saw on the input, resonance <= 0.8), unbounded above and under noise; the soak's seed
    105055 has it at 145 x the disturbance with a louder input.  No bound is claimed: the filter literal, its gains unbounded, everything in
    front exact (round 4's seeds 28336; 2127 / 2203 / 2360).  An envelope or an LFO on the cutoff (P3's sweep) is calibrated: contracted."""
    def patch(cv_type, res, lfo=False):
        g, (audio, cv, vcf, out) = chain(OSC, cv_type, VCF)
        g.set_field(vcf, W.VCF_RES, res)
        if lfo:
            g.set_field(cv, W.OSC_VAL, -7.0)
        g.connect(audio, SAW, vcf, 0)
        g.connect(cv, SQUARE if cv_type == OSC else 0, vcf, 1)
        g.connect(vcf, 0, out, 0)
        return g, audio, cv, vcf
    for cv_type in (OSC, NOISE):
        g, audio, cv, vcf = patch(cv_type, 0.5)
        r = g.run(probe)
        assert r["literal"][vcf] == 1 and r["exact_blep"][audio] == 1 and r["saw_fixed"][audio] == 0 and r["gain"][audio][SAW] == float("inf")
    g, audio, cv, vcf = patch(OSC, 0.5, lfo=True)            # a 3.4 Hz square: rare jumps — twice the still ladder's epsilon
    r = g.run(probe)
    assert r["literal"][vcf] == 0 and r["exact_blep"][audio] == 0 and 3.0e-6 < r["bound"] < BUDGET
    g, audio, cv, vcf = patch(ADSR, 0.5)                     # an envelope gated by a 3.4 Hz clock (P3's sweep): the same
    clock = g.add_module(OSC)
    g.set_field(clock, W.OSC_VAL, -7.0)
    g.connect(clock, SQUARE, cv, 0)
    r = g.run(probe)
    assert r["literal"][vcf] == 0 and r["exact_blep"][audio] == 0 and 3.0e-6 < r["bound"] < BUDGET
    # ... but an envelope moves as often as its gate opens: gated by the 440 Hz square (or by noise: tools/cpu_soak.py's seed 277445, the
    # contracted lowpass at 1.0e-5 where "rare jumps" had said 3e-6) it restarts at audio rate
    g, audio, cv, vcf = patch(ADSR, 0.5)
    g.connect(audio, SQUARE, cv, 0)
    r = g.run(probe)
    assert r["literal"][vcf] == 1 and r["exact_blep"][audio] == 1


def test_a_ladder_near_self_oscillation(probe):
    """The L1 norm is computed from the filter's own coefficients over the cutoffs it can reach: resonance 0.85 at cutoff 0.2 is 11 (fine), 0.95
    does not decay — an unbounded gain: the saw in front takes its exact form, and an oscillator whose pitch moves is evaluated exactly as a
    whole (round 2's finding: resonance >= 0.928 left the band in a few voices, up to 3e-3)."""
    def patch(res, vibrato=False, per_voice=None):
        g, (osc, vcf, lfo, out) = chain(OSC, VCF, OSC)
        g.set_field(vcf, W.VCF_RES, res)
        if per_voice is not None:
            g.override(vcf, W.VCF_RES, per_voice)
        g.connect(osc, SAW, vcf, 0)
        g.connect(vcf, 0, out, 0)
        if vibrato:
            g.connect(lfo, SINE, osc, 0)
        return g, osc, vcf
    g, osc, vcf = patch(0.85)
    r = g.run(probe)
    assert r["literal"][vcf] == 0 and 10.0 < r["gain"][osc][SAW] < 12.0
    assert r["exact_blep"][osc] == 1     # (11 x the f32 PolyBLEP's 3.2e-7 + the contracted ladder's 1.5e-6 is just over the budget; at 0.8 — L1 = 7 — both fit)
    assert patch(0.8)[0].run(probe)["exact_blep"][osc] == 0
    g, osc, vcf = patch(0.95)
    r = g.run(probe)
    assert r["literal"][vcf] == 1 and r["exact_blep"][osc] == 1 and r["gain"][osc][SAW] == float("inf") and not r["exact_patch"]
    g, osc, vcf = patch(0.95, vibrato=True)
    r = g.run(probe)
    assert not r["exact_patch"] and r["osc_exact"][osc] == 1 and r["osc_exact"][osc + 2] == 1   # the oscillator AND the LFO whose sine moves its pitch: exact as a whole
    g, osc, vcf = patch(0.5, per_voice=[0.1, 0.96, 0.3])     # one voice is enough
    assert g.run(probe)["literal"][vcf] == 1


# ---- cycles -------------------------------------------------------------------------------------------------------------------------------------
def test_cycles_converge_to_their_geometric_series_or_diverge(probe):
    def patch(feedback):
        g, (osc, mix, out) = chain(OSC, MIX)
        g.set_field(mix, W.MIX_GAIN0, 1.0)
        g.set_field(mix, W.MIX_GAIN1, feedback)
        g.connect(osc, SAW, mix, 0)
        g.connect(mix, 0, mix + 0, 1) if False else None
        return g, osc, mix
    # (a module may not be wired to itself: the loop goes through a second mixer with unit gain)
    def loop(feedback):
        g, (osc, a, b, out) = chain(OSC, MIX, MIX)
        g.set_field(a, W.MIX_GAIN1, feedback)
        g.connect(osc, SAW, a, 0)
        g.connect(a, 0, b, 0)
        g.connect(b, 0, a, 1)
        g.connect(a, 0, out, 0)
        return g, osc, a
    g, osc, a = loop(0.5)
    r = g.run(probe)
    assert r["gain"][osc][SAW] == pytest.approx(2.0, rel=1e-4) and r["exact_blep"][osc] == 0 and r["bound"] == pytest.approx(2 * (EPS_BLEP + 3.2e-12 + 2.0 ** -31 / (440.0 / 48000.0)), rel=1e-3)
    g, osc, a = loop(0.96)      # 25: the f32 PolyBLEP's 3.2e-7 becomes 8e-6
    r = g.run(probe)
    assert r["gain"][osc][SAW] == pytest.approx(25.0, rel=2e-2) and r["exact_blep"][osc] == 1
    g, osc, a = loop(0.99)      # 100: the fixpoint stops sweeping at ~400 with 1.6 % to go — the geometric tail is added, not dropped
    r = g.run(probe)
    assert r["gain"][osc][SAW] == pytest.approx(100.0, rel=2e-3)
    g, osc, a = loop(0.995)     # 200: 13 % to go after 400 sweeps (from 0.997 on a sweep still adds more than 1e-3: declared unbounded)
    r = g.run(probe)
    assert r["gain"][osc][SAW] == pytest.approx(200.0, rel=1e-2)
    assert loop(0.997)[0].run(probe)["gain"][osc][SAW] == float("inf")
    g, osc, a = loop(1.0)       # round 4's seed 40913: gain exactly one — an integrator: the gain AND the values have no bound
    r = g.run(probe)
    assert r["gain"][osc][SAW] == float("inf") and r["exact_blep"][osc] == 1 and r["exact_patch"] and "unbounded values" in r["why"]
    g, osc, a = loop(1.2)       # seed 4386: two mixers feeding each other with gains above one (its infinities and NaNs are the reference's only in the exact flavour)
    r = g.run(probe)
    assert r["gain"][osc][SAW] == float("inf") and r["exact_patch"]
    # a loop of gain one whose values ARE bounded — a ladder's lowpass in it, clamped to [-1, 1]: unbounded gain, and the saw in front takes its
    # exact form; nothing in the patch lacks one, so the flavour stays
    g, (osc, add, sub, vcf, out) = chain(OSC, MATH, MATH, VCF)
    g.set_field(sub, W.MATH_OPERATION, W.MATH_SUBTRACT)
    g.connect(osc, SAW, add, 1)
    g.connect(add, 0, sub, 0)
    g.connect(sub, 0, vcf, 0)
    g.connect(vcf, 0, add, 0)
    g.connect(add, 0, out, 0)
    r = g.run(probe)
    assert r["gain"][osc][SAW] == float("inf") and r["exact_blep"][osc] == 1 and r["literal"][vcf] == 1 and not r["exact_patch"]


def test_a_cycle_through_an_event_input_is_unbounded(probe):
    """Round 4's seed 725: the moved event comes back to what produced it."""
    g, (osc, env, vca, out) = chain(OSC, ADSR, VCA)
    g.connect(osc, SINE, vca, 0)
    g.connect(env, 0, vca, 1)
    g.connect(vca, 0, env, 0)          # the envelope's own output (times a sine) is its gate
    g.connect(vca, 0, out, 0)
    r = g.run(probe)
    assert r["gain"][osc][SINE] == float("inf") and r["osc_exact"][osc] == 1 and not r["exact_patch"]


def test_a_loop_through_a_pitch_is_unbounded(probe):
    """Config 4's loop: OSC_M.sine -> x beta -> OSC_M.cv.  A pitch input integrates, so first order every loop through one diverges.  In real
    arithmetic this one is neutral (the loop's multiplier per sample, 1 + a cos(...) with a = 0.02, has a logarithm that averages to -a^2 / 4) —
    but the sine on its way round is rounded to f32, a phase difference of 1e-12 flips one of those roundings now and then, and each flip
    kicks the pitch by 6e-8: tools/fm_sensitivity.c (the reference's arithmetic twice, one phase off by 1e-12: 1e-7 cycles apart after 35 s)
    and profiles/r05_horizon.json (round 4's default kernels: 4.6e-7 after a second, 1.5e-5 after a minute).  First order is right: the
    oscillator in the loop is evaluated exactly, and the render follows the reference bit for bit (same file: 0 for the whole minute);
    SRACK_RENDER_KEEP_DEFAULT keeps the fast kernels for a host that renders seconds, not minutes."""
    def p2(beta, port=SINE):
        g = Rec(48000, 1, 2)
        ids = W.build_p2(g, beta=beta)
        if port != SINE:
            g.lines = [l.replace(f"conn {ids['osc_m']} 0 {ids['mul_fb']} 0", f"conn {ids['osc_m']} {port} {ids['mul_fb']} 0") for l in g.lines]
        return g, ids
    for beta, port in ((0.3, SINE), (1.8, SINE), (0.3, SAW)):
        g, ids = p2(beta, port)
        r = g.run(probe)
        # the oscillator inside the loop is exact as a whole (the libm's pow, the reference's sine) — that one; the carrier behind it, which
        # nothing feeds back, keeps the default forms, and the patch its flavour
        assert r["gain"][ids["osc_m"]][port] == float("inf") and r["osc_exact"][ids["osc_m"]] == 1 and r["osc_exact"][ids["osc_c"]] == 0
        assert not r["exact_patch"] and r["sine_loose"][ids["osc_c"]] == 1 and r["sine_loose"][ids["osc_m"]] == 0
    # feed-forward FM — a sine into another oscillator's pitch, no way back — is bounded: the default forms stay
    g, (mod, gain, car, out) = chain(OSC, MATH, OSC)
    g.set_field(gain, W.MATH_OPERATION, W.MATH_MULTIPLY)
    g.set_field(gain, W.MATH_CONSTANT, 1.5)
    g.connect(mod, SINE, gain, 0)
    g.connect(gain, 0, car, 0)
    g.connect(car, SINE, out, 0)
    r = g.run(probe)
    assert not r["exact_patch"] and r["gain"][mod][SINE] < 1e8 and sum(r["osc_exact"]) == 0


# ---- a waveshaper's slope ------------------------------------------------------------------------------------------------------------------------
def test_a_shaper_with_an_exponent_below_one_is_steep_at_zero(probe):
    """sign(a) |a|^b (math.rs:203-205) has slope b |a|^(b-1): unbounded at 0 for b < 1 — (2.4e-7)^0.5 = 5e-4.  Not among round 4's rules (the
    fuzzer's patches have no NonLinearModule): derived."""
    def patch(exponent):
        g, (osc, shaper, out) = chain(OSC, NONLIN)
        g.set_field(shaper, W.NONLIN_CONSTANT, exponent)
        g.connect(osc, SAW, shaper, 0)
        g.connect(shaper, 0, out, 0)
        return g, osc, shaper
    g, osc, shaper = patch(0.5)
    r = g.run(probe)
    assert r["exact_blep"][osc] == 1 and r["saw_fixed"][osc] == 0
    assert r["nonlin_loose"][shaper] == 1      # (its own f32 power: 4e-6 relative to an output of up to 1 — fits, like P4's |sample| <= 1)
    g, osc, shaper = patch(2.0)
    r = g.run(probe)
    assert r["gain"][osc][SAW] == pytest.approx(2.0) and r["exact_blep"][osc] == 0 and r["saw_fixed"][osc] == 1   # b |a|^(b-1) at |a| = 1


def test_a_shaper_behind_an_unbounded_gain_loses_its_f32_form_only(probe):
    """until round 6 NonLinear's power WITHOUT the f32 form was the libm's powf to within an f32 ulp, never bit for bit (a table-driven log2,
    a polynomial 2^y) — and counted as 0: inside a loop through a pitch, where only identical bits follow the reference, the GPU parted from it (the
    fuzzer's seed 405576: 1.97).  Since round 6 that power IS the host libm's powf, operation for operation (modules.hip.h, powf_libm_plain;
    tests/libm_powf.py): denying the f32 form is enough again, and the oscillator in the loop is exact as a whole."""
    g, (osc, shaper, gain, out) = chain(OSC, NONLIN, MATH)
    g.set_field(gain, W.MATH_OPERATION, W.MATH_MULTIPLY)
    g.set_field(gain, W.MATH_CONSTANT, 0.3)
    g.connect(osc, SINE, shaper, 0)
    g.connect(shaper, 0, gain, 0)
    g.connect(gain, 0, osc, 0)          # osc.sine -> |.|^b -> x 0.3 -> osc.cv: a loop through a pitch
    g.connect(osc, SINE, out, 0)
    r = g.run(probe)
    assert r["gain"][shaper][0] == float("inf") and not r["exact_patch"] and r["nonlin_loose"][shaper] == 0 and r["osc_exact"][osc] == 1 and r["bound"] == 0.0
    # a shaper whose f32 form is denied by a steep reader (another shaper with an exponent below one): nothing left of it in the bound
    g, (osc, first, second, out) = chain(OSC, NONLIN, NONLIN)
    g.set_field(first, W.NONLIN_CONSTANT, 2.0)
    g.set_field(second, W.NONLIN_CONSTANT, 0.5)
    g.connect(osc, SAW, first, 0)
    g.connect(first, 0, second, 0)
    g.connect(second, 0, out, 0)
    r = g.run(probe)
    assert not r["exact_patch"] and r["nonlin_loose"][first] == 0 and r["gain"][first][0] == pytest.approx(1e4)
    assert r["bound"] == pytest.approx(4e-6 * max(1.0, r["mag"][second][0]) if r["nonlin_loose"][second] else 0.0, rel=1e-3)


# ---- the budget is shared ----------------------------------------------------------------------------------------------------------------------
def test_forms_are_denied_largest_first_until_the_channel_fits(probe):
    g, (osc, vcf, out) = chain(OSC, VCF)
    g.set_field(vcf, W.VCF_RES, 0.85)
    g.connect(osc, SAW, vcf, 0)
    g.connect(vcf, 1, out, 0)          # the bandpass at resonance 0.85: L1 = 23: the PolyBLEP's 5.6e-6 + the ladder's own 4.2e-6
    r = g.run(probe)
    assert r["exact_blep"][osc] == 1 and r["literal"][vcf] == 0 and r["bound"] == pytest.approx(4.2e-6, rel=1e-3)


# ---- hold / sweep ----------------------------------------------------------------------------------------------------------------------------------
def test_wire_sweeps(probe):
    g, (env, gain, osc, mix_a, mix_b, out) = chain(ADSR, MATH, OSC, MIX, MIX)
    g.connect(env, 0, gain, 0)
    g.connect(mix_a, 0, mix_b, 0)
    g.connect(mix_b, 0, mix_a, 0)
    g.connect(osc, SINE, out, 0)
    s = g.run(probe)["sweeps"]
    assert s[env] == 0 and s[gain] == 0 and s[osc] == 1 and s[mix_a] == 1   # an envelope holds, arithmetic on it too; a feedback cycle of arithmetic sweeps


# ---- the benchmarked workloads keep every fast form (config 4 apart: its loop) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", W.BENCH_WORKLOADS)
def test_benchmarked_workloads_keep_their_forms(probe, name):
    V = 4096
    B, build, overrides = W.bench_workload(name, V)
    g = Rec(48000, B, 2)
    ids = build(g)
    for m, f, v in overrides(ids):
        g.override(m, f, v)
    r = g.run(probe)
    if name.startswith("cfg4"):   # the FM pair's feedback loop runs through a pitch: the modulator is exact (test_a_loop_through_a_pitch_is_unbounded)
        assert not r["exact_patch"] and r["osc_exact"][ids["osc_m"]] == 1 and r["osc_exact"][ids["osc_c"]] == 0 and r["sine_loose"][ids["osc_c"]] == 1
        return
    assert not r["exact_patch"] and r["bound"] < BUDGET and sum(r["exact_blep"]) == 0 and sum(r["literal"]) == 0, r
    if name.startswith("cfg3") or name == "cfg2":
        assert r["saw_fixed"][ids["osc_a"]] == 1
    if name == "p4":
        assert r["nonlin_loose"][ids["shaper"]] == 1 and r["sine_loose"][ids["lfo"]] == 0


# ---- the numbers behind the ladder's epsilon, re-measured at a small size ---------------------------------------------------------------
@pytest.fixture(scope="module")
def calib():
    return _build(os.path.join(ROOT, "tests", "cpp", "ladder_calib"), [os.path.join(ROOT, "tools", "ladder_calib.c")], ["-O2", "-ffp-contract=off"])


def test_contracted_ladder_epsilon_covers_the_emulation(calib):
    """approx.cpp's kEpsLadder = (1.5e-6, 4.2e-6, 3.6e-6) for a cutoff that stands still or moves smoothly, twice that for rare jumps; a square or
    noise on the cutoff is outside any such figure (why such a filter has no contracted form)."""
    out = subprocess.run([calib, "own", "30000", "60"], capture_output=True, text=True, timeout=300).stdout
    rows = {l.split()[1]: [float(l.split()[k]) for k in (3, 5, 7)] for l in out.splitlines() if l.startswith("own")}
    for kind in ("none", "ramp", "sineLFO", "sine700", "saw"):
        assert all(e <= lim for e, lim in zip(rows[kind], (1.5e-6, 4.2e-6, 3.6e-6))), (kind, rows[kind])
    assert all(e <= 2 * lim for e, lim in zip(rows["squareLFO"], (1.5e-6, 4.2e-6, 3.6e-6)))
    assert max(rows["noise"]) > 1e-4


def test_an_overdriven_ladder_is_chaotic_where_its_cutoff_is_high(calib):
    """The reference clamps the ladder's stages, not its input (filter.rs:69-88).  An input above ~1.9 puts the stages into their clamps, flipping
    within a sample, and the last stage's cubic with the stage's own feedback around it is an expanding map: the literal ladder answers a 2.4e-7
    disturbance of its input with 1e3 .. 1e6 times that at LOW resonance — where the cutoff is high; below a cutoff of 0.4 the coefficients keep the
    last stage under the cubic's turning point whatever the drive.  (approx.cpp: kLadderDriveMax 1.75, kLadderTameCutoff 0.35; round 5's seeds
    105055 and 123042.)"""
    out = subprocess.run([calib, "amp", "48000", "60"], capture_output=True, text=True, timeout=300).stdout
    rows = {(float(l.split()[1]), l.split()[2]): (float(l.split()[4]), float(l.split()[6])) for l in out.splitlines() if l.startswith("amp")}
    for amp in (1.0, 1.5, 1.75):
        for kind in ("none", "ramp"):
            own, gain = rows[(amp, kind)]
            assert own <= 4.2e-6 and gain < 64.0, (amp, kind, own, gain)
    assert rows[(1.9, "ramp")][1] > 1e3 and rows[(3.0, "none")][1] > 1e3 and rows[(3.0, "none")][0] > 1e-4
    out = subprocess.run([calib, "tame", "48000", "60"], capture_output=True, text=True, timeout=300).stdout
    tame = {(float(l.split()[1]), float(l.split()[3])): float(l.split()[5]) for l in out.splitlines() if l.startswith("tame")}
    assert all(gain < 64.0 for (amp, cutoff), gain in tame.items() if cutoff <= 0.4), tame
    assert tame[(8.0, 0.5)] > 100.0


def test_a_ladder_behind_a_loud_mix(probe):
    """... and what the bound makes of it: a filter whose input can exceed 1.75 has no contracted form; where its cutoff can pass 0.35 its
    gains are unbounded and everything in front is exact.  One oscillator (|saw| <= 1) or a mix that stays below: every fast form."""
    def patch(gains, cutoff):
        g, ids = chain(OSC, OSC, OSC, MIX, VCF)
        a, b, c, mix, vcf, out = ids
        for k, (osc, gain) in enumerate(zip((a, b, c), gains)):
            g.set_field(osc, W.OSC_VAL, -1.0 - k)
            g.set_field(mix, W.MIX_GAIN0 + k, gain)
            g.connect(osc, SAW, mix, k)
        g.set_field(vcf, W.VCF_FREQ, cutoff)
        g.connect(mix, 0, vcf, 0)
        g.connect(vcf, 0, out, 0)
        return g, a, mix, vcf
    g, a, mix, vcf = patch((0.8, 0.5, 0.4), 0.6)          # sup 1.7
    r = g.run(probe)
    assert r["mag"][mix][0] == pytest.approx(1.7) and r["literal"][vcf] == 0 and r["exact_blep"][a] == 0 and r["bound"] < BUDGET
    g, a, mix, vcf = patch((1.0, 1.0, 1.0), 0.6)          # sup 3, cutoff 0.6: chaotic
    r = g.run(probe)
    assert r["literal"][vcf] == 1 and r["exact_blep"][a] == 1 and r["gain"][mix][0] == float("inf") and not r["exact_patch"]
    g, a, mix, vcf = patch((1.0, 1.0, 1.0), 0.2)          # sup 3, cutoff 0.2: the literal ladder, but a bounded one — the saws keep the f32 PolyBLEP
    r = g.run(probe)
    assert r["literal"][vcf] == 1 and r["exact_blep"][a] == 0 and r["gain"][mix][0] < 4.0 and r["bound"] < BUDGET


def test_noise_on_a_ladders_input(calib, probe):
    """A saw excites a ladder's resonance now and then, white noise all the time: with noise on the audio input the contracted form's error grows with
    the resonance — within 1.5 x the saw's figures up to 0.6 (3 x with an LFO square on the cutoff), 8.5e-6 / 1.2e-5 / 6.1e-6 from 0.8 up
    (tools/cpu_soak.py, noise family: seeds 235484, 227662, 239367 above their own bound).  The bound: a noise-like input (noise, a sample player,
    a reverb) costs 1.5 x, and above resonance 0.6 such a filter has no contracted form."""
    out = subprocess.run([calib, "noisein", "30000", "40"], capture_output=True, text=True, timeout=300).stdout
    rows = {(l.split()[1], float(l.split()[3])): [float(l.split()[k]) for k in (5, 7, 9)] for l in out.splitlines() if l.startswith("noisein")}
    eps = (1.5e-6, 4.2e-6, 3.6e-6)
    for (kind, res), e in rows.items():
        if res <= 0.5:   # (the bucket 0.5 is resonance 0.5 .. 0.6)
            assert all(x <= (3.0 if kind == "squareLFO" else 1.5) * lim for x, lim in zip(e, eps)), (kind, res, e)
    assert rows[("none", 0.8)][1] > 8e-6

    def patch(res):
        g, (noise, vcf, out_) = chain(NOISE, VCF)
        g.set_field(vcf, W.VCF_RES, res)
        g.connect(noise, 0, vcf, 0)
        g.connect(vcf, 0, out_, 0)
        return g, vcf
    g, vcf = patch(0.5)
    r = g.run(probe)
    assert r["literal"][vcf] == 0 and r["bound"] == pytest.approx(1.5 * 1.5e-6)
    g, vcf = patch(0.7)
    r = g.run(probe)
    assert r["literal"][vcf] == 1 and r["bound"] == 0.0


def test_raw_jumps_on_a_ladders_input_count_as_noise(probe):
    """A hard-synced oscillator resets mid-ramp, one without anti-aliasing has raw edges: jumps a band-limited saw does not have, at audio rate.
    tools/cpu_soak.py (seeds 226856, 405576): such saws into contracted ladders at resonance 0.79 / 0.91 came out at 5.2e-6 / 1.9e-6 against
    4.2e-6 / 1.5e-6.  They count as a noise-like input: 1.5 x up to resonance 0.6, no contracted form above."""
    def patch(res, synced=False, aa=True):
        g, (clock, osc, vcf, out_) = chain(OSC, OSC, VCF)
        g.set_field(vcf, W.VCF_RES, res)
        if not aa:
            g.set_field(osc, W.OSC_ANTIALIASING, 0)
        if synced:
            g.connect(clock, SAW, osc, 1)
        g.connect(osc, SAW, vcf, 0)
        g.connect(vcf, 0, out_, 0)
        return g, vcf
    for res, synced, aa, literal in ((0.8, False, True, 0), (0.8, True, True, 1), (0.8, False, False, 1), (0.5, True, True, 0), (0.5, False, False, 0)):
        g, vcf = patch(res, synced, aa)
        assert g.run(probe)["literal"][vcf] == literal, (res, synced, aa)


def test_a_synced_lfo_on_a_cutoff_jumps_at_the_sync_sources_rate(probe):
    """Round 5's soak, seed 66697 (200 voices x 6 000 samples): a 22 Hz saw — by its pitch an LFO, whose wraps are rare jumps — hard-synced by a
    filter's highpass and wired to a second filter's cutoff: every sync is a raw jump of the saw, at audio rate, and the second filter's
    contracted form came out at 4.5e-5 in 8 voices of 200.  An oscillator with its sync input connected moves like its sync source."""
    def patch(synced):
        g, (lfo, clock, osc, vcf, out) = chain(OSC, OSC, OSC, VCF)
        g.set_field(lfo, W.OSC_VAL, -4.3)
        g.connect(osc, SAW, vcf, 0)
        g.connect(lfo, SAW, vcf, 1)
        if synced:
            g.connect(clock, SAW, lfo, 1)
        g.connect(vcf, 0, out, 0)
        return g, lfo, vcf
    g, lfo, vcf = patch(False)
    assert g.run(probe)["literal"][vcf] == 0
    g, lfo, vcf = patch(True)
    assert g.run(probe)["literal"][vcf] == 1
