"""The oracle against everything the reference's own tests pin, and against its independent twin.

Reference tests replayed here (the only two that exist):
  * oscillator::dco_tests::produces_440   src/synth/oscillator.rs:284-305
  * synth::tests::topological_sort        src/synth.rs:537-613
Everything else is pinned by the C oracle and the NumPy restatement agreeing bit for bit.
"""
import random

import numpy as np
import pytest

from tests.npgraph import NumpyGraph


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ---- reference test 1: produces_440 ------------------------------------------------------------
def _produces_440(make):
    g = make(440 * 4, 17, 2)  # notice odd sized buffer
    m = g.add_module(1)
    g.module_calc(m)
    buf = g.get_output(m, 0)
    assert buf[0] == 0.0
    assert abs(buf[1] - 1.0) < 0.00001
    assert abs(buf[2]) < 0.00001
    assert abs(buf[3] + 1.0) < 0.00001
    assert abs(buf[4]) < 0.00001
    g.module_calc(m)
    buf = g.get_output(m, 0)
    assert abs(buf[0] - 1.0) < 0.00001  # should continue smoothly into next buffer


def test_produces_440_oracle(oracle):
    _produces_440(lambda sr, b, c: oracle.OraclePatch(sr, b, c))


def test_produces_440_numpy():
    class G(NumpyGraph):
        def module_calc(self, m):
            self.modules[m].calc()

        def get_output(self, m, port):
            return self.modules[m].outs[port]

    _produces_440(lambda sr, b, c: G(sr, b, c))


# ---- reference test 2: topological_sort --------------------------------------------------------
def _topo_graph(g):
    #     0 -> 1 -> 2 -> 3 -> o
    #      \----> 4 -----^
    #        5<->6^
    mods = [g.add_module(5) for _ in range(7)]
    out = g.add_module(0)
    free = {m: 0 for m in mods + [out]}

    def connect(src, sink):  # first unconnected input, synth.rs:523-535
        g.connect(src, 0, sink, free[sink])
        free[sink] += 1

    connect(mods[0], mods[1])
    connect(mods[1], mods[2])
    connect(mods[2], mods[3])
    connect(mods[3], out)
    connect(mods[0], mods[4])
    connect(mods[4], mods[3])
    connect(mods[6], mods[4])
    connect(mods[5], mods[6])
    connect(mods[6], mods[5])
    return mods, out


def _assert_topo(plan, mods, out):
    idx = {m: i for i, m in enumerate(plan)}
    assert len(plan) == 8
    assert idx[mods[0]] < idx[mods[1]] < idx[mods[2]] < idx[mods[3]] < idx[out]
    assert idx[mods[0]] < idx[mods[4]] < idx[mods[3]]
    assert idx[mods[6]] < idx[mods[4]]
    assert idx[mods[5]] < idx[mods[6]]


def test_topological_sort_oracle(oracle):
    g = oracle.OraclePatch(44100, 64, 2)
    mods, out = _topo_graph(g)
    rng = random.Random(1)
    for _ in range(1000):
        lst = mods + [out]
        rng.shuffle(lst)
        plan = g.plan(output=out, all_modules=lst)
        _assert_topo(plan, mods, out)
        # 6 is reached first from the output side, so the 6 -> 5 wire is the delayed one
        assert g.removed_edges() == [(mods[5], mods[6])]


def test_topological_sort_numpy_matches_oracle(oracle):
    from oracle import srack_numpy as N
    g = oracle.OraclePatch(44100, 64, 2)
    mods, out = _topo_graph(g)
    ng = NumpyGraph(44100, 64, 2)
    nmods, nout = _topo_graph(ng)
    rng = random.Random(2)
    for _ in range(200):
        lst = mods + [out]
        rng.shuffle(lst)
        plan = g.plan(output=out, all_modules=lst)
        nplan, nremoved = N.plan_execution(ng.modules[nout], [ng.modules[i] for i in lst])
        idx = {id(m): i for i, m in enumerate(ng.modules)}
        assert [idx[id(m)] for m in nplan] == plan
        assert [(idx[id(a)], idx[id(b)]) for a, b in nremoved] == g.removed_edges()


def test_planner_random_graphs_oracle_vs_numpy(oracle):
    """Random cyclic graphs: both planners give the same order and break the same wires."""
    from oracle import srack_numpy as N
    rng = random.Random(7)
    for trial in range(150):
        n = rng.randint(2, 9)
        g = oracle.OraclePatch(48000, 8, 2)
        ng = NumpyGraph(48000, 8, 2)
        types = [rng.choice([1, 2, 3, 4, 5, 6]) for _ in range(n)]
        pos_out = rng.randint(0, n)
        types.insert(pos_out, 0)
        for t in types:
            g.add_module(t)
            ng.add_module(t)
        n_in = {0: 2, 1: 2, 2: 2, 3: 1, 4: 2, 5: 4, 6: 2}
        n_out = {0: 0, 1: 3, 2: 3, 3: 1, 4: 1, 5: 1, 6: 1}
        for sink, t in enumerate(types):
            for port in range(n_in[t]):
                if rng.random() < 0.6:
                    src = rng.randrange(len(types))
                    if src == sink or n_out[types[src]] == 0:
                        continue
                    sp = rng.randrange(n_out[types[src]])
                    g.connect(src, sp, sink, port)
                    ng.connect(src, sp, sink, port)
        plan = g.plan()
        nplan, nremoved = ng.plan()
        assert plan == nplan, trial
        assert g.removed_edges() == nremoved, trial
        assert sorted(plan) == list(range(len(types)))  # every module scheduled exactly once


# ---- facts derived in SURVEY §8(c) -------------------------------------------------------------
def test_oscillator_known_values(oracle):
    g = oracle.OraclePatch(48000, 1024, 2)
    m = g.add_module(1)
    g.module_calc(m)
    saw, square = g.get_output(m, 2), g.get_output(m, 1)
    np.testing.assert_array_equal(saw[:4], np.array([0.0, -0.9816667, -0.9633333, -0.945], dtype=np.float32))
    assert square[0] == 0.0 and square[1] == -1.0
    # val = -8 => 1.71875 Hz; the square first goes > 0 at sample 13964
    g = oracle.OraclePatch(48000, 1024, 2)
    m = g.add_module(1)
    g.set_field(m, 0, -8.0)
    sq = []
    for _ in range(14):
        g.module_calc(m)
        sq.append(g.get_output(m, 1))
    sq = np.concatenate(sq)
    assert int(np.argmax(sq > 0)) == 13964


# ---- C oracle == NumPy restatement, bit for bit ------------------------------------------------
def _both(W, oracle, build, n, sr=48000, B=64, **kw):
    g = oracle.OraclePatch(sr, B, 2)
    ids = build(g, **kw)
    ng = NumpyGraph(sr, B, 2)
    build(ng, **kw)
    return g, ng, ids


@pytest.mark.parametrize("adsr", ["default", "finite"])
def test_p1_c_equals_numpy(W, oracle, adsr):
    # LFO at val=-2 (110 Hz) so several gate cycles fit in 4000 samples
    g, ng, ids = _both(W, oracle, W.build_p1, 4000, adsr=adsr, lfo_val=-2.0)
    for tap in [(ids["osc_a"], 2), (ids["osc_lfo"], 1), (ids["vcf"], 0), (ids["adsr"], 0)]:
        gg, nn, _ = _both(W, oracle, W.build_p1, 4000, adsr=adsr, lfo_val=-2.0)
        a, ta = gg.render(4000, tap=tap)
        b, tb = nn.render(4000, tap=tap)
        np.testing.assert_array_equal(bits(ta), bits(tb))
        np.testing.assert_array_equal(bits(a), bits(b))
    assert np.abs(g.render(4000)).max() > 0.05  # the patch actually sounds


@pytest.mark.parametrize("B", [1, 7, 64])
def test_p2_fm_feedback_c_equals_numpy(W, oracle, B):
    g, ng, ids = _both(W, oracle, W.build_p2, 1500, B=B, beta=0.3, index=1.0)
    assert g.plan() == ng.plan()[0]
    # the planner breaks OSC_M.sine -> MUL_FB: MUL_FB runs before OSC_M
    assert g.removed_edges() == [(ids["mul_fb"], ids["osc_m"])]
    a = g.render(1500)
    b = ng.render(1500)
    np.testing.assert_array_equal(bits(a), bits(b))
    assert np.abs(a).max() > 0.5


def test_all_module_types_c_equals_numpy(W, oracle):
    """A patch touching every port of every ★ module type, incl. sync, filter CV, HP/BP, mixer, math."""
    def build(g):
        lfo = g.add_module(1)      # 0: sync + CV source
        osc = g.add_module(1)      # 1: synced, CV-modulated, non-antialiased
        osc2 = g.add_module(1)     # 2
        vcf = g.add_module(2)      # 3
        adsr = g.add_module(3)     # 4
        vca = g.add_module(4)      # 5
        mix = g.add_module(5)      # 6
        sub = g.add_module(6)      # 7
        add = g.add_module(6)      # 8
        vca_neg = g.add_module(4)  # 9
        out = g.add_module(0)      # 10
        g.set_field(lfo, 0, -1.5)
        g.set_field(osc, 0, 0.25)
        g.set_field(osc, 1, 0)  # antialiasing off
        g.set_field(osc2, 0, 1.0 / 12.0)
        g.set_field(vcf, 0, 0.35)
        g.set_field(vcf, 1, 0.8)
        g.set_field(vcf, 2, 0.25)
        g.set_field(adsr, 0, 0.002)
        g.set_field(adsr, 1, 0.004)
        g.set_field(adsr, 2, 0.6)
        g.set_field(adsr, 3, 0.003)
        g.set_field(mix, 0, 0.5)
        g.set_field(mix, 2, 1.5)
        g.set_field(sub, 1, 1)  # Subtract
        g.set_field(sub, 0, 0.125)
        g.set_field(add, 0, -0.75)
        g.set_field(vca_neg, 0, 1)
        g.connect(lfo, 0, osc, 0)       # sine -> CV (vibrato)
        g.connect(lfo, 1, osc, 1)       # square -> sync
        g.connect(osc, 2, vcf, 0)       # saw -> filter
        g.connect(lfo, 2, vcf, 1)       # saw -> cutoff CV
        g.connect(lfo, 1, adsr, 0)      # gate
        g.connect(vcf, 1, vca, 0)       # bandpass -> VCA
        g.connect(adsr, 0, vca, 1)
        g.connect(vca, 0, mix, 0)
        g.connect(vcf, 2, mix, 2)       # highpass
        g.connect(osc2, 1, mix, 3)      # square
        g.connect(mix, 0, sub, 0)       # mix - constant
        g.connect(osc2, 0, add, 1)      # 0.0 + sine (in1 unconnected)
        g.connect(add, 0, vca_neg, 0)
        g.connect(lfo, 0, vca_neg, 1)   # negative CV passes when negative=true
        g.connect(sub, 0, out, 0)
        g.connect(vca_neg, 0, out, 1)
        return out

    g = oracle.OraclePatch(48000, 32, 2)
    build(g)
    ng = NumpyGraph(48000, 32, 2)
    build(ng)
    a, b = g.render(3000), ng.render(3000)
    np.testing.assert_array_equal(bits(a), bits(b))
    assert np.abs(a[0]).max() > 0.1 and np.abs(a[1]).max() > 0.1
    assert not np.array_equal(a[0], a[1])


def test_block_size_invariance_acyclic(W, oracle):
    """Acyclic graph: block-major evaluation is independent of buffer_size."""
    ref = None
    for B in (1, 17, 1024):
        g = oracle.OraclePatch(48000, B, 2)
        W.build_p1(g, lfo_val=-2.0)
        a = g.render(2500)
        if ref is None:
            ref = a
        np.testing.assert_array_equal(bits(a), bits(ref))


def test_feedback_delay_is_buffer_size(W, oracle):
    """Cyclic graph: the broken edge is a buffer_size-sample delay, so B is audible."""
    outs = {}
    for B in (1, 64):
        g = oracle.OraclePatch(48000, B, 2)
        W.build_p2(g)
        outs[B] = g.render(2000)
    assert not np.array_equal(outs[1], outs[64])
    # first B samples see zero feedback either way -> identical prefix of length 1
    assert outs[1][0, 0] == outs[64][0, 0]


def test_quirks(W, oracle):
    # VCA outputs zeros if either input is unconnected (vca.rs:127,142-144)
    g = oracle.OraclePatch(48000, 16, 2)
    osc, vca, out = g.add_module(1), g.add_module(4), g.add_module(0)
    g.connect(osc, 0, vca, 0)
    g.connect(vca, 0, out, 0)
    assert not g.render(64).any()
    # a gate already high at sample 0 is not an edge, but None->Attack is level-triggered
    g = oracle.OraclePatch(48000, 16, 2)
    const, adsr, out = g.add_module(6), g.add_module(3), g.add_module(0)
    g.set_field(const, 0, 1.0)  # 0.0 + 1.0
    g.connect(const, 0, adsr, 0)
    g.connect(adsr, 0, out, 0)
    a = g.render(8)[0]
    # default a_sec = 0: sample 0 enters Attack at phase 0 -> out = r_val = 0; sample 1: inf >= 1 -> Decay
    assert a[0] == 0.0 and a[1] == 1.0 and a[2] < 1.0
    # filter with freq = 0, res = 0 from the start never computes its coefficients (f stays 0)
    g = oracle.OraclePatch(48000, 16, 2)
    osc, vcf, out = g.add_module(1), g.add_module(2), g.add_module(0)
    g.set_field(vcf, 0, 0.0)
    g.set_field(vcf, 1, 0.0)
    g.connect(osc, 2, vcf, 0)
    g.connect(vcf, 0, out, 0)
    g.render(64)
    assert g.get_field(vcf, 3) == 0.0  # SRACK_VCF_ST_F
    # no OutputModule => empty plan, silence
    g = oracle.OraclePatch(48000, 16, 2)
    g.add_module(1)
    assert g.plan() == [] and not g.render(32).any()


def test_voice_uniform_matches_numpy_generator(W, oracle):
    for k in (0, 1):
        a = oracle.voice_uniform(W.SEED, 257, k, first_voice=1000)
        b = W.voice_uniform(257, k, W.SEED, first_voice=1000)
        np.testing.assert_array_equal(bits(a), bits(b))
        assert 0.0 <= a.min() and a.max() < 1.0


def test_render_batch_matches_single_renders(W, oracle):
    g = oracle.OraclePatch(48000, 64, 2)
    ids = W.build_p1(g, lfo_val=-2.0)
    det, cut = W.p1_voice_params(5)
    frames, mix = g.render_batch(5, 700, [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)],
                                 frames=True, mix=True, threads=3)
    for v in range(5):
        s = oracle.OraclePatch(48000, 64, 2)
        W.build_p1(s, lfo_val=-2.0)
        s.set_field(ids["osc_a"], W.OSC_VAL, det[v])
        s.set_field(ids["vcf"], W.VCF_FREQ, cut[v])
        np.testing.assert_array_equal(bits(s.render(700)), bits(frames[:, :, v]))
    np.testing.assert_allclose(mix, frames.astype(np.float64).sum(axis=2), rtol=0, atol=1e-12)


# ---- scope table (f) rank 1: the sequencers (no reference test pins them: double restatement only) ----------------
@pytest.mark.parametrize("B", [1, 64, 1024])
def test_p3_sequencers_c_equals_numpy(W, oracle, B):
    g, ng, ids = _both(W, oracle, W.build_p3, 9000, B=B)
    assert g.plan() == ng.plan()[0]
    for tap in [(ids["grid"], 0), (ids["grid"], 1), (ids["grid"], 2), (ids["pat"], 1), (ids["pat"], 8)]:
        gg, nn, _ = _both(W, oracle, W.build_p3, 9000, B=B)
        a, ta = gg.render(9000, tap=tap)
        b, tb = nn.render(9000, tap=tap)
        np.testing.assert_array_equal(bits(ta), bits(tb))
        np.testing.assert_array_equal(bits(a), bits(b))
    a = g.render(9000)
    assert np.abs(a[0]).max() > 0.1 and a[1].max() == 1.0 and a[1].min() == 0.0


def test_sequencer_semantics(oracle):
    """Stepping, wrap, sync reset, held CV over rests, gate = clock vs held (sequencer.rs:219-243)."""
    g = oracle.OraclePatch(48000, 8, 2)
    clock, grid, out = g.add_module(1), g.add_module(7), g.add_module(0)
    g.set_field(clock, 0, 3.0)           # 3520 Hz: a rising edge every ~13.6 samples
    g.set_field(grid, 2, 3)              # length 3
    g.set_step(grid, 0, 0, 1, 12)        # step 0: note 12 -> 1.0 V, gate follows the clock
    g.set_step(grid, 0, 2, 2, 6)         # step 2: note 6 -> 0.5 V, gate held
    g.connect(clock, 1, grid, 0)
    g.connect(grid, 0, out, 0)
    g.connect(grid, 1, out, 1)
    a = g.render(120)
    cv, gate = a[0], a[1]
    assert cv[0] == 1.0                                   # starts on step 0
    assert set(np.unique(cv)) == {np.float32(0.5), np.float32(1.0)}   # step 1 is a rest: holds 1.0, never 0
    assert gate.max() == 1.0 and (gate[cv == 0.5] == 1.0).all()       # held gate on step 2
    # default sequence (all None): CV stays at `last` = 0, gate 0, sync 1 only on step 0
    g = oracle.OraclePatch(48000, 8, 2)
    clock, pat, out = g.add_module(1), g.add_module(8), g.add_module(0)
    g.set_field(clock, 0, 3.0)
    g.set_field(pat, 0, 4)
    g.connect(clock, 1, pat, 0)
    g.connect(pat, 8, out, 0)
    g.connect(pat, 3, out, 1)
    a = g.render(200)
    assert a[0][0] == 1.0 and 0.2 < a[0].mean() < 0.3 and not a[1].any()


# ---- scope table (f) rank 4: NonLinear and Sample (no reference test pins them: double restatement only) -----------
@pytest.mark.parametrize("B", [1, 64, 1024])
def test_p4_sample_nonlinear_c_equals_numpy(W, oracle, B):
    g, ng, ids = _both(W, oracle, W.build_p4, 6000, B=B)
    assert g.plan() == ng.plan()[0]
    a, ta = g.render(6000, tap=(ids["smp"], 0))
    b, tb = ng.render(6000, tap=(ids["smp"], 0))
    np.testing.assert_array_equal(bits(ta), bits(tb))
    np.testing.assert_array_equal(bits(a), bits(b))
    assert np.abs(a[0]).max() > 0.3 and np.abs(a[1]).max() > 0.3 and not np.array_equal(a[0], a[1])


def test_sample_semantics(W, oracle):
    """Trigger, run-out, pitch and the empty WaveBox (sample.rs:206-238)."""
    wave = np.arange(1, 11, dtype=np.float32) / 16        # 10 samples, all different, none zero
    def make(rate=48000.0, clock=3.0, load=True):
        g = oracle.OraclePatch(48000, 16, 2)
        clk, smp, out = g.add_module(1), g.add_module(10), g.add_module(0)
        g.set_field(clk, 0, clock)
        if load:
            g.set_wave(smp, wave, rate)
        g.connect(clk, 1, smp, 0)
        g.connect(smp, 0, out, 0)
        g.connect(clk, 1, out, 1)
        return g, smp
    # same rate: one wave sample per tick after the first rising edge; when the wave runs out, sample 0 is held
    g, smp = make(clock=-1.0)            # 220 Hz: one rising edge every ~218 samples
    a = g.render(300)
    gate = a[1] > 0
    first = int(np.argmax(gate[1:] & ~gate[:-1])) + 1
    np.testing.assert_array_equal(a[0][:first], np.full(first, wave[0]))   # not playing yet: pos stays 0 -> samples[0]
    np.testing.assert_array_equal(a[0][first:first + 10], wave)
    np.testing.assert_array_equal(a[0][first + 10:first + 60], np.full(50, wave[0]))
    assert g.get_field(smp, W.SAMPLE_PLAYING) in (0.0, 1.0) and g.get_field(smp, W.SAMPLE_WAVE_NEW) == 0.0
    # half rate: every wave sample twice
    g, _ = make(rate=24000.0, clock=-1.0)
    a = g.render(300)
    np.testing.assert_array_equal(a[0][first:first + 20], np.repeat(wave, 2))
    # retrigger before the end restarts from 0
    g, _ = make(clock=6.0)               # 28160 Hz: edges every ~1.7 samples
    a = g.render(64)
    assert a[0].max() <= wave[3] and len(np.unique(a[0])) > 1
    # no wave loaded: silence
    g, _ = make(load=False)
    assert not g.render(64)[0].any()


def test_nonlinear_semantics(oracle):
    """Sign-preserving power, the unconnected-input cases and libm's edge cases (math.rs:203-205, 299-304)."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    g = oracle.OraclePatch(48000, 64, 2)
    osc, nl, nl0, out = g.add_module(1), g.add_module(9), g.add_module(9), g.add_module(0)
    g.set_field(nl, 0, 0.5)
    g.set_field(nl0, 0, 0.0)             # (None, None) with constant 0: -( (-0.0)^0 ) = -1
    g.connect(osc, 0, nl, 0)
    g.connect(nl, 0, out, 0)
    g.connect(nl0, 0, out, 1)
    a, s = g.render(64, tap=(osc, 0))
    want = np.array([libm.powf(float(x), 0.5) if x > 0 else -libm.powf(float(-x), 0.5) for x in s], dtype=np.float32)
    np.testing.assert_array_equal(bits(a[0]), bits(want))
    assert (np.sign(a[0]) == np.sign(s)).all() and (a[1] == -1.0).all()


def test_pow2_libm_formula_matches_glibc():
    """The device computes `2.0_f32.powf(cv)` (sample.rs:236) by restating glibc's powf for base 2: exact log2 step, then
    the exp2 kernel (32-entry table of rounded 2^(i/32), cubic, one rounding).  Same formula here, against libm itself."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    tab = (np.exp2(np.arange(32) / 32.0).view(np.uint64) - (np.arange(32, dtype=np.uint64) << np.uint64(47)))
    rng = np.random.default_rng(5)
    y = np.concatenate([rng.uniform(-20, 20, 200000), rng.uniform(-149.9, 127.99, 20000), [0.0, -0.0, 1.0, -126.0, -149.0, 127.0]]).astype(np.float32)
    xd = y.astype(np.float64)
    shift = float.fromhex("0x1.8p+52") / 32.0
    kd = xd + shift
    ki = kd.view(np.uint64)
    kd = kd - shift
    r = xd - kd
    with np.errstate(over="ignore"):
        s = (tab[(ki & np.uint64(31)).astype(np.int64)] + (ki << np.uint64(47))).view(np.float64)
    z = float.fromhex("0x1.c6af84b912394p-5") * r + float.fromhex("0x1.ebfce50fac4f3p-3")
    p = float.fromhex("0x1.62e42ff0c52d6p-1") * r + 1.0
    p = (z * (r * r) + p) * s
    mine = p.astype(np.float32)
    ref = np.array([libm.powf(2.0, float(v)) for v in y], dtype=np.float32)
    np.testing.assert_array_equal(bits(mine), bits(ref))


# ---- random patches: the two restatements agree bit for bit on graphs nobody hand-picked -----------------------------
@pytest.mark.parametrize("seed,noise", [(s, False) for s in range(80)] + [(s, True) for s in range(20)])
def test_random_patches_c_equals_numpy(oracle, seed, noise):
    from tests.fuzz_patches import random_patch
    B, build, overrides = random_patch(seed, noise)
    T = 500 if B < 1024 else 1300
    g = oracle.OraclePatch(48000, B, 2)
    ids = build(g)
    ng = NumpyGraph(48000, B, 2)
    build(ng)
    for m, f, fn in overrides:  # one voice: voice 3's draw
        v = float(fn(8)[3])
        g.set_field(ids[m], f, v)
        ng.set_field(ids[m], f, v)
    assert g.plan() == ng.plan()[0]
    a, b = g.render(T), ng.render(T)
    np.testing.assert_array_equal(bits(a), bits(b))


# ---- NoiseModule (oscillator.rs:308-393): the reference's draw is OS-seeded, so what is pinned is the map from 24 random bits
# to the sample, the published splitmix64 vectors for the bit source, and C == NumPy ---------------------------------------------
def _noise_patch(g, seed, voice):
    nz, vcf, out = g.add_module(11), g.add_module(2), g.add_module(0)
    g.connect(nz, 0, vcf, 0)
    g.connect(vcf, 0, out, 0)
    g.connect(nz, 0, out, 1)
    g.set_noise_seed(seed, voice)
    return nz


def test_splitmix64_published_vectors(oracle):
    from oracle.srack_numpy import splitmix64
    # Vigna's splitmix64.c seeded with 0: x += gamma; return mix(x) — the first four outputs
    want = [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F, 0xF88BB8A8724C81EC]
    assert [splitmix64((n * 0x9E3779B97F4A7C15) & (2**64 - 1)) for n in range(4)] == want
    # the C oracle's copy, through the one export that uses it: (sm(seed ^ (2 voice + k)) >> 40) * 2^-24
    assert oracle.lib().or_voice_uniform(0, 0, 0) == np.float32((want[0] >> 40) * 2.0 ** -24)


@pytest.mark.parametrize("seed,voice", [(0, 0), (1234, 7), (2**64 - 1, 2**40 + 3)])
def test_noise_c_equals_numpy_and_is_the_documented_stream(oracle, seed, voice):
    from oracle.srack_numpy import splitmix64
    B, T = 16, 400
    g, ng = oracle.OraclePatch(48000, B, 2), NumpyGraph(48000, B, 2)
    nz = _noise_patch(g, seed, voice)
    _noise_patch(ng, seed, voice)
    a, b = g.render(T), ng.render(T)
    np.testing.assert_array_equal(bits(a), bits(b))
    key = splitmix64(splitmix64(seed ^ splitmix64(nz)) ^ voice)
    want = [((splitmix64((key + n * 0x9E3779B97F4A7C15) % 2**64) >> 40) * 2.0 ** -24 - 0.5) * 2.0 for n in range(T)]
    np.testing.assert_array_equal(a[1], np.array(want, dtype=np.float32))  # channel 1 = the raw noise
    assert np.abs(a[0]).max() > 0.01                                        # channel 0 went through the ladder


def test_noise_distribution_matches_rand_standard_f32(oracle):
    """rand 0.8 Standard f32 = 24 uniform bits * 2^-24; (r - 0.5) * 2 => the 2^24 values k * 2^-23 - 1, equally likely."""
    g = oracle.OraclePatch(48000, 1024, 2)
    _noise_patch(g, 99, 0)
    x = g.render(1 << 18)[1].astype(np.float64)
    k = (x + 1.0) * 2.0 ** 23
    assert (k == np.round(k)).all() and k.min() >= 0 and k.max() < 2 ** 24 and x.min() >= -1.0 and x.max() < 1.0
    n = x.size
    assert abs(x.mean()) < 5 * (1 / np.sqrt(3)) / np.sqrt(n) and abs(x.var() - 1 / 3) < 0.005
    for lag in (1, 2, 7, 1024):
        assert abs(np.corrcoef(x[:-lag], x[lag:])[0, 1]) < 5 / np.sqrt(n)
    counts = np.bincount((k.astype(np.int64) >> 18), minlength=64)       # 64 equal bins: chi-square, 63 degrees of freedom
    chi2 = ((counts - n / 64) ** 2 / (n / 64)).sum()
    assert 25 < chi2 < 120
    low = np.bincount(k.astype(np.int64) & 63, minlength=64)              # and the LOW six of the 24 bits
    assert 25 < ((low - n / 64) ** 2 / (n / 64)).sum() < 120


def test_noise_voices_are_independent_streams_and_shard_by_first_voice(oracle):
    B, T, V = 32, 512, 6
    g = oracle.OraclePatch(48000, B, 2)
    _noise_patch(g, 5, 0)
    fr, _ = g.render_batch(V, T)
    assert len({fr[1, :, v].tobytes() for v in range(V)}) == V
    assert abs(np.corrcoef(fr[1, :, 0], fr[1, :, 1])[0, 1]) < 0.2
    h = oracle.OraclePatch(48000, B, 2)
    _noise_patch(h, 5, 4)                                                  # a shard that starts at global voice 4
    np.testing.assert_array_equal(h.render_batch(2, T)[0], fr[:, :, 4:6])
    k = oracle.OraclePatch(48000, B, 2)
    _noise_patch(k, 6, 0)                                                  # another seed: another stream
    assert not np.array_equal(k.render(T), g.render(T))


# ---- FreeverbModule (freeverb.rs): the module is in the tree, the freeverb crate 0.1.0 is not — restated, PARITY UNPINNED ------------
def _freeverb_patch(g, params=(), both=True):
    osc, osc2, fv, out = g.add_module(1), g.add_module(1), g.add_module(12), g.add_module(0)
    g.set_field(osc, 0, -1.0)
    g.set_field(osc2, 0, 0.37)
    g.connect(osc, 2, fv, 0)
    if both:
        g.connect(osc2, 1, fv, 1)
    g.connect(fv, 0, out, 0)
    g.connect(fv, 1, out, 1)
    for f, v in params:
        g.set_field(fv, f, v)
    return fv


@pytest.mark.parametrize("params,both", [((), True), (((0, 1.7), (2, 0.6), (3, 1.0), (4, 0.9), (5, 0.8)), True), (((1, 1), (5, 0.25)), False),
                                         (((3, 0.0), (4, 0.0), (0, 0.0)), True)])
def test_freeverb_c_equals_numpy(oracle, params, both):
    B, T = 64, 3000                       # past the longest comb (1760 + 25 samples at 48 kHz): feedback has come round
    g, ng = oracle.OraclePatch(48000, B, 2), NumpyGraph(48000, B, 2)
    _freeverb_patch(g, params, both)
    _freeverb_patch(ng, params, both)
    a, b = g.render(T), ng.render(T)
    np.testing.assert_array_equal(bits(a), bits(b))
    assert np.isfinite(a).all() and np.abs(a[:, 2000:]).max() > 1e-3


def test_freeverb_published_structure(oracle):
    """What Jezar's Freeverb is known to do, observed from outside: a dry-only setting is the identity on the fed channel, and
    the tank stays silent until the shortest comb (1116 samples; 1116 + 23 on the right) has come round."""
    B, sr = 32, 44100                      # at 44.1 kHz the tunings are the published ones unscaled
    # dry only: output = input * dry on the left, 0 on the right (input.1 = 0)
    g = oracle.OraclePatch(sr, B, 2)
    k, fv, out = g.add_module(6), g.add_module(12), g.add_module(0)
    g.set_field(k, 0, 0.5)
    g.connect(k, 0, fv, 0)
    g.connect(fv, 0, out, 0)
    g.connect(fv, 1, out, 1)
    g.set_field(fv, 2, 0.0)                # wet 0
    g.set_field(fv, 5, 0.75)               # dry
    y = g.render(3000)
    np.testing.assert_array_equal(y[0], np.full(3000, np.float32(0.5 * 0.75)))
    assert not y[1].any()
    # wet only, constant input 0.5: nothing comes out before the shortest comb (1116) has wrapped: the allpasses pass
    # -input + delayed of a zero comb sum, so the first non-zero sample is at n = 1116 on the left, 1139 on the right
    g = oracle.OraclePatch(sr, B, 2)
    k, fv, out = g.add_module(6), g.add_module(12), g.add_module(0)
    g.set_field(k, 0, 0.5)
    g.connect(k, 0, fv, 0)
    g.connect(fv, 0, out, 0)
    g.connect(fv, 1, out, 1)
    g.set_field(fv, 3, 1.0)                # width 1: wet_gains = (wet, 0): the channels do not mix
    y = g.render(1300)
    assert np.flatnonzero(y[0])[0] == 1116 and np.flatnonzero(y[1])[0] == 1116 + 23
    assert y[0][1116] == np.float32(0.5 * 0.015 * 3.0)   # the comb's first echo through four sign-flipping allpasses, times wet 1.0 * SCALE_WET


def _nan_gate_patch(g):
    # s = m + 1; m = 2 s (through the loop's one-block delay): 2, 6, 14, ... overflows to inf after ~128 blocks of one sample;
    # d = m - m is 0 until then and NaN afterwards; gate = d + 1: high, then NaN
    s_, m, d, gate, adsr, out = g.add_module(6), g.add_module(6), g.add_module(6), g.add_module(6), g.add_module(3), g.add_module(0)
    g.set_field(s_, 0, 1.0)
    g.connect(m, 0, s_, 0)
    g.set_field(m, 0, 2.0)
    g.set_field(m, 1, 2)              # MULTIPLY
    g.connect(s_, 0, m, 0)
    g.set_field(d, 1, 1)              # SUBTRACT
    g.connect(m, 0, d, 0)
    g.connect(m, 0, d, 1)
    g.set_field(gate, 0, 1.0)
    g.connect(d, 0, gate, 0)
    g.connect(gate, 0, adsr, 0)
    for f, v in ((0, 0.0005), (1, 0.0005), (2, 0.6), (3, 0.01)):
        g.set_field(adsr, f, v)
    g.connect(adsr, 0, out, 0)
    g.connect(gate, 0, out, 1)


def test_adsr_sustain_holds_on_a_nan_gate(oracle):
    """adsr.rs:175: Sustain leaves on `gate <= 0.0`, not on `!(gate > 0.0)`: a NaN gate (an overflowed patch) holds the level.
    Found by tools/fuzz_soak.py (seed 320 with sine ports): the NumPy twin and the GPU both had the shortcut, the C oracle did not."""
    g, ng = oracle.OraclePatch(48000, 1, 2), NumpyGraph(48000, 1, 2)
    _nan_gate_patch(g)
    _nan_gate_patch(ng)
    a, b = g.render(600), ng.render(600)
    np.testing.assert_array_equal(bits(a[0]), bits(b[0]))
    assert a[1][60] == 1.0 and np.isnan(a[1][200:]).all()          # the gate: high, then NaN
    assert a[0][100] == np.float32(0.6) and (a[0][200:] == np.float32(0.6)).all()   # sustain reached, and held through the NaNs


# ---- 2^cv of the exact render mode: the host libm's pow, operation for operation --------------------------------------------------
def test_exp2_libm_transliteration_is_the_hosts_pow():
    """`exp2_libm` (modules.hip.h) ports glibc's pow for the base 2.0 — the x86-64 FMA build, which contracts some of its products — so
    that the exact render mode's increments are the reference's to the last bit, misroundings included (pow is a 0.52-ulp function: one
    argument in 1300 is not the correctly rounded 2^e, tools/pow_misround.py).  Here: the same operations in Python with exact fused
    multiply-adds, against the pow of the libm this host runs (the one the oracle calls), over oscillator-like and wide arguments; the
    device itself is compared with the host's pow by tools/powcheck.hip and, end to end, by the fuzzer's chaotic seeds (-m gpu)."""
    import math
    import platform
    from tests import libm_pow2 as L
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the port follows glibc's x86-64 FMA build of pow")
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("a host without FMA runs glibc's other build of pow")
    tab = L.header_table()
    assert len(tab) == 256 and tab[0] == 0 and tab[1] == 0x3FF0000000000000
    rng = np.random.default_rng(7)
    args = np.concatenate([rng.uniform(-6, 6, 6000).astype(np.float32).astype(np.float64) + rng.uniform(-6, 3, 6000).astype(np.float32).astype(np.float64),
                           rng.uniform(-700, 700, 3000), rng.uniform(-1, 1, 1000) * 2.0 ** rng.integers(-80, 0, 1000), [0.0, -0.0, 1.0, -1.0, 0.5, 1e-300]])
    differ = [float(e) for e in args if L.exp2_libm(float(e), tab) != math.pow(2.0, float(e))]
    assert not differ, differ[:5]


# ---- a.powf(b) of NonLinearModule: the host libm's powf, operation for operation -------------------------------------------------------------
def test_powf_libm_transliteration_is_the_hosts_powf():
    """`powf_libm_plain` (modules.hip.h, round 6) ports glibc's powf — the x86-64 FMA build — so that a waveshaper's samples are the reference's
    to the last bit, misroundings included (powf is a 0.82-ulp function).  Here: the same operations in Python with exact fused multiply-adds and the
    DEVICE HEADER's tables, against the powf of the libm this host runs (the one the oracle calls): waveshaper-like arguments, the whole range of
    x (subnormals among them) with exponents of either sign, results that overflow, underflow and land on subnormals."""
    import ctypes
    import ctypes.util
    import platform
    import struct
    from tests import libm_powf as L
    if platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc":
        pytest.skip("the port follows glibc's x86-64 FMA build of powf")
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("a host without FMA runs glibc's other build of powf")
    libm = ctypes.CDLL(ctypes.util.find_library("m"))
    libm.powf.restype = ctypes.c_float
    libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    log2_tab, exp2f_tab = L.header_tables()
    assert log2_tab[9] == (0x3FF0000000000000, 0) and exp2f_tab[0] == 0x3FF0000000000000
    rng = np.random.default_rng(11)
    n = 6000
    xs = np.concatenate([rng.uniform(0, 2, n), 10.0 ** rng.uniform(-38, 38, n), rng.uniform(0, 1e-39, n // 4), np.abs(rng.normal(size=n)), [1.0, 2.0, 0.5, 1e-45, 3.4e38]]).astype(np.float32)
    ys = np.concatenate([rng.uniform(0.3, 3, n), rng.uniform(-4, 4, n), rng.uniform(-2, 2, n // 4), rng.choice([0.5, 2.0, 3.0, 0.75, 1.0, -1.0, 100.0, -100.0], n), [1.0, 0.5, 2.0, 1.0, 1.5]]).astype(np.float32)
    differ = []
    for x, y in zip(xs, ys):
        if not (x > 0 and np.isfinite(x)) or y == 0:
            continue
        got, want = L.powf_libm(float(x), float(y), log2_tab, exp2f_tab), float(libm.powf(float(x), float(y)))
        if struct.pack("<f", got) != struct.pack("<f", want):
            differ.append((float(x).hex(), float(y), got, want))
    assert not differ, differ[:5]
