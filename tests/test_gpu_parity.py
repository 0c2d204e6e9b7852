"""Parity of the HIP path with the CPU oracle, through the C ABI.  Needs a real MI355X (-m gpu).

Tolerance (BASELINE.json north_star: 1e-5 relative fp32; SURVEY §7 gives it a denominator):
    |gpu - ref| <= 1e-5 * max(|ref|, 1)      per sample.
The f32 modules are bit-exact by construction; with SRACK_RENDER_EXACT_OSC the oscillator's saw and
square are too, so chains that use only those ports are compared bit for bit in that mode.
"""
import os
import re

import numpy as np
import pytest

import srack_pkg

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-5


@pytest.fixture(scope="module")
def S():
    S = srack_pkg.load()
    assert S.device_count() > 0, "no GPU visible: the render path has no CPU fallback"
    return S


def assert_close(gpu, ref, tol=TOL):
    gpu, ref = np.asarray(gpu, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert gpu.shape == ref.shape
    err = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1.0)
    assert np.isfinite(gpu).all()
    assert err.max() <= tol, f"max rel err {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    return err.max()


def read_plane(S, d_ptr, T, V, pick, rows=512, workers=6):
    """One plane of frames [T][V] on the device, read back a slice of rows at a time: the sampled voices `pick`, and per row the f64 sum and
    the f64 sum of magnitudes over ALL voices (what the mix is held against).  The copies on this thread, the sums on a few others (numpy
    lets go of the interpreter lock while it sums): 50 GB take seconds."""
    import ctypes as C
    import queue
    from concurrent.futures import ThreadPoolExecutor
    got = np.empty((T, len(pick)), dtype=np.float32)
    own, scale = np.empty(T), np.empty(T)
    free = queue.Queue()
    for _ in range(workers + 1):
        free.put((np.empty((rows, V), dtype=np.float32), np.empty((rows, V), dtype=np.float32)))

    def sums(pair, t0, n):
        buf, tmp = pair
        got[t0:t0 + n] = buf[:n, pick]
        own[t0:t0 + n] = buf[:n].sum(axis=1, dtype=np.float64)
        np.abs(buf[:n], out=tmp[:n])
        scale[t0:t0 + n] = tmp[:n].sum(axis=1, dtype=np.float64)
        free.put(pair)

    with ThreadPoolExecutor(workers) as pool:
        jobs = []
        for t0 in range(0, T, rows):
            n = min(rows, T - t0)
            pair = free.get()
            assert S.lib.srack_device_to_host(pair[0].ctypes.data_as(C.c_void_p), C.c_void_p(d_ptr + t0 * V * 4), n * V * 4, None) == 0
            assert S.lib.srack_device_sync(None) == 0
            jobs.append(pool.submit(sums, pair, t0, n))
        for j in jobs:
            j.result()
    return got, own, scale


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# render flags: 1 exact oscillator, 2 no fused kernels (tile interpreter), 4 no uniform hoisting (everything per voice)
#               32 the general path through a kernel specialised for the program at run time (with 2: instead of a fused kernel)
MODES_BASE = [pytest.param(0, id="fused"), pytest.param(2, id="interp"), pytest.param(1, id="fused-exact"), pytest.param(3, id="interp-exact"),
              pytest.param(4, id="fused-nohoist"), pytest.param(6, id="interp-nohoist"), pytest.param(5, id="fused-exact-nohoist")]
MODES = MODES_BASE + [pytest.param(34, id="special"), pytest.param(35, id="special-exact"), pytest.param(38, id="special-nohoist"),
                      pytest.param(39, id="special-exact-nohoist")]
FAST_MODES = [pytest.param(0, id="fused"), pytest.param(2, id="interp"), pytest.param(4, id="fused-nohoist"), pytest.param(6, id="interp-nohoist"),
              pytest.param(34, id="special"), pytest.param(38, id="special-nohoist")]


# ---- the reference's oscillator test, through the whole GPU path --------------------------------
@pytest.mark.parametrize("flags", [0, 1])
def test_produces_440(S, flags):
    """oscillator::dco_tests::produces_440 (oscillator.rs:284-305): sr 1760, B 17, two calc() calls."""
    p = S.Patch(440 * 4, 17, 2)
    osc, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_OUTPUT)
    p.connect(osc, S.OSC_OUT_SINE, out, 0)
    p.configure_voices(1)
    buf = p.render_channels(17, flags)[0, :, 0]
    assert buf[0] == 0.0
    assert abs(buf[1] - 1.0) < 0.00001
    assert abs(buf[2]) < 0.00001
    assert abs(buf[3] + 1.0) < 0.00001
    assert abs(buf[4]) < 0.00001
    buf = p.render_channels(17, flags)[0, :, 0]  # second block continues from the carried phase
    assert abs(buf[0] - 1.0) < 0.00001


# ---- cfg1 golden: 1 voice, P1, 1 s -----------------------------------------------------------------
@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("adsr", ["default", "finite"])
def test_cfg1_golden(S, adsr, flags):
    gold = np.load(os.path.join(GOLD, f"cfg1_p1_{adsr}.npz"))["audio"]
    p = S.Patch(48000, 1024, 2)
    S.build_p1(p, adsr=adsr)
    p.configure_voices(1)
    out = p.render_channels(48000, flags)
    assert ("fused=1" in p.info()) == (not flags & 2)  # one voice: nothing is hoisted
    assert ("kernel=render_interp" in p.info()) == (flags & 34 == 2) and ("kernel=render_specialized" in p.info()) == (flags & 34 == 34)
    if flags & 1:  # exact oscillator: saw and square are pure f64 arithmetic => the whole chain is bit-identical
        np.testing.assert_array_equal(bits(out[0, :, 0]), bits(gold))
    else:
        assert_close(out[0, :, 0], gold)
    np.testing.assert_array_equal(out[0], out[1])
    assert not out[0, :13964].any() and np.abs(out[0, 13964:]).max() > 0.1


# ---- cfg3 golden + fresh oracle draw: per-voice detune / cutoff -----------------------------------------
@pytest.mark.parametrize("flags", MODES)
def test_cfg3_golden_voices(S, flags):
    z = np.load(os.path.join(GOLD, "cfg3_p1_voices8.npz"))
    gold = z["audio"]
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p, lfo_val=float(z["lfo_val"]))
    p.configure_voices(8)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, z["detune"])
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, z["cutoff"])
    out = p.render_channels(gold.shape[0], flags)[0]
    if flags & 1:
        np.testing.assert_array_equal(bits(out), bits(gold))
    else:
        assert_close(out, gold)


@pytest.mark.parametrize("flags", FAST_MODES)
@pytest.mark.parametrize("V", [1, 63, 64, 65, 300])
def test_p1_voices_vs_oracle_and_mix(S, oracle, V, flags):
    T = 6000
    det, cut = S.p1_voice_params(V, first_voice=12345)
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = S.build_p1(o, adsr="finite", lfo_val=-3.0)
    ref, ref_mix = o.render_batch(V, T, [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)], mix=True, threads=8)
    p = S.Patch(48000, 1024, 2)
    S.build_p1(p, adsr="finite", lfo_val=-3.0)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    fr, mix = p.render(T, frames=True, mix=True, flags=flags)
    assert fr.shape == (1, T, V)
    want = "render_specialized" if flags & 34 == 34 else "render_interp" if flags & 2 else ("render_voice_chain" if (flags & 4 or V == 1) else "render_voice_chain_track")
    assert p.info().endswith("kernel=" + want), p.info()
    assert_close(fr[0], ref[0])
    # mix-down: sum over voices, checked against the f64 sum of the GPU's own frames and of the oracle's
    own = fr[0].astype(np.float64).sum(axis=1)
    scale = np.abs(fr[0].astype(np.float64)).sum(axis=1)
    assert (np.abs(mix[0] - own) <= 1e-5 * np.maximum(scale, 1.0)).all()
    assert (np.abs(mix[0] - ref_mix[0]) <= 2e-5 * np.maximum(scale, 1.0)).all()
    np.testing.assert_array_equal(mix[0], mix[1])


@pytest.mark.parametrize("lfo_val", [pytest.param(-8.0, id="as-benchmarked"), pytest.param(-2.0, id="fast-gate")])
def test_cfg2_identical_voices(S, oracle, lfo_val):
    """config 2 at full size — 4096 identical voices, the whole second (BASELINE's gate LFO at 1.72 Hz first rises at sample 13 964; a
    110 Hz gate beside it keeps the envelope busy throughout): every column equals the 1-voice oracle render, the kernel is the one
    `bench.py` times for config 2 (five control units + a broadcast)."""
    T, V = 48000, 4096
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o, lfo_val=lfo_val)
    ref = np.concatenate([o.render(1024)[0] for _ in range(47)])[:T]   # 47 ticks of buffer_size samples, truncated: SURVEY 8(a1)
    p = S.Patch(48000, 1024, 2)
    S.build_p1(p, lfo_val=lfo_val)
    p.configure_voices(V)
    fr, mix = p.render(T)
    assert "kernel=render_specialized" in p.info() and p.info().count("ctl[") == 5, p.info()
    assert (fr[0] == fr[0][:, :1]).all()
    assert_close(fr[0][:, 0], ref)
    assert_close(mix[0] / V, ref, tol=2e-5)
    assert np.abs(ref).max() > 0.05


# ---- control units evaluated across lanes (voice-invariant modules of a specialised kernel) --------------------------------
@pytest.mark.parametrize("gate_port", [1, 2, 0])                                # square, saw, sine as the envelope's gate
@pytest.mark.parametrize("env", [(0.0, 0.0, 0.5, 0.0), (1e-4, 2e-4, 0.3, 1e-4), (0.01, 0.1, 0.5, 0.2), (5.0, 5.0, 1.0, 5.0)])
@pytest.mark.parametrize("lfo_val", [-2.0, 3.0, 5.4])
def test_control_units_across_lanes(S, oracle, lfo_val, env, gate_port):
    """Identical voices: every module is voice-invariant and lands in a co-scheduled control unit, where lane j evaluates sample j of a
    tile — the oscillators as a phase recurrence + per-lane outputs, the envelope in runs between the events that can end its segment,
    the VCA across lanes, the filter sample by sample.  Envelopes with zero-length segments (infinite increments: a segment per sample),
    very short and very long ones; a gate that toggles every few samples (3520 Hz) and one above a quarter cycle per sample (the
    oscillator then has no tile-wise form and its unit falls back); a render length with a ragged last tile, and the same render split at
    a sample that is no tile border.  Exact mode bit for bit, default mode inside the contract."""
    T, V = 5000 + 13, 70
    def build(g):
        ids = S.build_p1(g, lfo_val=lfo_val)
        for f, v in zip((S.ADSR_A_SEC, S.ADSR_D_SEC, S.ADSR_S_VAL, S.ADSR_R_SEC), env):
            g.set_field(ids["adsr"], f, v)
        g.disconnect(ids["adsr"], 0)
        g.connect(ids["osc_lfo"], gate_port, ids["adsr"], 0)
        return ids
    o = oracle.OraclePatch(48000, 1024, 2)
    build(o)
    ref = o.render(T)[0]
    assert np.abs(ref).max() > 1e-3
    for flags in (33, 32):
        p = S.Patch(48000, 1024, 2)
        build(p)
        p.configure_voices(V)
        out = p.render_channels(T, flags)[0]
        assert "kernel=render_specialized" in p.info() and "ctl[" in p.info(), p.info()
        assert (out == out[:, :1]).all()
        if flags & 1:
            np.testing.assert_array_equal(bits(out[:, 0]), bits(ref))
        elif gate_port == 1:   # (a saw or sine gate is a threshold on an approximated value: default mode may move an edge by a sample)
            assert_close(out[:, 0], ref)
        p.configure_voices(V)
        a = p.render_channels(1777, flags)[0]
        b = p.render_channels(T - 1777, flags)[0]
        np.testing.assert_array_equal(bits(np.concatenate([a, b])), bits(out))


# ---- FM patch with a feedback edge (config 4) ---------------------------------------------------------
@pytest.mark.parametrize("flags", [0, 1, 64])   # (0: since round 5 the modulator is exact as a whole — a loop through a pitch, approx.cpp; 64 = KEEP_DEFAULT: the fast kernels)
@pytest.mark.parametrize("B", [1, 7, 16, 64, 1024])
def test_p2_feedback_vs_oracle(S, oracle, B, flags):
    T, V = 3000, 70
    beta, index = S.p2_voice_params(V)
    o = oracle.OraclePatch(48000, B, 2)
    ids = S.build_p2(o)
    ref, _ = o.render_batch(V, T, [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)], threads=8)
    p = S.Patch(48000, B, 2)
    S.build_p2(p)
    p.configure_voices(V)
    p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta)
    p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
    assert p.delayed_edges() == [(ids["osc_m"], 0, ids["mul_fb"], 0)]
    out = p.render_channels(T, flags)
    assert_close(out[0], ref[0])
    assert np.abs(out[0]).max() > 0.9
    assert ("; exact osc 0]" in p.info()) == (flags == 0) and ("approx[kept default" in p.info()) == (flags == 64), p.info()


@pytest.mark.parametrize("flags", [0, 1, 2])
@pytest.mark.parametrize("B", [1, 64, 1024])
def test_fm_pair_sample_loops(S, oracle, B, flags, monkeypatch):
    """(With SRACK_RENDER_KEEP_DEFAULT: since round 4 the flattener gives a patch whose feedback loop can AMPLIFY — here feedback gains
    above 1 — the exact flavour, see test_fm_feedback_gain_above_one_takes_the_exact_flavour; this test keeps the default-mode kernels'
    per-wave classes covered with the draw that reaches all of them.)
    The fused FM kernels choose their sample loop per wave from what the wave can prove about its 64 voices (default mode): both
    CVs within 1/2 (no range reduction in 2^x, one-instruction phase wrap), the carrier's within two octaves ((2^(cv/4))^4), beyond
    (range reduction), or nothing at all — a voice whose feedback gain lets 2^x overflow, after which its phase is NaN as in the reference, and its 63
    well-behaved neighbours, which take the literal forms with it.  One wave of each, plus a modulator `val` that moves a wave from
    one class to the next; every voice against the oracle."""
    # (buffer_size 1024: the time-parallel pair takes calls of 4096 samples and more — shorter ones keep the ring kernel, see below)
    flags |= S.RENDER_KEEP_DEFAULT
    T, W, cut = (9000, 64, 4501) if B == 1024 else (2500, 64, 1001)
    rng = np.random.default_rng(11)
    beta = np.concatenate([rng.uniform(0.05, 0.45, W), rng.uniform(0.1, 0.4, W), rng.uniform(0.6, 1.8, W), rng.uniform(0.1, 0.4, W),
                           rng.uniform(0.1, 0.3, W), rng.uniform(0.6, 1.8, W), rng.uniform(0.6, 1.8, W), rng.uniform(0.1, 0.4, W)]).astype(np.float32)
    index = np.concatenate([rng.uniform(0.05, 0.45, W), rng.uniform(0.5, 1.5, W), rng.uniform(0.5, 2.5, W), rng.uniform(0.5, 1.5, W),
                            rng.uniform(0.1, 0.3, W), rng.uniform(0.1, 0.4, W), rng.uniform(0.6, 1.9, W), rng.uniform(2.1, 3.0, W)]).astype(np.float32)
    # (waves 5 - 7: the remaining pairings of the per-oscillator classes — modulator with a range reduction beside a carrier without one or
    # with the two-octave form, and a small modulator beside a carrier whose index passes two octaves)
    val_m = np.zeros(8 * W, np.float32)
    val_m[4 * W:5 * W] = rng.uniform(0.25, 0.6, W)      # a per-voice pitch offset: folded into the voice's scale in the proved loops
    beta[3 * W + 17] = 3.0e4                             # 2^(30000 sin) overflows on the second sample (1500 would not at buffer_size 1: a huge
                                                         # finite increment is an integer, the phase lands on 0 and the sine starts over)
    V = 8 * W
    o = oracle.OraclePatch(48000, B, 2)
    ids = S.build_p2(o)
    over = [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index), (ids["osc_m"], S.OSC_VAL, val_m)]
    ref, _ = o.render_batch(V, T, over, threads=8)
    p = S.Patch(48000, B, 2)
    S.build_p2(p)
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    out = p.render_channels(T, flags)
    if not (flags & 2):
        assert "kernel=render_fm_pair" in p.info(), p.info()
    bad = 3 * W + 17
    assert np.isnan(ref[0][:, bad]).any() and np.isnan(ref[0][-1, bad])
    np.testing.assert_array_equal(np.isnan(out[0]), np.isnan(ref[0]))
    ok = ~np.isnan(ref[0])
    err = np.abs(out[0].astype(np.float64) - ref[0]) / np.maximum(np.abs(ref[0]), 1.0)
    # (the contract's bar over the first 2500 samples, as for the other buffer sizes; the 9000-sample render the time-parallel pair needs
    # — two calls of 4096 samples and more — gives this draw's strongest feedback gains, 2^(1.8 sin) on their own pitch, time to integrate
    # a last-bit difference up: 1.8e-5 at the end, where config 4's own draw stays below 3e-6 for the whole second)
    e = np.where(ok, err, 0.0)
    assert e[:2500].max() <= TOL and e.max() <= 1e-4, f"max rel err {e[:2500].max():.3e} / {e.max():.3e} at {np.unravel_index(e.argmax(), e.shape)}"
    # and split in two calls at a point that is no tile border: the proofs are per launch / per tile, the state carries over bit for bit
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    a = p.render_channels(cut, flags)
    b = p.render_channels(T - cut, flags)
    if B == 1024 and flags == 0:
        assert "kernel=render_fm_pair_block" in p.info(), p.info()
    if "kernel=render_fm_pair_block" in p.info():
        # the time-parallel pair (buffer_size 256 ... 1024, default mode): a phase is a prefix sum over 256-sample chunks counted from the
        # launch's first sample, so a split render adds the same increments in other groups — 1e-16 in a phase, an occasional last bit of an
        # f32 sample, never a different NaN
        ab = np.concatenate([a[0], b[0]])
        np.testing.assert_array_equal(np.isnan(ab), np.isnan(out[0]))
        fin = ~np.isnan(ab)
        assert np.abs(ab[fin].astype(np.float64) - out[0][fin]).max() <= 4e-7 and (bits(ab) != bits(out[0])).mean() < 1e-3
    else:
        np.testing.assert_array_equal(bits(np.concatenate([a[0], b[0]])), bits(out[0]))


@pytest.mark.parametrize("B", [1, 1024])
def test_fm_feedback_gain_above_one_takes_the_exact_flavour(S, oracle, B):
    """A cycle that can amplify iterates whatever approximation enters it: FM feedback with a gain above 1 (2^(1.8 sin) on its own pitch)
    let the default mode drift to 1.8e-5 within 9000 samples.  Since round 5 ANY loop through a pitch has its oscillator evaluated exactly as
    a whole (approx.cpp: the loop's gain has no bound): the modulator follows the reference bit for bit, the carrier behind it keeps the
    default forms; config 4's own draw (gains up to 0.4) renders the same way, its fast kernels under SRACK_RENDER_KEEP_DEFAULT."""
    V, T = 128, 6000
    rng = np.random.default_rng(5)
    beta, index = rng.uniform(0.6, 1.8, V).astype(np.float32), rng.uniform(0.5, 1.5, V).astype(np.float32)
    o = oracle.OraclePatch(48000, B, 2)
    ids = S.build_p2(o)
    over = [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)]
    ref, _ = o.render_batch(V, T, over, threads=8)
    p = S.Patch(48000, B, 2)
    S.build_p2(p)
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    out = p.render_channels(T, 0)
    assert assert_close(out[0], ref[0]) < 5e-7   # (the carrier's sine in f32 after the exact fold, on a phase that is the reference's to 1e-12)
    q = S.Patch(48000, B, 2)
    S.build_p2(q)
    q.configure_voices(V)
    b2, i2 = S.p2_voice_params(V)
    q.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, b2)
    q.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, i2)
    q.render_channels(4096, S.RENDER_KEEP_DEFAULT)   # (round 5: config 4's own draw goes exact by default as well — any loop through a pitch; its fast kernels on request)
    assert ("render_fm_pair_block" in q.info()) if B == 1024 else ("render_fm_pair" in q.info() or "render_specialized" in q.info())
    # the modulator exact as a whole: the general path, or — from round 6, where the delay allows time lanes — the pair's kernel for exactly that program
    assert ("kernel=render_fm_pair_block_x" if B == 1024 else "kernel=render_fm_pair_x") in p.info() and "; exact osc 0]" in p.info(), p.info()


@pytest.mark.parametrize("B,T", [(256, 9216), (1000, 9000), (1024, 9216), (1024, 8191), (640, 4097), (1, 4096), (1, 2501)])
def test_fm_pair_with_the_modulator_exact_across_time_lanes(S, oracle, B, T):
    """render_fm_pair_block_x (round 6): config 4's program as default mode renders it — the modulator exact as a whole, the carrier in its default
    forms — with a delay of 256 ... 1024 samples: increments (the libm's 2^cv, the correctly rounded quotient) and sines across time lanes, only
    `pos = (pos + delta) % 1.0` serial (oscillator.rs:152-153).  70 voices (two full workgroups of 32 and a ragged one), ring lengths that are and
    are not multiples of the 64-sample chunk, render lengths that are and are not.  The frames against the oracle (the carrier's f32 sine on a phase
    that is the reference's to 1e-12: 5e-7), the MODULATOR's phase after the render bit for bit, a second call continuing the first, and a voice
    whose feedback gain overflows 2^cv (NaNs where the reference has them: the scan's fmod1 path).  buffer_size 1: render_fm_pair_x, the same program
    with the fed-back sine in a register — nothing across time there, the same exact forms per sample."""
    V = 70
    beta, index = S.p2_voice_params(V)
    beta, index = beta.copy(), index.copy()
    wild = 41
    beta[wild] = 3.0e4   # 2^(30000 sin) overflows: the increment is inf, the phase NaN from there on — in that voice only
    over = lambda ids: [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)]
    o = oracle.OraclePatch(48000, B, 2)
    ids = S.build_p2(o)
    ref, _ = o.render_batch(V, 2 * T, over(ids), threads=8)
    p = S.Patch(48000, B, 2)
    S.build_p2(p)
    p.configure_voices(V)
    for m, f, v in over(ids):
        p.set_voice_field(m, f, v)
    a = p.render_channels(T, 0)
    assert ("kernel=render_fm_pair_x" if B == 1 else "kernel=render_fm_pair_block_x") in p.info() and "; exact osc 0]" in p.info(), p.info()
    if T % B == 0:   # the oracle ticks block by block: its modules hold the state after whole blocks
        pos = p.get_voice_field(ids["osc_m"], S.OSC_POS)
        for v in (0, 31, 32, 63, 64, 69, wild):
            q = oracle.OraclePatch(48000, B, 2)
            S.build_p2(q)
            q.set_field(ids["mul_fb"], S.MATH_CONSTANT, float(beta[v]))
            q.set_field(ids["mul_idx"], S.MATH_CONSTANT, float(index[v]))
            q.render(T)
            want = q.get_field(ids["osc_m"], S.OSC_POS)
            assert np.float64(pos[v]).view(np.uint64) == np.float64(want).view(np.uint64) or (np.isnan(pos[v]) and np.isnan(want)), (v, pos[v], want)
    b = p.render_channels(T, 0)
    out = np.concatenate([a[0], b[0]])
    assert np.isnan(ref[0][:, wild]).any() and not np.isnan(np.delete(ref[0], wild, axis=1)).any()
    np.testing.assert_array_equal(np.isnan(out), np.isnan(ref[0]))
    ok = ~np.isnan(ref[0])
    err = np.where(ok, np.abs(out.astype(np.float64) - ref[0]) / np.maximum(np.abs(ref[0]), 1.0), 0.0)
    assert err.max() < 5e-7, (err.max(), np.unravel_index(err.argmax(), err.shape))
    assert np.abs(out[ok]).max() > 0.9
    # the general path (flags 2: no fusion) renders the same program — the same exact modulator, the carrier through osc_step: within the carrier's 4e-7
    g2 = S.Patch(48000, B, 2)
    S.build_p2(g2)
    g2.configure_voices(V)
    for m, f, v in over(ids):
        g2.set_voice_field(m, f, v)
    c = g2.render_channels(T, 2)
    assert "render_fm_pair_block_x" not in g2.info() and "render_fm_pair_x" not in g2.info()
    fin = ~np.isnan(c[0])
    assert np.abs(c[0][fin].astype(np.float64) - a[0][fin]).max() < 8e-7


@pytest.mark.parametrize("seed", [78, 87] + list(range(0, 24)))
def test_fm_x_soak_seeds(S, oracle, seed):
    """tools/fm_x_soak.py's draw — random sample rates, ring lengths (any), voice counts, render lengths, a second call, per-voice gains / indices /
    pitches / initial phases, now and then a voice whose 2^cv overflows or a carrier outside every bounded class — through round 6's FM kernels.
    Seeds 78 and 87 are its finds: the odd slice of a wave's pair started behind `pair - mine`, and a voice whose modulator had just overflowed
    (a NaN in its own slice's total) turned the two or three samples BEFORE the overflow into NaNs too."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fm_x_soak
    k, nan_ok, err, state_bad, info = fm_x_soak.one(seed, S, oracle)
    assert k in ("render_fm_pair_x", "render_fm_pair_block_x"), info
    assert nan_ok and err <= 1e-5 and state_bad == 0, (nan_ok, err, state_bad, info)


def _envelope_fm(g, S):
    """FM with an envelope on the index: gate LFO -> ADSR -> VCA.cv, modulator sine -> VCA.audio -> x index -> carrier pitch."""
    lfo, env, om, vca, idx, oc, out = (g.add_module(t) for t in (S.MOD_OSCILLATOR, S.MOD_ADSR, S.MOD_OSCILLATOR, S.MOD_VCA, S.MOD_MATH, S.MOD_OSCILLATOR, S.MOD_OUTPUT))
    g.set_field(lfo, S.OSC_VAL, -4.0)   # 27.5 Hz: a gate edge every 1745 samples
    for f, v in zip((S.ADSR_A_SEC, S.ADSR_D_SEC, S.ADSR_S_VAL, S.ADSR_R_SEC), (0.002, 0.01, 0.4, 0.01)):
        g.set_field(env, f, v)
    g.set_field(om, S.OSC_VAL, 1.0)
    g.set_field(idx, S.MATH_OPERATION, 2)
    g.set_field(idx, S.MATH_CONSTANT, 1.5)
    for a, ap, b, bp in ((lfo, 1, env, 0), (om, 0, vca, 0), (env, 0, vca, 1), (vca, 0, idx, 0), (idx, 0, oc, 0), (oc, 0, out, 0), (oc, 0, out, 1)):
        g.connect(a, ap, b, bp)
    return dict(env=env, om=om, idx=idx)


@pytest.mark.parametrize("flags", [0, 2, 4, 34, 38, 35])
def test_envelope_scaled_fm_index(S, oracle, flags):
    """An oscillator whose pitch CV is sine x envelope x index: the kernel generator bounds it by the envelope's hull (adsr_bound) and
    versions the carrier per wave — indices 0.1 ... 2.6 put the waves of this render in all three classes, a sustain level of 3 in one
    voice raises that wave's bound past the (2^(cv/4))^4 form, and a negative attack time in another sends its wave to the literal forms."""
    V, T = 256, 6000
    index = np.linspace(0.1, 2.6, V).astype(np.float32)
    sustain = np.linspace(0.1, 0.9, V).astype(np.float32)
    sustain[70] = 3.0
    attack = np.full(V, 0.002, np.float32)
    attack[200] = -1.0   # (-2e-5 per sample: the attack phase drifts below 0 — mildly, the render stays comparable — and adsr_tame says no)
    detune = np.linspace(0.9, 1.1, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = _envelope_fm(o, S)
    over = [(ids["idx"], S.MATH_CONSTANT, index), (ids["env"], S.ADSR_S_VAL, sustain), (ids["env"], S.ADSR_A_SEC, attack), (ids["om"], S.OSC_VAL, detune)]
    ref, _ = o.render_batch(V, T, over, threads=8)
    p = S.Patch(48000, 1024, 2)
    _envelope_fm(p, S)
    p.configure_voices(V)
    for m, f, v in over:
        p.set_voice_field(m, f, v)
    out = p.render_channels(T, flags)
    if flags & 32:
        assert "kernel=render_specialized" in p.info()
    assert np.abs(ref[0]).max() > 0.9 and np.isfinite(ref).all()
    if flags & 1:
        np.testing.assert_array_equal(bits(out[0]), bits(ref[0]))   # exact mode proves nothing and is bit-identical (the sine port: within tolerance)
    else:
        assert_close(out[0], ref[0])


@pytest.mark.parametrize("B", [1, 1024])
def test_cfg4_golden(S, B):
    z = np.load(os.path.join(GOLD, f"cfg4_p2_b{B}.npz"))
    p = S.Patch(48000, B, 2)
    S.build_p2(p, beta=float(z["beta"]), index=float(z["index"]))
    p.configure_voices(3)
    out = p.render_channels(len(z["audio"]))
    for v in range(3):
        assert_close(out[0, :, v], z["audio"])


# ---- every port of every module type, through the interpreter ---------------------------------------------
def test_cfg4_full_second_per_voice_params(S, oracle):
    """Config 4 for the whole second with the per-voice beta / index draw: the modulator's sine drives two pitches, so its
    error is integrated — default mode has to deliver the reference's own half-ulp sine to stay inside the contract."""
    V, T = 48, 48000
    beta, index = S.p2_voice_params(V)
    for B in (1, 1024):
        o = oracle.OraclePatch(48000, B, 2)
        ids = S.build_p2(o)
        ref, _ = o.render_batch(V, T, [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)], threads=8)
        for flags in (0, 1, S.RENDER_KEEP_DEFAULT):
            p = S.Patch(48000, B, 2)
            S.build_p2(p)
            p.configure_voices(V)
            p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta)
            p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
            assert assert_close(p.render_channels(T, flags), ref) < (2e-6 if flags == S.RENDER_KEEP_DEFAULT else 5e-7)


def _everything(g):
    lfo, osc, osc2, vcf, adsr, vca, mix, sub, add, vca_neg, out = [g.add_module(t) for t in (1, 1, 1, 2, 3, 4, 5, 6, 6, 4, 0)]
    for m, f, v in ((lfo, 0, -1.5), (osc, 0, 0.25), (osc, 1, 0), (osc2, 0, 1.0 / 12.0), (vcf, 0, 0.35), (vcf, 1, 0.8), (vcf, 2, 0.25),
                    (adsr, 0, 0.002), (adsr, 1, 0.004), (adsr, 2, 0.6), (adsr, 3, 0.003), (mix, 0, 0.5), (mix, 2, 1.5),
                    (sub, 1, 1), (sub, 0, 0.125), (add, 0, -0.75), (vca_neg, 0, 1)):
        g.set_field(m, f, v)
    for s, sp, d, dp in ((lfo, 0, osc, 0), (lfo, 1, osc, 1), (osc, 2, vcf, 0), (lfo, 2, vcf, 1), (lfo, 1, adsr, 0), (vcf, 1, vca, 0),
                         (adsr, 0, vca, 1), (vca, 0, mix, 0), (vcf, 2, mix, 2), (osc2, 1, mix, 3), (mix, 0, sub, 0), (osc2, 0, add, 1),
                         (add, 0, vca_neg, 0), (lfo, 0, vca_neg, 1), (sub, 0, out, 0), (vca_neg, 0, out, 1)):
        g.connect(s, sp, d, dp)
    return dict(lfo=lfo, osc=osc, osc2=osc2, vcf=vcf, adsr=adsr)


@pytest.mark.parametrize("flags", [0, 1])
def test_all_module_types_vs_oracle(S, oracle, flags):
    T, V = 3000, 130
    rng = np.random.default_rng(5)
    val = rng.uniform(-0.5, 0.5, V).astype(np.float32)
    cut = rng.uniform(0.1, 0.6, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 32, 2)
    ids = _everything(o)
    ref, ref_mix = o.render_batch(V, T, [(ids["osc"], S.OSC_VAL, val), (ids["vcf"], S.VCF_FREQ, cut)], mix=True, threads=8)
    p = S.Patch(48000, 32, 2)
    _everything(p)
    p.configure_voices(V)
    p.set_voice_field(ids["osc"], S.OSC_VAL, val)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    assert p.planes() == (2, [0, 1])
    fr, mix = p.render(T, flags=flags)
    assert_close(fr[0], ref[0])
    assert_close(fr[1], ref[1])
    scale = np.abs(ref.astype(np.float64)).sum(axis=2)
    assert (np.abs(mix - ref_mix) <= 2e-5 * np.maximum(scale, 1.0)).all()
    assert not np.array_equal(fr[0], fr[1])


def test_quirks_on_gpu(S, oracle):
    def both(build, T=64, B=16):
        o, p = oracle.OraclePatch(48000, B, 2), S.Patch(48000, B, 2)
        build(o)
        build(p)
        p.configure_voices(2)
        return o.render(T), p.render_channels(T)[:, :, 0], p

    def vca_open(g):  # VCA with one input unconnected => zeros (vca.rs:142-144)
        osc, vca, out = g.add_module(1), g.add_module(4), g.add_module(0)
        g.connect(osc, 0, vca, 0)
        g.connect(vca, 0, out, 0)
    ref, gpu, _ = both(vca_open)
    assert not gpu.any() and not ref.any()

    def gate_high(g):  # gate high at sample 0: not an edge, but None -> Attack is level triggered; a_sec = 0 => one-sample attack
        c, adsr, out = g.add_module(6), g.add_module(3), g.add_module(0)
        g.set_field(c, 0, 1.0)
        g.connect(c, 0, adsr, 0)
        g.connect(adsr, 0, out, 1)
    ref, gpu, _ = both(gate_high, T=30000)
    np.testing.assert_array_equal(bits(gpu), bits(ref))  # pure f32 path: bit-exact
    assert gpu[1][0] == 0.0 and gpu[1][1] == 1.0 and not gpu[0].any()

    def f0r0(g):  # freq = res = 0 from the start: coefficients never computed, f stays 0 (filter.rs:61)
        osc, vcf, out = g.add_module(1), g.add_module(2), g.add_module(0)
        g.set_field(vcf, 0, 0.0)
        g.set_field(vcf, 1, 0.0)
        g.connect(osc, 2, vcf, 0)
        g.connect(vcf, 0, out, 0)
        return vcf
    ref, gpu, p = both(f0r0)
    assert_close(gpu, ref)
    assert (p.get_voice_field(1, S.VCF_ST_F) == 0.0).all()

    def no_output(g):
        g.add_module(1)
    o, p = oracle.OraclePatch(48000, 16, 2), S.Patch(48000, 16, 2)
    no_output(o)
    no_output(p)
    p.configure_voices(2)
    fr, mix = p.render(32)
    assert fr.shape == (0, 32, 2) and not mix.any() and not o.render(32).any()


# ---- properties that do not need the oracle -----------------------------------------------------------------
@pytest.mark.parametrize("flags", FAST_MODES)
def test_render_continues_from_state(S, flags):
    """execute() carries state between calls: render(T) == render(a) ++ render(T - a), bit for bit."""
    V, T = 200, 5000
    det, cut = S.p1_voice_params(V)

    def make():
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
        p.configure_voices(V)
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        return p, ids
    p, ids = make()
    whole, mix_whole = p.render(T, flags=flags)
    end_pos = p.get_voice_field(ids["osc_a"], S.OSC_POS)
    q, _ = make()
    parts = [q.render(n, flags=flags) for n in (1, 999, 33, T - 1033)]
    np.testing.assert_array_equal(bits(np.concatenate([f[0] for f, _ in parts], axis=0)), bits(whole[0]))
    # the mix's summation order depends on where a sample falls in its 32-row tile, so chunked mixes agree to rounding only
    mix_parts = np.concatenate([m for _, m in parts], axis=1)
    scale = np.abs(whole[0].astype(np.float64)).sum(axis=1)
    assert (np.abs(mix_parts.astype(np.float64) - mix_whole) <= 1e-5 * np.maximum(scale, 1.0)).all()
    np.testing.assert_array_equal(q.get_voice_field(ids["osc_a"], S.OSC_POS), end_pos)
    assert ((end_pos >= 0) & (end_pos < 1)).all()


def test_feedback_ring_continues_across_renders(S, oracle):
    T, V = 2500, 66
    for B in (1, 16, 100):
        o = oracle.OraclePatch(48000, B, 2)
        S.build_p2(o)
        ref = o.render(T)[0]
        p = S.Patch(48000, B, 2)
        S.build_p2(p)
        p.configure_voices(V)
        parts = [p.render_channels(n)[0] for n in (7, 1000, T - 1007)]
        out = np.concatenate(parts, axis=0)
        assert_close(out[:, 0], ref)
        assert (out == out[:, :1]).all()


def test_mix_is_additive_over_voice_shards(S):
    """The multi-GPU contract: voices shard with no exchange; mix(all) == sum of the shards' mixes."""
    V, T = 1024, 3000
    det, cut = S.p1_voice_params(V)

    def shard(v0, v1):
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
        p.configure_voices(v1 - v0)
        d, c = S.p1_voice_params(v1 - v0, first_voice=v0)
        np.testing.assert_array_equal(d, det[v0:v1])
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, d)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, c)
        return p.render(T)
    fr, mix = shard(0, V)
    parts = [shard(a, b) for a, b in ((0, 256), (256, 512), (512, 1024))]
    np.testing.assert_array_equal(bits(np.concatenate([f[0] for f, _ in parts], axis=1)), bits(fr[0]))
    total = sum(m.astype(np.float64) for _, m in parts)
    scale = np.abs(fr[0].astype(np.float64)).sum(axis=1)
    assert (np.abs(mix[0] - total[0]) <= 1e-5 * np.maximum(scale, 1.0)).all()


def test_full_size_voices_short_render(S, oracle):
    """cfg3's voice count (262 144) for 512 samples: sampled voices vs the oracle, and the mix vs an f64 sum."""
    V, T = 262144, 512
    det, cut = S.p1_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=2.0)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    fr, mix = p.render(T)
    pick = np.unique(np.concatenate([np.arange(0, V, 4099), [0, 63, 64, V - 65, V - 64, V - 1]]))
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o, adsr="finite", lfo_val=2.0)
    ref, _ = o.render_batch(len(pick), T, [(ids["osc_a"], S.OSC_VAL, det[pick]), (ids["vcf"], S.VCF_FREQ, cut[pick])], threads=8)
    assert_close(fr[0][:, pick], ref[0])
    own = fr[0].astype(np.float64).sum(axis=1)
    scale = np.abs(fr[0].astype(np.float64)).sum(axis=1)
    assert (np.abs(mix[0] - own) <= 1e-5 * np.maximum(scale, 1.0)).all()


def test_two_million_voices_on_one_gpu(S, oracle):
    """Maximum sizes: config 5's 2 097 152 voices as ONE patch on one GPU (2048 samples: 17 GB of frames), first / last / sampled
    voices against the oracle, the mix against f64 sums of whole frame rows; and the limit itself (2^24 voices: a tile of 32 frame
    rows must fit a 31-bit buffer offset) is refused with an error, not rendered wrongly."""
    import ctypes as C
    V, T = 2097152, 2048
    det, cut = S.p1_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=-2.0)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_voice_chain_track" in p.info()
        pick = np.unique(np.concatenate([np.arange(0, V, 52429), [0, 63, 64, V - 65, V - 64, V - 1]]))
        ts = np.unique(np.concatenate([np.arange(0, T, 16), [31, 32, 1023, 1024, T - 1]]))
        row = np.empty(V, dtype=np.float32)
        got, own, scale = np.empty((len(ts), len(pick)), dtype=np.float32), np.empty(len(ts)), np.empty(len(ts))
        for k, t in enumerate(ts):
            assert S.lib.srack_device_to_host(row.ctypes.data_as(C.c_void_p), C.c_void_p(d_fr.value + int(t) * V * 4), V * 4, None) == 0
            got[k], own[k], scale[k] = row[pick], row.sum(dtype=np.float64), np.abs(row).sum(dtype=np.float64)
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o, adsr="finite", lfo_val=-2.0)
    ref, _ = o.render_batch(len(pick), T, [(ids["osc_a"], S.OSC_VAL, det[pick]), (ids["vcf"], S.VCF_FREQ, cut[pick])], threads=8)
    assert_close(got, ref[0][ts])
    assert np.abs(ref).max() > 0.3
    assert (np.abs(mix[0][ts].astype(np.float64) - own) <= 1e-5 * np.maximum(scale, 1.0)).all() and np.array_equal(mix[0], mix[1])
    q = S.Patch(48000, 1024, 2)
    S.build_p1(q)
    with pytest.raises(S.SrackError):
        q.configure_voices((1 << 24) + 1)


def test_cfg3_ticked_as_benchmarked(S):
    """`cfg3_ticked_1024_ms_per_step` on the bench line, checked: config 3 at full size driven the way the reference's audio callback drives
    `execute` — 47 calls of 1024 samples (a tick session: the control track of call c + 1 is computed under the voices of call c) —
    against the same second in ONE call, on the device: whole frame rows at a stride and both mixes, bit for bit."""
    import ctypes as C
    V, T, L = 262144, 48000, 1024
    det, cut = S.p1_voice_params(V)

    def patch():
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p1(p)
        p.configure_voices(V)
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        return p

    bufs = [C.c_void_p() for _ in range(4)]
    for b, n in zip(bufs, (T * V * 4, T * V * 4, 2 * T * 4, 2 * T * 4)):
        assert S.lib.srack_device_alloc(C.byref(b), n) == 0
    fr_one, fr_tick, mx_one, mx_tick = bufs
    try:
        one = patch()
        one.render_raw(T, fr_one, mx_one, 0, None)
        tick = patch()
        for t in range(0, T, L):   # a call's mix is [2][n] of its own: at 2 t floats into the buffer
            n = min(L, T - t)
            tick.render_raw(n, C.c_void_p(fr_tick.value + t * V * 4), C.c_void_p(mx_tick.value + 2 * t * 4), 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_voice_chain_track" in tick.info()
        a, b = np.empty(V, dtype=np.float32), np.empty(V, dtype=np.float32)
        loud = 0.0
        for t in list(range(0, T, 89)) + [1023, 1024, 1025, T - 897, T - 896, T - 1]:
            for dst, src in ((a, fr_one), (b, fr_tick)):
                assert S.lib.srack_device_to_host(dst.ctypes.data_as(C.c_void_p), C.c_void_p(src.value + t * V * 4), V * 4, None) == 0
            assert S.lib.srack_device_sync(None) == 0
            np.testing.assert_array_equal(bits(a), bits(b), err_msg=f"row {t}")
            loud = max(loud, float(np.abs(a).max()))
        assert loud > 0.05
        m1, m2 = np.empty((2, T), dtype=np.float32), np.empty(2 * T, dtype=np.float32)
        assert S.lib.srack_device_to_host(m1.ctypes.data_as(C.c_void_p), mx_one, m1.nbytes, None) == 0
        assert S.lib.srack_device_to_host(m2.ctypes.data_as(C.c_void_p), mx_tick, m2.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
        for t in range(0, T, L):
            n = min(L, T - t)
            blk = m2[2 * t:2 * t + 2 * n].reshape(2, n)
            np.testing.assert_array_equal(bits(blk), bits(m1[:, t:t + n]), err_msg=f"mix of the call at {t}")
    finally:
        for b in bufs:
            S.lib.srack_device_free(b)


def test_cfg3_exactly_as_benchmarked(S, oracle):
    """BASELINE config 3 at full size, the very workload bench.py times: 262 144 voices x 48 000 samples (50 GB of frames,
    kept on the device), default mode.  67 sampled voices against the oracle for the whole second, and the stereo mix
    against an f64 sum of all the frames (read back 256 rows at a time)."""
    import ctypes as C
    V, T = 262144, 48000
    det, cut = S.p1_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_voice_chain_track" in p.info()
        pick = np.unique(np.concatenate([np.arange(0, V, 4099), [0, 63, 64, V - 65, V - 64, V - 1]]))
        got, own, scale = read_plane(S, d_fr.value, T, V, pick)
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o)
    ref, _ = o.render_batch(len(pick), T, [(ids["osc_a"], S.OSC_VAL, det[pick]), (ids["vcf"], S.VCF_FREQ, cut[pick])], threads=8)
    assert assert_close(got, ref[0]) < 2e-6
    assert np.abs(got).max() > 0.1
    assert (np.abs(mix[0].astype(np.float64) - own) <= 1e-5 * np.maximum(scale, 1.0)).all() and np.array_equal(mix[0], mix[1])


def test_cfg3_poly_exactly_as_benchmarked(S, oracle):
    """The fully per-voice variant of config 3 at full size, the workload `bench.py --workload cfg3_poly` (and the default line's
    `cfg3_poly_*`) times: 262 144 voices x 48 000 samples of P1 with per-voice detune, cutoff, gate-LFO rate and envelope times (nothing is
    voice-invariant: every voice has its own notes), default mode.  Sampled voices against the oracle for the whole second — a gate edge
    one sample off would be an error of an envelope increment, 1e-2 — and the mix against an f64 sum of all the frames."""
    import ctypes as C
    V, T = 262144, 48000
    pv = S.p1_poly_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    p.configure_voices(V)
    for m, f, v in S.p1_poly_overrides(ids, pv):
        p.set_voice_field(m, f, v)
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_specialized" in p.info() and "ctl[" not in p.info(), p.info()   # nothing hoisted: no control program
        pick = np.unique(np.concatenate([np.arange(0, V, 2731), [0, 63, 64, V - 65, V - 64, V - 1]]))
        got, own, scale = read_plane(S, d_fr.value, T, V, pick)
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o)
    ref, _ = o.render_batch(len(pick), T, [(m, f, v[pick]) for m, f, v in S.p1_poly_overrides(ids, pv)], threads=8)
    assert assert_close(got, ref[0]) < 2e-6
    assert np.abs(got).max() > 0.1
    assert (np.abs(mix[0].astype(np.float64) - own) <= 1e-5 * np.maximum(scale, 1.0)).all() and np.array_equal(mix[0], mix[1])


@pytest.mark.parametrize("flags", [0, 1, 2, 3, 4, 16, 32 | 2])
@pytest.mark.parametrize("V,T", [(192, 48000), (4096 + 37, 6000)])
def test_cfg3_poly_modes(S, oracle, flags, V, T):
    """cfg3_poly's patch through every render mode (default / exact, the fused kernel, the specialised kernel, the interpreter), a ragged
    last wave, a whole second at the small size (several notes per voice, every envelope segment): exact modes bit for bit, default modes
    within the contract — and the envelope alone (a second patch: ADSR straight to the output) bit for bit in EVERY mode, gate edges included."""
    pv = S.p1_poly_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    p.configure_voices(V)
    ov = S.p1_poly_overrides(ids, pv)
    for m, f, v in ov:
        p.set_voice_field(m, f, v)
    fr, mix = p.render(T, flags=flags)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p1(o)
    ref, _ = o.render_batch(V, T, ov, threads=8)
    if flags & 1:
        np.testing.assert_array_equal(bits(fr[0]), bits(ref[0]))
    else:
        assert assert_close(fr[0], ref[0]) < 2e-6
    assert np.abs(fr[0]).max() > 0.1
    # the envelope by itself: LFO square -> ADSR -> OUT
    q = S.Patch(48000, 1024, 2)
    lfo, adsr, out = q.add_module(S.MOD_OSCILLATOR), q.add_module(S.MOD_ADSR), q.add_module(S.MOD_OUTPUT)
    q.connect(lfo, S.OSC_OUT_SQUARE, adsr, 0)
    q.connect(adsr, 0, out, 0)
    q.configure_voices(V)
    ov2 = [(lfo, S.OSC_VAL, pv["lfo_val"]), (adsr, S.ADSR_A_SEC, pv["a_sec"]), (adsr, S.ADSR_D_SEC, pv["d_sec"]), (adsr, S.ADSR_S_VAL, pv["s_val"]),
           (adsr, S.ADSR_R_SEC, pv["r_sec"])]
    for m, f, v in ov2:
        q.set_voice_field(m, f, v)
    env, _ = q.render(T, flags=flags)
    o2 = oracle.OraclePatch(48000, 1024, 2)
    for t in (S.MOD_OSCILLATOR, S.MOD_ADSR, S.MOD_OUTPUT):
        o2.add_module(t)
    o2.connect(lfo, S.OSC_OUT_SQUARE, adsr, 0)
    o2.connect(adsr, 0, out, 0)
    ref2, _ = o2.render_batch(V, T, ov2, threads=8)
    np.testing.assert_array_equal(bits(env[0]), bits(ref2[0]))
    assert env[0].max() > 0.9


# (since round 3 the kernels bench.py times for this patch: buffer_size 1 — the kernel specialised at run time, whose generator derives
# the bounded pitch CVs render_fm_pair proves by hand; buffer_size 1024 — the time-parallel pair with its ring in LDS)
@pytest.mark.parametrize("B,flags,kernel", [(1, 64, "render_specialized"), (1024, 64, "render_fm_pair_block"), (1, 0, "render_fm_pair_x"), (1024, 0, "render_fm_pair_block_x")])
def test_cfg4_exactly_as_benchmarked(S, oracle, B, flags, kernel):
    """BASELINE config 4 at full size, the workload `bench.py --workload cfg4` (and cfg4_b1024) times: 65 536 voices x 48 000
    samples of the 2-operator FM patch with its feedback edge, per-voice feedback / index (12.6 GB of frames, kept on the device) —
    flags 0: the modulator exact as a whole, as the flattener renders a loop through a pitch (`cfg4_*` on the bench line); KEEP_DEFAULT: the fast kernels
    (`cfg4_fast_*`), within the contract for a render of seconds (tests/test_gpu_horizon.py has the minute).  37 sampled voices against the oracle for the whole second — the feedback makes every error an integrated
    one — and the mix against an f64 sum of all the frames."""
    import ctypes as C
    V, T = 65536, 48000
    beta, index = S.p2_voice_params(V)
    p = S.Patch(48000, B, 2)
    ids = S.build_p2(p)
    p.configure_voices(V)
    p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta)
    p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, flags, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=" + kernel in p.info() and ("approx[kept default" if flags else "; exact osc 0]") in p.info(), p.info()
        pick = np.unique(np.concatenate([np.arange(0, V, 2113), [0, 63, 64, V - 65, V - 64, V - 1]]))
        got, own, scale = read_plane(S, d_fr.value, T, V, pick)
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, B, 2)
    S.build_p2(o)
    ref, _ = o.render_batch(len(pick), T, [(ids["mul_fb"], S.MATH_CONSTANT, beta[pick]), (ids["mul_idx"], S.MATH_CONSTANT, index[pick])], threads=8)
    assert assert_close(got, ref[0]) < (3e-6 if flags else 5e-7)   # (the carrier's sine is evaluated in f32 after the exact fold: 2e-7, not integrated by anything; with the modulator exact that is all there is)
    assert np.abs(got).max() > 0.9
    assert (np.abs(mix[0].astype(np.float64) - own) <= 1e-5 * np.maximum(scale, 1.0)).all() and np.array_equal(mix[0], mix[1])


# ---- edge cases: channel counts, ragged lengths, chunk boundaries, empty renders ----------------------------------
def test_channel_layouts(S, oracle):
    """1 and 4 output channels; distinct wires, a shared wire and an unconnected channel."""
    def build(g, channels):
        osc, osc2, vcf, out = g.add_module(1), g.add_module(1), g.add_module(2), g.add_module(0)
        g.set_field(osc2, 0, 0.5)
        g.connect(osc, 2, vcf, 0)
        g.connect(vcf, 0, out, 0)
        if channels == 4:
            g.connect(osc2, 0, out, 1)   # a second plane
            g.connect(vcf, 0, out, 3)    # shares channel 0's plane; channel 2 stays unconnected
    for channels in (1, 4):
        o, p = oracle.OraclePatch(48000, 256, channels), S.Patch(48000, 256, channels)
        build(o, channels)
        build(p, channels)
        V, T = 70, 900
        val = np.linspace(-0.3, 0.3, V).astype(np.float32)
        ref, ref_mix = o.render_batch(V, T, [(0, S.OSC_VAL, val)], mix=True, threads=4)
        p.configure_voices(V)
        p.set_voice_field(0, S.OSC_VAL, val)
        n_planes, cp = p.planes()
        assert (n_planes, cp) == ((1, [0]) if channels == 1 else (2, [0, 1, -1, 0]))
        got = p.render_channels(T)
        assert_close(got, ref)
        _, mix = p.render(T, frames=False, mix=True)  # continues from the state: compare against the oracle's next T samples
        ref2, ref_mix2 = o.render_batch(V, 2 * T, [(0, S.OSC_VAL, val)], mix=True, threads=4)
        scale = np.abs(ref2[:, T:, :].astype(np.float64)).sum(axis=2)
        assert (np.abs(mix - ref_mix2[:, T:]) <= 2e-5 * np.maximum(scale, 1.0)).all()
        if channels == 4:
            assert not got[2].any() and not mix[2].any()


@pytest.mark.parametrize("T", [1, 31, 33, 1023, 1025, 3073, 7169, 65536 + 4097])
def test_ragged_lengths_across_tiles_and_chunks(S, oracle, T):
    """Lengths around the 32-sample tile, the 1024/2048/4096-sample chunk borders of the pipelined render and the
    65536-sample segments a long render is cut into (bounded scratch)."""
    V = 96 if T < 65536 else 40
    det, cut = S.p1_voice_params(V, first_voice=99)
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = S.build_p1(o, adsr="finite", lfo_val=1.0)
    ref, _ = o.render_batch(V, T, [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)], threads=8)
    for flags in (0, 2):
        p = S.Patch(48000, 1024, 2)
        S.build_p1(p, adsr="finite", lfo_val=1.0)
        p.configure_voices(V)
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        fr, mix = p.render(T, flags=flags)
        assert_close(fr[0], ref[0])
        if T > 65536:  # the mix of a segmented render lands at the right offsets of [channels][T]
            want = ref[0].astype(np.float64).sum(axis=1)
            scale = np.abs(ref[0].astype(np.float64)).sum(axis=1)
            assert (np.abs(mix[0] - want) <= 2e-5 * np.maximum(scale, 1.0)).all() and np.array_equal(mix[0], mix[1])


def test_empty_render_and_output_selection(S):
    p = S.Patch(48000, 64, 2)
    ids = S.build_p1(p, lfo_val=2.0)
    p.configure_voices(130)
    p.render_raw(0)  # zero samples: nothing to do, no device buffers needed
    fr, mix = p.render(500)
    fr2, _ = S_render_again(S, frames=True, mix=False)
    _, mix2 = S_render_again(S, frames=False, mix=True)
    np.testing.assert_array_equal(fr2, fr)     # frames do not depend on whether the mix is requested
    np.testing.assert_array_equal(mix2, mix)   # and vice versa


def S_render_again(S, frames, mix):
    p = S.Patch(48000, 64, 2)
    S.build_p1(p, lfo_val=2.0)
    p.configure_voices(130)
    return p.render(500, frames=frames, mix=mix)


def test_render_without_outputs_still_advances_state(S):
    """d_frames = d_mix = NULL is a skip-ahead: state moves exactly as if the frames had been written."""
    def make():
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p1(p, adsr="finite", lfo_val=1.0)
        p.configure_voices(100)
        det, cut = S.p1_voice_params(100)
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        return p
    a, b = make(), make()
    a.render(3000)
    b.render_raw(3000)  # no outputs
    fa, _ = a.render(500)
    fb, _ = b.render(500)
    np.testing.assert_array_equal(bits(fa), bits(fb))


# ---- scope table (f) rank 1: sequencer-driven patch ---------------------------------------------------------------
@pytest.mark.parametrize("flags", [pytest.param(0, id="hoist"), pytest.param(4, id="nohoist"), pytest.param(1, id="exact"), pytest.param(2, id="hoist-interp"),
                                   pytest.param(8, id="hoist-one-control-unit"), pytest.param(10, id="hoist-interp-one-control-unit"),
                                   pytest.param(32, id="special"), pytest.param(34, id="special-nofusion"), pytest.param(35, id="special-exact"),
                                   pytest.param(42, id="special-one-control-unit"), pytest.param(38, id="special-nohoist")])
def test_p3_sequencers_vs_oracle(S, oracle, flags):
    V, T = 150, 12000
    transpose = np.linspace(-2.0, 0.5, V).astype(np.float32)
    cut = np.linspace(0.05, 0.4, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = S.build_p3(o)
    ref, ref_mix = o.render_batch(V, T, [(ids["transpose"], S.MATH_CONSTANT, transpose), (ids["vcf"], S.VCF_FREQ, cut)], mix=True, threads=8)
    p = S.Patch(48000, 1024, 2)
    S.build_p3(p)
    p.configure_voices(V)
    p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, transpose)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    assert p.planes() == (2, [0, 1])
    fr, mix = p.render(T, flags=flags)
    assert ("ctl[" in p.info()) == (not flags & 4)
    if not flags & 4:  # five pipelined control units, or one when staging is off
        assert p.info().count("ctl[") == (1 if flags & 8 else 5)
        assert ("fused=5" in p.info()) == (not flags & 3)
    if flags & 32:  # the specialised kernel, with the control units as the first blocks of its launches
        assert "kernel=render_specialized" in p.info()
    assert_close(fr[0], ref[0])
    np.testing.assert_array_equal(fr[1], ref[1])  # a raw pattern gate: exactly 0.0 / 1.0 / the clock's square
    scale = np.abs(ref.astype(np.float64)).sum(axis=2)
    assert (np.abs(mix - ref_mix) <= 2e-5 * np.maximum(scale, 1.0)).all()
    assert np.abs(fr[0]).max() > 0.1
    # sequencer state is readable per voice (one shared state when it lives in the control program)
    steps = p.get_voice_field(ids["grid"], S.GRIDSEQ_CURRENT_STEP)
    assert (steps == steps[0]).all() and 0 <= steps[0] < 8


def test_p3_exactly_as_benchmarked(S, oracle):
    """`bench.py --workload p3` at full size (the sequencers as control tracks): 262 144 voices x 48 000 samples, two planes
    (100 GB of frames, kept on the device), default mode — the specialised kernel with five control units on its launches.  Sampled voices
    against the oracle over the whole second (plane 1, the raw pattern gate, exactly); plane 0's mix against the f64 sum of all its frames,
    plane 1's against voices x gate."""
    import ctypes as C
    V, T = 262144, 48000
    u0, u1 = S.voice_uniform(V, 0), S.voice_uniform(V, 1)
    transpose = (u0 * np.float32(2.5) - np.float32(2.0)).astype(np.float32)   # bench.py's draw
    cut = (np.float32(0.05) + u1 * np.float32(0.35)).astype(np.float32)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p3(p)
    p.configure_voices(V)
    p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, transpose)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    assert p.planes() == (2, [0, 1])
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), 2 * T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_specialized" in p.info() and p.info().count("ctl[") == 5
        pick = np.unique(np.concatenate([np.arange(0, V, 8209), [0, 63, 64, V - 65, V - 64, V - 1]]))
        got = np.empty((2, T, len(pick)), dtype=np.float32)
        # plane 0 in full (50 GB over PCIe), plane 1 (every voice the same gate) at the sampled voices
        got[0], own, scale = read_plane(S, d_fr.value, T, V, pick)
        row = np.empty(V, dtype=np.float32)
        for t in range(0, T, 997):   # plane 1: whole rows at a stride (every voice must carry the same gate)
            assert S.lib.srack_device_to_host(row.ctypes.data_as(C.c_void_p), C.c_void_p(d_fr.value + ((T + t) * V) * 4), V * 4, None) == 0
            assert S.lib.srack_device_sync(None) == 0
            assert (row == row[0]).all(), t
            got[1, t] = row[pick]
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p3(o)
    ref, _ = o.render_batch(len(pick), T, [(ids["transpose"], S.MATH_CONSTANT, transpose[pick]), (ids["vcf"], S.VCF_FREQ, cut[pick])], threads=8)
    assert assert_close(got[0], ref[0]) < 5e-6
    at = np.arange(0, T, 997)
    np.testing.assert_array_equal(got[1][at], ref[1][at])   # the raw gate: exactly 0.0 / 1.0 / the clock's square
    assert np.abs(got[0]).max() > 0.1 and ref[1].max() > 0.5
    assert (np.abs(mix[0].astype(np.float64) - own) <= 1e-5 * np.maximum(scale, 1.0)).all()
    assert (np.abs(mix[1].astype(np.float64) - V * ref[1][:, 0].astype(np.float64)) <= 1e-5 * np.maximum(V * np.abs(ref[1][:, 0]), 1.0)).all()


@pytest.mark.parametrize("flags", [pytest.param(0, id="fused"), pytest.param(2, id="interp")])
def test_p3_extreme_transpose_takes_the_literal_oscillator(S, oracle, flags):
    """Notes whose phase increment reaches a quarter cycle (>= 12 kHz at 48 kHz) leave the carried-phase form: lanes on both
    sides of that limit share waves here, and the filter has no CV (the second shape of the fused chain)."""
    V, T = 100, 6000
    transpose = np.linspace(3.0, 5.6, V).astype(np.float32)
    def build(g):
        ids = S.build_p3(g)
        g.disconnect(ids["vcf"], 1)
        return ids
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = build(o)
    ref, _ = o.render_batch(V, T, [(ids["transpose"], S.MATH_CONSTANT, transpose)], threads=8)
    p = S.Patch(48000, 1024, 2)
    build(p)
    p.configure_voices(V)
    p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, transpose)
    fr, _ = p.render(T, flags=flags)
    assert ("fused=5" in p.info()) == (flags == 0)
    assert_close(fr[0], ref[0])
    np.testing.assert_array_equal(fr[1], ref[1])


@pytest.mark.parametrize("flags", [pytest.param(0, id="fused-pipelined"), pytest.param(2, id="interp-pipelined"), pytest.param(8, id="fused-one-unit"),
                                   pytest.param(32, id="special-pipelined"), pytest.param(40, id="special-one-unit")])
def test_p3_render_continues_across_calls(S, flags):
    """The control pipeline fills and drains inside every call: render(T) == render(a) ++ render(b) ++ ..., bit for bit,
    for call lengths around the chunk sizes (1024 x depth, then doubling) and tiles."""
    V, T = 70, 21000
    transpose = np.linspace(-1.5, 0.5, V).astype(np.float32)

    def make():
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p3(p, clock_val=-2.0)
        p.configure_voices(V)
        p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, transpose)
        return p, ids
    p, ids = make()
    whole, _ = p.render(T, flags=flags)
    q, _ = make()
    parts = [q.render(n, flags=flags)[0] for n in (1, 1023, 1025, 3000, 4097, 63, T - 9209)]
    got = np.concatenate(parts, axis=1)
    np.testing.assert_array_equal(bits(got), bits(whole))
    for f in (S.GRIDSEQ_CURRENT_STEP, S.GRIDSEQ_LAST):
        np.testing.assert_array_equal(q.get_voice_field(ids["grid"], f), p.get_voice_field(ids["grid"], f))
    np.testing.assert_array_equal(q.get_voice_field(ids["adsr_flt"], S.ADSR_PHASE), p.get_voice_field(ids["adsr_flt"], S.ADSR_PHASE))
    assert np.abs(whole[0]).max() > 0.1 and len(np.unique(whole[1])) >= 2


def test_p3_per_voice_clocks(S, oracle):
    """Per-voice clock rates: the sequencers step at different times in different lanes."""
    V, T = 96, 9000
    rate = np.linspace(-5.0, -3.0, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 64, 2)
    ids = S.build_p3(o)
    ref, _ = o.render_batch(V, T, [(ids["clock"], S.OSC_VAL, rate)], threads=8)
    p = S.Patch(48000, 64, 2)
    S.build_p3(p)
    p.configure_voices(V)
    p.set_voice_field(ids["clock"], S.OSC_VAL, rate)
    fr, _ = p.render(T)
    assert "ctl[" not in p.info()
    assert_close(fr[0], ref[0])
    assert_close(fr[1], ref[1])
    steps = p.get_voice_field(ids["grid"], S.GRIDSEQ_CURRENT_STEP)
    assert len(np.unique(steps)) > 1


def test_p3_golden(S):
    z = np.load(os.path.join(GOLD, "p3_sequencers.npz"))
    gold = z["audio"]  # [T][2]
    p = S.Patch(48000, int(z["buffer_size"]), 2)
    S.build_p3(p)
    p.configure_voices(2)
    out = p.render_channels(gold.shape[0])
    for v in range(2):
        assert_close(out[0, :, v], gold[:, 0])
        np.testing.assert_array_equal(out[1, :, v], gold[:, 1])


# ---- scope table (f) rank 4: sample player + sign-preserving waveshaper -----------------------------------------------
def _p4_pair(S, oracle, V, T, B=1024, **kw):
    depth, expo = S.p4_voice_params(V)
    o = oracle.OraclePatch(48000, B, 2)
    ids = S.build_p4(o, **kw)
    ref, ref_mix = o.render_batch(V, T, [(ids["depth"], S.MATH_CONSTANT, depth), (ids["shaper"], S.NONLIN_CONSTANT, expo)], mix=True, threads=8)
    p = S.Patch(48000, B, 2)
    S.build_p4(p, **kw)
    p.configure_voices(V)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth)
    p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
    return p, ids, ref, ref_mix


@pytest.mark.parametrize("flags", [pytest.param(1, id="exact"), pytest.param(5, id="exact-nohoist"), pytest.param(35, id="special-exact"),
                                   pytest.param(39, id="special-exact-nohoist")])
def test_p4_sample_nonlinear_vs_oracle(S, oracle, flags):
    """With the exact oscillator the vibrato CV has the oracle's bits, so the read position — an index — must too."""
    V, T = 130, 9000
    p, ids, ref, ref_mix = _p4_pair(S, oracle, V, T)
    assert p.planes() == (2, [0, 1])
    fr, mix = p.render(T, flags=flags)
    np.testing.assert_array_equal(bits(fr[1]), bits(ref[1]))   # the raw sample player: wave values, bit for bit
    np.testing.assert_array_equal(bits(fr[0]), bits(ref[0]))   # the waveshaper: the host libm's powf, operation for operation (round 6: powf_libm_plain;
                                                               # until then a correctly rounded power of the kernels' own, an f32 ulp off now and then)
    scale = np.abs(ref.astype(np.float64)).sum(axis=2)
    assert (np.abs(mix - ref_mix) <= 2e-5 * np.maximum(scale, 1.0)).all()
    assert np.abs(fr[1]).max() > 0.3 and len(np.unique(fr[1][:, 0])) > 100
    pos = p.get_voice_field(ids["smp"], S.SAMPLE_POS)
    assert len(np.unique(pos)) > 1 and (pos >= 0).all()


def test_p4_default_mode_index_slips_are_rare(S, oracle):
    """Default mode evaluates the LFO's sine in f32 (|err| ~1e-7): the vibrato CV, hence the read position, carries that
    error, and a position within ~1e-4 of an integer may truncate to the neighbouring wave sample.  Everything that is
    not such a slip is still bit-exact; slips stay below 0.1 % of the samples."""
    V, T = 130, 9000
    p, ids, ref, _ = _p4_pair(S, oracle, V, T)
    fr, _ = p.render(T)
    same = bits(fr[1]) == bits(ref[1])
    assert same.mean() > 0.999
    err = np.abs(fr[0].astype(np.float64) - ref[0]) / np.maximum(np.abs(ref[0]), 1.0)
    assert (err[same] <= TOL).all()


def test_p4_exactly_as_benchmarked(S, oracle):
    """`bench.py --workload p4` at full size (SURVEY 8(f4)'s modules): 131 072 voices x 48 000 samples, two planes (50 GB of frames, kept on
    the device), default mode — the specialised kernel with the planes' shared mix tile and NonLinear's power through the f32
    transcendental unit.  34 sampled voices against the oracle for the whole second: the raw sample plane bit for bit but for the rare
    index slips of the default mode's f32 LFO sine (< 0.1 % of the samples), the shaper within the contract wherever the player read
    the same sample; both mixes against f64 sums of all the frames."""
    import ctypes as C
    V, T = 131072, 48000
    depth, expo = S.p4_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p4(p)
    p.configure_voices(V)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth)
    p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
    assert p.planes() == (2, [0, 1])
    d_fr, d_mx = C.c_void_p(), C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), 2 * T * V * 4) == 0 and S.lib.srack_device_alloc(C.byref(d_mx), 2 * T * 4) == 0
    try:
        p.render_raw(T, d_fr, d_mx, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        assert "kernel=render_specialized" in p.info()
        src = p.kernel_source(0)
        assert "emit_rows_flush<16>" in src and re.search(r"nonlin_step\(0x[0-9a-f]*2[0-9a-f]{2}u", src)   # the shared tile; NONLIN_LOOSE (0x200)
        pick = np.unique(np.concatenate([np.arange(0, V, 4099), [0, 63, 64, V - 65, V - 64, V - 1]]))
        got = np.empty((2, T, len(pick)), dtype=np.float32)
        own, scale = np.empty((2, T)), np.empty((2, T))
        for plane in range(2):
            got[plane], own[plane], scale[plane] = read_plane(S, d_fr.value + plane * T * V * 4, T, V, pick)
        mix = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(mix.ctypes.data_as(C.c_void_p), d_mx, mix.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
    finally:
        S.lib.srack_device_free(d_fr)
        S.lib.srack_device_free(d_mx)
    o = oracle.OraclePatch(48000, 1024, 2)
    S.build_p4(o)
    ref, _ = o.render_batch(len(pick), T, [(ids["depth"], S.MATH_CONSTANT, depth[pick]), (ids["shaper"], S.NONLIN_CONSTANT, expo[pick])], threads=8)
    same = bits(got[1]) == bits(ref[1])
    assert same.mean() > 0.999, same.mean()
    err = np.abs(got[0].astype(np.float64) - ref[0]) / np.maximum(np.abs(ref[0]), 1.0)
    assert (err[same] <= TOL).all(), err[same].max()
    assert np.abs(got[1]).max() > 0.3
    for plane in range(2):
        assert (np.abs(mix[plane].astype(np.float64) - own[plane]) <= 1e-5 * np.maximum(scale[plane], 1.0)).all(), plane


@pytest.mark.parametrize("B", [1, 64])
def test_p4_constant_pitch_is_exact_in_default_mode(S, oracle, B):
    """No oscillator in the pitch path: CV from per-voice constants only => indices exact in every mode."""
    V, T = 70, 5000
    cv = np.linspace(-2.0, 2.0, V).astype(np.float32)
    def build(g):
        clk, off, smp, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_MATH), g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_OUTPUT)
        g.set_field(clk, S.OSC_VAL, -2.0)
        g.set_wave(smp, S.p4_wave(700), 32000.0)
        g.connect(clk, S.OSC_OUT_SQUARE, smp, 0)
        g.connect(off, 0, smp, 1)             # Add(None, constant) = 0.0 + constant
        g.connect(smp, 0, out, 0)
        return off
    o = oracle.OraclePatch(48000, B, 2)
    off = build(o)
    ref, _ = o.render_batch(V, T, [(off, S.MATH_CONSTANT, cv)], threads=8)
    for flags in (0, 4):
        p = S.Patch(48000, B, 2)
        build(p)
        p.configure_voices(V)
        p.set_voice_field(off, S.MATH_CONSTANT, cv)
        fr, _ = p.render(T, flags=flags)
        np.testing.assert_array_equal(bits(fr[0]), bits(ref[0]))
        assert p.planes() == (1, [0, -1])          # channel 1 unconnected: no plane, silence


def test_sample_edge_cases_on_gpu(S, oracle):
    """Empty WaveBox, unconnected gate / CV, huge and NaN-producing pitch CVs, a one-sample wave."""
    def run(build, T=600, V=3):
        o = oracle.OraclePatch(48000, 32, 2)
        build(o)
        ref, _ = o.render_batch(V, T, [], threads=1)
        p = S.Patch(48000, 32, 2)
        build(p)
        p.configure_voices(V)
        fr = p.render_channels(T, 1)
        np.testing.assert_array_equal(bits(fr), bits(ref))
        return fr

    def empty(g):
        clk, smp, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_OUTPUT)
        g.set_field(clk, S.OSC_VAL, 2.0)
        g.connect(clk, 1, smp, 0)
        g.connect(smp, 0, out, 0)
    assert not run(empty).any()

    def no_gate(g):
        smp, out = g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_OUTPUT)
        g.set_wave(smp, S.p4_wave(50), 48000.0)
        g.set_field(smp, S.SAMPLE_WAVE_NEW, 0)     # keep the state set below
        g.set_field(smp, S.SAMPLE_PLAYING, 1)      # a saved patch caught mid-playback
        g.set_field(smp, S.SAMPLE_POS, 7.0)
        g.connect(smp, 0, out, 0)
    fr = run(no_gate)
    np.testing.assert_array_equal(fr[0, :43, 0], S.p4_wave(50)[7:])

    for cv in (200.0, -200.0, 127.5, -140.0):
        def wild(g, cv=cv):
            clk, k, smp, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_MATH), g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_OUTPUT)
            g.set_field(clk, S.OSC_VAL, 1.0)
            g.set_field(k, S.MATH_CONSTANT, cv)
            g.set_wave(smp, S.p4_wave(90), 44100.0)
            g.connect(clk, 1, smp, 0)
            g.connect(k, 0, smp, 1)
            g.connect(smp, 0, out, 0)
            g.connect(clk, 1, out, 1)
        run(wild)

    def one(g):
        clk, smp, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_SAMPLE), g.add_module(S.MOD_OUTPUT)
        g.set_wave(smp, np.array([0.625], dtype=np.float32), 8000.0)
        g.connect(clk, 1, smp, 0)
        g.connect(smp, 0, out, 0)
    assert (run(one) [0] == 0.625).all()


@pytest.mark.parametrize("flags", [pytest.param(1, id="exact"), pytest.param(0, id="default"), pytest.param(32, id="default-special")])
def test_nonlinear_edge_cases_on_gpu(S, oracle, flags):
    """Negative / zero / huge bases and exponents, both inputs wired, In1 unconnected (math.rs:299-304).  Exact mode: only powf itself is
    compared (purely relative).  Default modes: the power goes through the f32 transcendental unit here (nothing integrates these outputs:
    NONLIN_LOOSE) while |b log2 x| < 32, through the f64 table form beyond and ocml's powf for the special cases — held to the contract's
    bar, with NaNs where the oracle has them and infinities where it has them (but for results within a hair of the overflow threshold)."""
    V, T = 64, 400
    expo = np.concatenate([np.linspace(0.5, 2.0, 40), [0.0, 1.0, 3.0, -1.0, -0.5, 40.0, -40.0, 0.25], np.linspace(2.0, 9.0, 16)]).astype(np.float32)
    def build(g):
        osc, gain, nl, nl2, nl3, out = (g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_MATH), g.add_module(S.MOD_NONLINEAR),
                                        g.add_module(S.MOD_NONLINEAR), g.add_module(S.MOD_NONLINEAR), g.add_module(S.MOD_OUTPUT))
        g.set_field(osc, S.OSC_VAL, 1.5)
        g.set_field(gain, S.MATH_OPERATION, S.MATH_MULTIPLY)
        g.set_field(gain, S.MATH_CONSTANT, 3.0)
        g.connect(osc, S.OSC_OUT_SAW, gain, 0)
        g.connect(gain, 0, nl, 0)          # base in [-3, 3], exponent = per-voice constant
        g.connect(osc, S.OSC_OUT_SAW, nl2, 0)
        g.connect(gain, 0, nl2, 1)         # both wired: base saw, exponent 3*saw
        g.connect(gain, 0, nl3, 1)         # In1 unconnected: operation(0.0, b)
        mix = g.add_module(S.MOD_MONO_MIXER)
        g.connect(nl2, 0, mix, 0)
        g.connect(nl3, 0, mix, 1)
        g.connect(nl, 0, out, 0)
        g.connect(nl2, 0, out, 1)
        return nl, nl3
    o = oracle.OraclePatch(48000, 64, 2)
    nl, _ = build(o)
    ref, _ = o.render_batch(V, T, [(nl, S.NONLIN_CONSTANT, expo)], threads=4)
    p = S.Patch(48000, 64, 2)
    build(p)
    p.configure_voices(V)
    p.set_voice_field(nl, S.NONLIN_CONSTANT, expo)
    fr, _ = p.render(T, flags=flags)       # (flags 1: exact oscillator — identical bases, so only powf itself is compared)
    fin = np.isfinite(ref)
    np.testing.assert_array_equal(np.isnan(fr), np.isnan(ref))
    if flags & 1:
        np.testing.assert_array_equal(fr[~fin & ~np.isnan(ref)], ref[~fin & ~np.isnan(ref)])   # +-inf where libm overflows
        # powf spans 1e-38 .. 1e38 here, lands on subnormals and on zero: the libm's own bits everywhere (round 6)
        np.testing.assert_array_equal(bits(fr[fin]), bits(ref[fin]))
        return
    if flags & 32:
        assert "kernel=render_specialized" in p.info() and "powf_pos_loose" not in p.info()
        # (which form of the power runs is approx.cpp's call — tests/test_approx.py; with bases up to 3 and exponents from -40 to 40 the values
        # have no bound, and the patch is rendered in the exact flavour: "approx[exact: unbounded values ...")
        assert "approx[exact: unbounded values" in p.info(), p.info()
    # The default mode's saw is the oracle's within 1e-7, and a power is as sensitive to its base as it likes: x^0.5 at a zero crossing
    # turns 1e-7 into 3e-4, x^40 multiplies a relative 1e-7 by 40.  The bar is the contract's where the power is well conditioned (the
    # base — the same for every voice: tapped from the oracle — at least 0.02 from zero); everywhere: finite where the oracle is but at
    # the overflow threshold, the sign right, the typical relative error that of an f32.
    o1 = oracle.OraclePatch(48000, 64, 2)
    build(o1)
    _, base = o1.render(T, tap=(1, 0))      # the Multiply's output: 3 x saw (nl's base; nl2's base is a third of it, its exponent this)
    calm = (np.abs(base) > 0.06)[None, :, None] & np.ones(fr.shape, dtype=bool)
    both = fin & np.isfinite(fr)
    g64, r64 = fr.astype(np.float64), ref.astype(np.float64)
    sel = both & calm
    assert sel.mean() > 0.8
    # (channel 0: the exponent b is the voice's; a relative 2e-7 of the base is a relative 2e-7 |b| of the power)
    bar = 1e-5 * np.maximum(np.abs(r64), 1.0) * np.stack([np.broadcast_to(np.maximum(np.abs(expo.astype(np.float64)), 1.0), (T, V)), np.ones((T, V))])
    with np.errstate(invalid="ignore"):
        d64 = np.abs(g64 - r64)            # (inf - inf where both overflowed: not selected)
    assert (d64[sel] <= bar[sel]).all(), (d64[sel] / bar[sel]).max()
    rel = np.abs(g64[both] - r64[both]) / np.maximum(np.abs(r64[both]), 1e-300)
    assert np.median(rel) < 1e-6, np.median(rel)
    odd = fin != np.isfinite(fr)            # one side overflowed, the other did not: only at the threshold
    assert (np.abs(np.where(np.isfinite(fr), fr, ref)[odd]) > 1e37).all() and odd.mean() < 1e-3
    ok = both & (np.abs(ref) > 1e-30)
    assert (np.signbit(fr[ok]) == np.signbit(ref[ok])).all()


def test_p4_golden(S):
    z = np.load(os.path.join(GOLD, "p4_sample_nonlinear.npz"))
    gold = z["audio"]  # [T][2]
    p = S.Patch(48000, int(z["buffer_size"]), 2)
    S.build_p4(p)
    p.configure_voices(2)
    out = p.render_channels(gold.shape[0], 1)
    for v in range(2):
        np.testing.assert_array_equal(bits(out[1, :, v]), bits(gold[:, 1]))
        assert_close(out[0, :, v], gold[:, 0])


def test_wave_roundtrip_and_state_readback(S):
    p = S.Patch(48000, 64, 2)
    smp, out = p.add_module(S.MOD_SAMPLE), p.add_module(S.MOD_OUTPUT)
    w = S.p4_wave(33)
    p.set_wave(smp, w, 22050.0)
    got, sr = p.get_wave(smp)
    np.testing.assert_array_equal(got, w)
    assert sr == 22050.0 and p.get_field(smp, S.SAMPLE_WAVE_NEW) == 1.0
    p.connect(smp, 0, out, 0)
    p.configure_voices(5)
    fr, _ = p.render(10)
    assert (fr[0] == w[0]).all()                       # never triggered: holds samples[0]
    assert (p.get_voice_field(smp, S.SAMPLE_PLAYING) == 0).all()


def test_dist_reduce_mix_through_rccl(S):
    """srack_dist_reduce_mix is ncclReduce(sum, f32, root) on the caller's communicator: exercised here with a one-rank
    RCCL communicator made through the library's own C API (the N > 1 arithmetic is covered by tests/test_dist.py)."""
    import ctypes as C
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        try:
            rccl = C.CDLL("/opt/rocm/lib/librccl.so")
        except OSError:
            pytest.skip("librccl not found")
    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        p = S.Patch(48000, 64, 2)
        S.build_p1(p, adsr="finite", lfo_val=2.0)
        p.configure_voices(100)
        T = 700
        _, want = p.render(T)
        q = S.Patch(48000, 64, 2)
        S.build_p1(q, adsr="finite", lfo_val=2.0)
        q.configure_voices(100)
        d_mix = C.c_void_p()
        assert S.lib.srack_device_alloc(C.byref(d_mix), 2 * T * 4) == 0
        q.render_raw(T, None, d_mix, 0, None)
        assert S.lib.srack_dist_reduce_mix(comm, d_mix, 2 * T, 0, None) == 0, S.lib.srack_last_error()
        got = np.empty((2, T), dtype=np.float32)
        assert S.lib.srack_device_to_host(got.ctypes.data_as(C.c_void_p), d_mix, got.nbytes, None) == 0 and S.lib.srack_device_sync(None) == 0
        S.lib.srack_device_free(d_mix)
        np.testing.assert_array_equal(got, want)
        assert S.lib.srack_dist_reduce_mix(None, d_mix, 1, 0, None) == S.ERR_INVALID
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_reserve_changes_nothing_but_the_first_call(S):
    """srack_render_reserve does the first-use set-up ahead of time: same samples, voice state untouched."""
    def make():
        p = S.Patch(48000, 1024, 2)
        ids = S.build_p3(p)
        p.configure_voices(90)
        p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, np.linspace(-1, 0, 90).astype(np.float32))
        return p, ids
    a, ids = make()
    a.reserve(5000)
    assert (a.get_voice_field(ids["osc"], S.OSC_POS) == 0).all()
    fa, ma = a.render(5000)
    b, _ = make()
    fb, mb = b.render(5000)
    np.testing.assert_array_equal(bits(fa), bits(fb))
    np.testing.assert_array_equal(bits(ma), bits(mb))


def test_long_render_crosses_segments_with_a_control_pipeline(S, oracle):
    """Three seconds of P3 (two segment borders at 65536 and 131072 samples, pipelined control units refilling each time)."""
    V, T = 40, 144000
    transpose = np.linspace(-2.0, 0.0, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 1024, 2)
    ids = S.build_p3(o)
    ref, _ = o.render_batch(V, T, [(ids["transpose"], S.MATH_CONSTANT, transpose)], threads=8)
    for flags in (0, 2, 32):
        p = S.Patch(48000, 1024, 2)
        S.build_p3(p)
        p.configure_voices(V)
        p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, transpose)
        fr, mix = p.render(T, flags=flags)
        assert_close(fr[0], ref[0])
        np.testing.assert_array_equal(fr[1], ref[1])
        want = ref.astype(np.float64).sum(axis=2)
        scale = np.abs(ref.astype(np.float64)).sum(axis=2)
        assert (np.abs(mix - want) <= 2e-5 * np.maximum(scale, 1.0)).all()


# ---- NoiseModule (oscillator.rs:308-393): the library's counter-based stream, bit for bit against the oracle ------------------
def _noise_patch(g, S, seed, first_voice):
    nz, vcf, vca, out = g.add_module(S.MOD_NOISE), g.add_module(S.MOD_MOOG_FILTER), g.add_module(S.MOD_VCA), g.add_module(S.MOD_OUTPUT)
    g.connect(nz, 0, vcf, 0)
    g.connect(vcf, 0, vca, 0)
    g.connect(nz, 0, vca, 1)       # noise as a CV too: the VCA opens where the noise is positive
    g.connect(vca, 0, out, 0)
    g.connect(nz, 0, out, 1)       # channel 1 = the raw noise
    g.set_noise_seed(seed, first_voice)
    return dict(nz=nz, vcf=vcf)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", MODES)
@pytest.mark.parametrize("V", [1, 70, 300])
def test_noise_vs_oracle(S, oracle, V, flags):
    T, seed, first = 700, 0xC0FFEE1234, 2**33 + 5
    o = oracle.OraclePatch(48000, 64, 2)
    ids = _noise_patch(o, S, seed, first)
    cut = np.linspace(0.05, 0.6, V).astype(np.float32)
    ref, ref_mix = o.render_batch(V, T, [(ids["vcf"], S.VCF_FREQ, cut)], mix=True, threads=4)
    p = S.Patch(48000, 64, 2)
    _noise_patch(p, S, seed, first)
    p.configure_voices(V)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    fr, mix = p.render(T, flags=flags)
    np.testing.assert_array_equal(bits(fr[1]), bits(ref[1]))           # the draw itself: exact in every mode
    if flags & 1:
        np.testing.assert_array_equal(bits(fr[0]), bits(ref[0]))       # exact modes: the ladder and the VCA too
    else:
        assert_close(fr[0], ref[0])
    scale = np.abs(ref.astype(np.float64)).sum(axis=2)
    assert (np.abs(mix - ref_mix) <= 2e-5 * np.maximum(scale, 1.0)).all()
    assert len({fr[1, :, v].tobytes() for v in range(V)}) == V         # one stream per voice


@pytest.mark.gpu
def test_noise_is_chunk_and_shard_invariant(S):
    V, T = 200, 1000
    def make(first, voices):
        p = S.Patch(48000, 32, 2)
        _noise_patch(p, S, 77, first)
        p.configure_voices(voices)
        return p
    whole, _ = make(0, V).render(T)
    q = make(0, V)
    parts = [q.render(n)[0] for n in (1, 31, 32, 500, 436)]
    np.testing.assert_array_equal(bits(np.concatenate(parts, axis=1)), bits(whole))   # sample n is a function of n, not of the chunking
    lo, hi = make(0, 120).render(T)[0], make(120, 80).render(T)[0]                    # two ranks: voices [0,120) and [120,200)
    np.testing.assert_array_equal(bits(np.concatenate([lo, hi], axis=2)), bits(whole))
    r = make(0, V)
    r.set_noise_seed(78)
    r.configure_voices(V)
    assert not np.array_equal(r.render(T)[0][1], whole[1])


@pytest.mark.gpu
def test_noise_statistics_at_scale(S):
    """65 536 voices x 512 samples of raw noise: the grid of rand's Standard f32, mean, variance, no voice-to-voice correlation."""
    V, T = 65536, 512
    p = S.Patch(48000, 1024, 1)
    nz, out = p.add_module(S.MOD_NOISE), p.add_module(S.MOD_OUTPUT)
    p.connect(nz, 0, out, 0)
    p.configure_voices(V)
    fr, mix = p.render(T)
    x = fr[0].astype(np.float64)
    k = (x + 1.0) * 2.0 ** 23
    assert (k == np.round(k)).all() and x.min() >= -1.0 and x.max() < 1.0
    n = x.size
    assert abs(x.mean()) < 5 / np.sqrt(3 * n) and abs(x.var() - 1 / 3) < 1e-3
    assert abs(np.corrcoef(x[:, 0::2].ravel(), x[:, 1::2].ravel())[0, 1]) < 5 / np.sqrt(n / 2)   # neighbouring voices
    assert abs(np.corrcoef(x[:-1].ravel(), x[1:].ravel())[0, 1]) < 5 / np.sqrt(n)                 # consecutive samples
    counts = np.bincount((k.astype(np.int64) >> 18).ravel(), minlength=64)
    assert 25 < ((counts - n / 64) ** 2 / (n / 64)).sum() < 120
    assert (np.abs(mix[0] - x.sum(axis=1)) <= 1e-5 * np.maximum(np.abs(x).sum(axis=1), 1.0)).all()


# ---- FreeverbModule: the module's routing (freeverb.rs) and the restated freeverb crate, GPU against the oracle ----------------
def _freeverb_patch(g, S, params=(), right=True):
    osc, osc2, fv, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_FREEVERB), g.add_module(S.MOD_OUTPUT)
    g.set_field(osc, S.OSC_VAL, -1.0)
    g.set_field(osc2, S.OSC_VAL, 0.37)
    g.connect(osc, S.OSC_OUT_SAW, fv, 0)
    if right:
        g.connect(osc2, S.OSC_OUT_SQUARE, fv, 1)
    g.connect(fv, 0, out, 0)
    g.connect(fv, 1, out, 1)
    for f, v in params:
        g.set_field(fv, f, v)
    return dict(osc=osc, fv=fv)


FV_PARAMS = [pytest.param((), True, id="defaults"),
             pytest.param(((0, 1.7), (2, 0.6), (3, 1.0), (4, 0.9), (5, 0.8)), True, id="long-wide-dry"),
             pytest.param(((1, 1), (5, 0.25), (4, 0.3)), False, id="frozen-left-only")]


@pytest.mark.gpu
@pytest.mark.parametrize("flags", MODES_BASE)   # (the reverb is not covered by the kernel generator: the interpreter renders it)
@pytest.mark.parametrize("params,right", FV_PARAMS)
def test_freeverb_per_voice_vs_oracle(S, oracle, params, right, flags):
    """Per-voice detune in front of the reverb => one reverb per voice (24 delay lines each, in HBM)."""
    V, T = 70, 4200                                    # the longest comb is 1785 samples at 48 kHz: two trips round, and past a 4096-sample launch
    det = np.linspace(-2.0, 1.0, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 64, 2)
    ids = _freeverb_patch(o, S, params, right)
    ref, _ = o.render_batch(V, T, [(ids["osc"], S.OSC_VAL, det)], threads=8)
    p = S.Patch(48000, 64, 2)
    _freeverb_patch(p, S, params, right)
    p.configure_voices(V)
    p.set_voice_field(ids["osc"], S.OSC_VAL, det)
    fr, _ = p.render(T, flags=flags)
    if flags & 1:
        np.testing.assert_array_equal(bits(fr), bits(ref))
    else:
        assert_close(fr, ref)
    assert np.abs(ref[:, 3000:]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1, 2, 3, 9, 11])
def test_freeverb_voice_invariant_runs_in_the_control_program(S, oracle, flags):
    """Nothing upstream of the reverb differs between voices: it is evaluated once, by a control unit, and reaches the voices as tracks."""
    V, T = 96, 5000
    o = oracle.OraclePatch(48000, 128, 2)
    _freeverb_patch(o, S, ((2, 0.8), (5, 0.4)))
    ref = o.render(T)
    p = S.Patch(48000, 128, 2)
    _freeverb_patch(p, S, ((2, 0.8), (5, 0.4)))
    p.configure_voices(V)
    fr, mix = p.render(T, flags=flags)
    assert "ctl[" in p.info() and "tracks=2" in p.info(), p.info()   # (info describes the program the last render used)
    for v in (0, 17, V - 1):
        if flags & 1:
            np.testing.assert_array_equal(bits(fr[:, :, v]), bits(ref))
        else:
            assert_close(fr[:, :, v], ref)


@pytest.mark.gpu
def test_freeverb_continues_across_calls(S):
    V, T = 40, 9000
    det = np.linspace(-1.0, 1.0, V).astype(np.float32)
    for flags in (0, 1):
        outs = []
        for parts in ([T], [1, 63, 4097, 1000, 3839]):
            p = S.Patch(48000, 32, 2)
            ids = _freeverb_patch(p, S)
            p.configure_voices(V)
            p.set_voice_field(ids["osc"], S.OSC_VAL, det)
            outs.append(np.concatenate([p.render(n, flags=flags)[0] for n in parts], axis=1))
        np.testing.assert_array_equal(bits(outs[0]), bits(outs[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", MODES)
def test_adsr_sustain_holds_on_a_nan_gate(S, oracle, flags):
    """adsr.rs:175 spells Sustain's exit `gate <= 0.0`: a NaN gate is neither high nor low (tools/fuzz_soak.py found the shortcut)."""
    from tests.test_oracle import _nan_gate_patch
    o = oracle.OraclePatch(48000, 1, 2)
    _nan_gate_patch(o)
    ref = o.render(600)
    for V in (1, 70):
        p = S.Patch(48000, 1, 2)
        _nan_gate_patch(p)
        p.configure_voices(V)
        fr, _ = p.render(600, flags=flags)
        for v in (0, V - 1):
            np.testing.assert_array_equal(bits(fr[0, :, v]), bits(ref[0]))
            np.testing.assert_array_equal(np.isnan(fr[1, :, v]), np.isnan(ref[1]))
    assert (ref[0][200:] == np.float32(0.6)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("flags", MODES)
def test_filter_clamps_a_nan_to_plus_one(S, oracle, flags):
    """filter.rs:89 `x.min(1.0).max(-1.0)`: Rust's min / max drop a NaN operand, so a NaN that reaches the ladder turns its states
    into +1.0 (not NaN).  Exact modes reproduce that; the default mode's one-instruction clamp (v_med3) saturates to -1.0 instead —
    the documented difference for a patch that has already blown up (modules.hip.h, clamp1)."""
    def build(g):
        s_, m, d, vcf, out = g.add_module(S.MOD_MATH), g.add_module(S.MOD_MATH), g.add_module(S.MOD_MATH), g.add_module(S.MOD_MOOG_FILTER), g.add_module(S.MOD_OUTPUT)
        g.set_field(s_, S.MATH_CONSTANT, 1.0)
        g.connect(m, 0, s_, 0)
        g.set_field(m, S.MATH_CONSTANT, 2.0)
        g.set_field(m, S.MATH_OPERATION, S.MATH_MULTIPLY)
        g.connect(s_, 0, m, 0)             # m = 2 (m + 1): overflows to inf after ~128 one-sample blocks
        g.set_field(d, S.MATH_OPERATION, S.MATH_SUBTRACT)
        g.connect(m, 0, d, 0)
        g.connect(m, 0, d, 1)              # m - m: 0, then NaN
        g.connect(d, 0, vcf, 0)
        g.connect(vcf, 0, out, 0)          # low-pass
        g.connect(vcf, 1, out, 1)          # band-pass = 3 (b3 - b4)
    o = oracle.OraclePatch(48000, 1, 2)
    build(o)
    ref = o.render(400)
    assert (ref[0][200:] == 1.0).all() and (ref[1][200:] == 0.0).all()
    p = S.Patch(48000, 1, 2)
    build(p)
    p.configure_voices(70)
    fr, _ = p.render(400, flags=flags)
    for v in (0, 69):
        if flags & 1:
            np.testing.assert_array_equal(fr[:, :, v], ref)
        else:
            np.testing.assert_array_equal(fr[:, :150, v][np.isfinite(ref[:, :150]) & (ref[:, :150] == 0)], 0.0)   # before the NaN arrives
            assert (np.abs(fr[0, 200:, v]) == 1.0).all() and (fr[1, 200:, v] == 0.0).all()                        # saturated, finite


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [pytest.param(1, id="fused-exact"), pytest.param(3, id="interp-exact"), pytest.param(35, id="special-exact"),
                                   pytest.param(39, id="special-exact-nohoist"), pytest.param(7, id="interp-exact-nohoist")])
def test_filter_with_an_exploding_input(S, oracle, flags):
    """The exact ladder clamps with v_med3 only while that cannot be told from `min(1.0).max(-1.0)`: an input that grows through 1e6, 1e30
    and inf makes the cubic overflow into inf - inf = NaN inside the filter — the literal form has to take over in time, sample for sample."""
    def build(g):
        s_, m, vcf, out = g.add_module(S.MOD_MATH), g.add_module(S.MOD_MATH), g.add_module(S.MOD_MOOG_FILTER), g.add_module(S.MOD_OUTPUT)
        g.set_field(s_, S.MATH_CONSTANT, 1.0)
        g.connect(m, 0, s_, 0)
        g.set_field(m, S.MATH_CONSTANT, 1.7)
        g.set_field(m, S.MATH_OPERATION, S.MATH_MULTIPLY)
        g.connect(s_, 0, m, 0)             # m = 1.7 (m + 1): 1e6 after ~25 one-sample blocks, inf after ~170
        g.connect(m, 0, vcf, 0)
        g.connect(vcf, 0, out, 0)          # low-pass
        g.connect(vcf, 2, out, 1)          # high-pass = input - b4: follows the input up to inf
        return dict(vcf=vcf)
    V = 70
    res = np.linspace(0.0, 0.85, V).astype(np.float32)
    o = oracle.OraclePatch(48000, 1, 2)
    ids = build(o)
    ref, _ = o.render_batch(V, 300, [(ids["vcf"], S.VCF_RES, res)], threads=4)
    assert np.isinf(ref[1, 250:]).all() or np.isnan(ref[1, 250:]).any()
    p = S.Patch(48000, 1, 2)
    build(p)
    p.configure_voices(V)
    p.set_voice_field(ids["vcf"], S.VCF_RES, res)
    fr, _ = p.render(300, flags=flags)
    same = (bits(fr) == bits(ref)) | (np.isnan(fr) & np.isnan(ref))
    assert same.all(), f"{1 - same.mean():.4f} of the samples differ, first at {np.argwhere(~same)[0]}"


@pytest.mark.gpu
@pytest.mark.parametrize("flags", MODES)
def test_oscillator_increments_beyond_half_a_cycle(S, oracle, flags):
    """At sample rate 1000 the audible range reaches increments of 0.1 ... 1.1 cycles per sample.  Past 1/2 the two PolyBLEP windows
    overlap (oscillator.rs:53-66 tests `t < dt` first, then `t > 1.0 - dt`) and the branches no longer meet at a border — and an
    oscillator that starts at phase 0 lands exactly ON the border after one step.  (tools/fv_soak.py found the default mode deciding
    that border from a rounded f32 quotient: errors of 0.7.)"""
    sr, V, T = 1000, 64, 3000
    val = np.linspace(-3.0, 1.32, V).astype(np.float32)
    def build(g):
        osc, osc2, out = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_OUTPUT)
        g.connect(osc, S.OSC_OUT_SAW, out, 0)
        g.connect(osc2, S.OSC_OUT_SQUARE, out, 1)
        g.set_field(osc2, S.OSC_VAL, 0.5)
        g.connect(osc, S.OSC_OUT_SAW, osc2, 0)       # and one driven through its pitch CV
        return osc
    o = oracle.OraclePatch(sr, 64, 2)
    osc = build(o)
    ref, _ = o.render_batch(V, T, [(osc, S.OSC_VAL, val)], threads=8)
    delta = 440.0 * 2.0 ** val.astype(np.float64) / sr
    assert delta.min() < 0.1 and delta.max() > 1.05
    p = S.Patch(sr, 64, 2)
    build(p)
    p.configure_voices(V)
    p.set_voice_field(osc, S.OSC_VAL, val)
    fr, _ = p.render(T, flags=flags)
    if flags & 1:
        np.testing.assert_array_equal(bits(fr[0]), bits(ref[0]))
    assert_close(fr[0], ref[0])
    # (channel 1 integrates the saw into a pitch: exact oscillator is forced by the flattener, compared loosely in default modes)
    assert_close(fr[1], ref[1], tol=1e-5 if flags & 1 else 2.5)


@pytest.mark.gpu
@pytest.mark.parametrize("flags", FAST_MODES)
def test_p1_near_self_oscillation_stays_within_tolerance(S, oracle, flags):
    """Resonance 0.97, cutoffs up to 0.85, band-pass port: the ladder's feedback gain multiplies every rounding difference.  The default
    modes used to leave 1e-5 here (tools/shape_soak.py); the flattener now hands such a patch the literal ladder."""
    V, T = 70, 5000
    def build(g):
        ids = S.build_p1(g, adsr="finite", lfo_val=-4.0)
        g.set_field(ids["vcf"], S.VCF_RES, 0.97)
        g.set_field(ids["vcf"], S.VCF_EXP_AMT, 0.4)
        g.disconnect(ids["vca"], 0)
        g.connect(ids["vcf"], S.VCF_OUT_BANDPASS, ids["vca"], 0)
        return ids
    det, cut = np.linspace(-2.0, 2.0, V).astype(np.float32), np.linspace(0.1, 0.85, V).astype(np.float32)
    o = oracle.OraclePatch(8000, 1024, 2)
    ids = build(o)
    ref, _ = o.render_batch(V, T, [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)], threads=8)
    p = S.Patch(8000, 1024, 2)
    build(p)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    fr, _ = p.render(T, flags=flags)        # (one plane: both channels carry the VCA)
    assert_close(fr[0], ref[0])
    assert np.abs(ref).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3])
def test_state_carried_across_an_edit_by_hand(S, flags):
    """The documented route for keeping the voices running across a patch edit (srack_hip.h): read every state field back, set it
    per voice, edit, render on.  Exact modes: bit-identical to an uninterrupted render with the new parameter from that sample on."""
    V, T1, T2 = 70, 1500, 1700
    det = np.linspace(-2.0, 1.0, V).astype(np.float32)
    cut = np.linspace(0.05, 0.5, V).astype(np.float32)
    STATE = {S.MOD_OSCILLATOR: [S.OSC_POS, S.OSC_SYNC_LAST], S.MOD_MOOG_FILTER: list(range(S.VCF_ST_F, S.VCF_ST_RES + 1)),
             S.MOD_ADSR: [S.ADSR_PHASE, S.ADSR_MODE, S.ADSR_R_VAL, S.ADSR_FROM_A_VAL, S.ADSR_GATE_LAST]}
    def make():
        p = S.Patch(48000, 64, 2)
        ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
        p.configure_voices(V)
        p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
        return p, ids
    # reference run: the parameter changes after T1 samples, the voices keep running — emulated by carrying the state by hand too,
    # but into a FRESH patch built with the new parameter (so the two ways of getting there must agree) ...
    p, ids = make()
    a1 = p.render_channels(T1, flags)
    state = {(m, f): p.get_voice_field(m, f) for m in range(p.num_modules()) for f in STATE.get(p.module_type(m), [])}
    p.set_field(ids["vca"], S.VCA_NEGATIVE, 1)                      # the edit: re-flattens, the voices would restart ...
    for (m, f), vals in state.items():
        p.set_voice_field(m, f, vals if f == S.OSC_POS else vals.astype(np.float32))   # ... unless their state is put back
    a2 = p.render_channels(T2, flags)
    q, ids = make()
    q.set_field(ids["vca"], S.VCA_NEGATIVE, 1)
    for (m, f), vals in state.items():
        q.set_voice_field(m, f, vals if f == S.OSC_POS else vals.astype(np.float32))
    b2 = q.render_channels(T2, flags)
    np.testing.assert_array_equal(bits(a2), bits(b2))
    # ... and with an edit that changes nothing audible (negative = 0 again) the hand-carried render continues the first one exactly
    r, ids = make()
    whole = r.render_channels(T1 + T2, flags)
    s_, ids = make()
    s_.render_channels(T1, flags)
    st = {(m, f): s_.get_voice_field(m, f) for m in range(s_.num_modules()) for f in STATE.get(s_.module_type(m), [])}
    s_.set_field(ids["vca"], S.VCA_NEGATIVE, 0)
    for (m, f), vals in st.items():
        s_.set_voice_field(m, f, vals if f == S.OSC_POS else vals.astype(np.float32))
    cont = s_.render_channels(T2, flags)
    np.testing.assert_array_equal(bits(a1), bits(whole[:, :T1]))
    np.testing.assert_array_equal(bits(cont), bits(whole[:, T1:]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3, 0, 2, 35, 34])
@pytest.mark.parametrize("shape", ["p1", "p3", "p4"])
def test_keep_state_carries_the_voices_across_edits(S, shape, flags):
    """srack_patch_keep_state: an edit between renders re-flattens the patch but the modules' state (per voice, and once for the
    modules the control program evaluates) is carried over.  An edit that changes nothing audible must therefore continue the render:
    bit for bit in exact modes; in default modes the fused kernels' 64-bit fixed-point phase passes through a double (2^-53)."""
    V, T1, T2, T3 = 70, 1500, 900, 1300
    def make():
        p = S.Patch(48000, 64, 2)
        if shape == "p1":
            ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
            p.configure_voices(V)
            p.set_voice_field(ids["osc_a"], S.OSC_VAL, np.linspace(-2.0, 1.0, V).astype(np.float32))
            p.set_voice_field(ids["vcf"], S.VCF_FREQ, np.linspace(0.05, 0.5, V).astype(np.float32))
            knob = (ids["vca"], S.VCA_NEGATIVE, 0)
        elif shape == "p4":   # the sample player: its `wavebox.new` must count as consumed once it has run (tools/fuzz_soak_keep.py)
            ids = S.build_p4(p)
            p.configure_voices(V)
            depth, expo = S.p4_voice_params(V)
            p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth)
            p.set_voice_field(ids["shaper"], S.NONLIN_CONSTANT, expo)
            knob = (ids["shaper"], S.NONLIN_CONSTANT, 0.75)
        else:
            ids = S.build_p3(p, clock_val=-3.0, length=6)
            p.configure_voices(V)
            p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, np.linspace(-2.0, 0.5, V).astype(np.float32))
            p.set_voice_field(ids["vcf"], S.VCF_FREQ, np.linspace(0.05, 0.4, V).astype(np.float32))
            knob = (ids["vca"], S.VCA_NEGATIVE, 0)
        return p, knob
    whole = make()[0].render_channels(T1 + T2 + T3, flags)
    p, knob = make()
    p.keep_state(True)
    parts = [p.render_channels(T1, flags)]
    p.set_field(*knob)                                 # an edit (same value): re-flatten
    parts.append(p.render_channels(T2, flags))
    p.set_field(*knob)
    parts.append(p.render_channels(T3, flags))
    got = np.concatenate(parts, axis=1)
    if flags & 1:
        np.testing.assert_array_equal(bits(got), bits(whole))
    else:
        assert_close(got, whole)
    # without keep_state the same sequence restarts the voices at every edit
    q, knob = make()
    q.render_channels(T1, flags)
    q.set_field(*knob)
    np.testing.assert_array_equal(bits(q.render_channels(T2, flags)), bits(whole[:, :T2]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3, 0, 2, 35])
@pytest.mark.parametrize("B,reverb", [(1, True), (8, True), (64, True), (1, False), (1024, False)])   # (without the reverb: the fused FM kernels)
def test_keep_state_carries_feedback_rings_and_reverb_lines(S, B, reverb, flags):
    """... and the state that is not a module field: the delay ring of a feedback edge (a state row of the voice table for
    buffer_size <= 16, a ring in HBM above) and a reverb's 24 delay lines, device to device into the re-flattened program."""
    V, T1, T2 = 66, 2100, 1900
    if reverb and flags & 32:
        pytest.skip("the reverb is the interpreter's: no specialised kernel for this program")
    def make():
        p = S.Patch(48000, B, 2)
        ids = S.build_p2(p, beta=0.25, index=0.8)                      # FM pair: OSC_M.sine -> MUL_FB -> OSC_M.cv is a delayed edge
        # (under keep_state every planned module is evaluated, wired or not: the specialised runs edit a Math module instead)
        fv = p.add_module(S.MOD_FREEVERB if not flags & 32 else S.MOD_MATH)
        if reverb:
            p.disconnect(ids["out"], 1)
            p.connect(ids["osc_c"], S.OSC_OUT_SINE, fv, 0)
            p.connect(fv, 1, ids["out"], 1)                             # channel 1 through a reverb
        p.configure_voices(V)
        p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, np.linspace(0.0, 0.5, V).astype(np.float32))
        p.set_voice_field(ids["osc_c"], S.OSC_VAL, np.linspace(-1.0, 1.0, V).astype(np.float32))
        return p, ((fv, S.FREEVERB_DRY, 0.0) if not flags & 32 else (fv, S.MATH_CONSTANT, 0.5))
    whole = make()[0].render_channels(T1 + T2, flags)
    p, knob = make()
    p.keep_state(True)
    first = p.render_channels(T1, flags)
    p.set_field(*knob)
    p.set_field(*knob)                                                  # two edits before the next render: still one carry
    second = p.render_channels(T2, flags)
    got = np.concatenate([first, second], axis=1)
    if flags & 1:
        np.testing.assert_array_equal(bits(got), bits(whole))
    else:
        assert_close(got, whole)
    assert np.abs(whole[1, T1:]).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3, 5])
def test_keep_state_parameter_change_matches_the_oracle(S, oracle, flags):
    """The reference's case: a slider moves while the graph runs — the next block sees the new value, module state is untouched.
    Oracle: one stateful patch object per sampled voice, set_field between two renders (block-aligned).  GPU: keep_state."""
    V, B, T1, T2 = 70, 64, 1536, 2048
    det = np.linspace(-2.0, 1.0, V).astype(np.float32)
    cut = np.linspace(0.05, 0.5, V).astype(np.float32)
    def edits(ids):
        return [(ids["vcf"], S.VCF_RES, 0.8), (ids["adsr"], S.ADSR_S_VAL, 0.9), (ids["osc_lfo"], S.OSC_VAL, -2.0), (ids["vca"], S.VCA_NEGATIVE, 1)]
    p = S.Patch(48000, B, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    p.keep_state(True)
    a = p.render_channels(T1, flags)
    for m, f, v in edits(ids):
        p.set_field(m, f, v)
    b = p.render_channels(T2, flags)
    for v in (0, 17, 42, 69):
        o = oracle.OraclePatch(48000, B, 2)
        oi = S.build_p1(o, adsr="finite", lfo_val=-3.0)
        o.set_field(oi["osc_a"], S.OSC_VAL, float(det[v]))
        o.set_field(oi["vcf"], S.VCF_FREQ, float(cut[v]))
        ra = o.render(T1)
        for m, f, x in edits(oi):
            o.set_field(m, f, x)
        rb = o.render(T2)
        np.testing.assert_array_equal(bits(a[:, :, v]), bits(ra))
        np.testing.assert_array_equal(bits(b[:, :, v]), bits(rb))
    assert np.abs(b).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3, 5])
def test_keep_state_rewiring_matches_the_oracle(S, oracle, flags):
    """... and a cable plugged in while the graph runs: the envelope also sweeps the filter from the next block on, a second
    oscillator is added and mixed in.  State of everything that already ran is untouched (stateful oracle objects per sampled voice)."""
    V, B, T1, T2 = 70, 64, 1536, 2048
    det = np.linspace(-2.0, 1.0, V).astype(np.float32)
    cut = np.linspace(0.05, 0.4, V).astype(np.float32)
    def rewire(g, ids):
        g.connect(ids["adsr"], 0, ids["vcf"], 1)                      # envelope -> cutoff CV
        osc2, mix = g.add_module(S.MOD_OSCILLATOR), g.add_module(S.MOD_MONO_MIXER)
        g.set_field(osc2, S.OSC_VAL, 0.5)
        g.disconnect(ids["vcf"], 0)
        g.connect(ids["osc_a"], S.OSC_OUT_SAW, mix, 0)
        g.connect(osc2, S.OSC_OUT_SQUARE, mix, 1)
        g.connect(mix, 0, ids["vcf"], 0)                              # saw + a new square into the filter
    p = S.Patch(48000, B, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    p.keep_state(True)
    a = p.render_channels(T1, flags)
    rewire(p, ids)
    b = p.render_channels(T2, flags)
    for v in (0, 23, 69):
        o = oracle.OraclePatch(48000, B, 2)
        oi = S.build_p1(o, adsr="finite", lfo_val=-3.0)
        o.set_field(oi["osc_a"], S.OSC_VAL, float(det[v]))
        o.set_field(oi["vcf"], S.VCF_FREQ, float(cut[v]))
        ra = o.render(T1)
        rewire(o, oi)
        rb = o.render(T2)
        np.testing.assert_array_equal(bits(a[:, :, v]), bits(ra))
        np.testing.assert_array_equal(bits(b[:, :, v]), bits(rb))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3])
def test_keep_state_runs_modules_nobody_hears_yet(S, oracle, flags):
    """The reference's execute() ticks every module of the workspace, wired to the output or not (plan_execution covers all_modules).
    With keep_state so does the flattened program: an oscillator patched in later has been running all along."""
    V, B, T1, T2 = 66, 64, 1024, 1024
    def build(g):
        ids = S.build_p1(g, adsr="finite", lfo_val=-3.0)
        ids["spare"] = g.add_module(S.MOD_OSCILLATOR)       # free-running, wired to nothing
        g.set_field(ids["spare"], S.OSC_VAL, -1.3)
        return ids
    def patch_in(g, ids):
        g.disconnect(ids["out"], 1)
        g.connect(ids["spare"], S.OSC_OUT_SAW, ids["out"], 1)
    p = S.Patch(48000, B, 2)
    ids = build(p)
    p.configure_voices(V)
    p.keep_state(True)
    p.render_channels(T1, flags)
    patch_in(p, ids)
    b = p.render_channels(T2, flags)
    o = oracle.OraclePatch(48000, B, 2)
    oi = build(o)
    o.render(T1)
    patch_in(o, oi)
    rb = o.render(T2)
    for v in (0, V - 1):
        np.testing.assert_array_equal(bits(b[:, :, v]), bits(rb))
    assert abs(float(rb[1][0]) + 1.0) > 1e-3        # the saw did not start from phase 0


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3])
def test_saving_a_running_rack_under_keep_state(S, flags):
    """The app saves the rack as it plays (ui.rs:98-114 serialises the live module structs).  With keep_state, save_srk writes the
    running state of voice 0: a patch loaded from that file continues voice 0's render bit for bit (P1: no port buffer is ever read)."""
    V, B, T1, T2 = 5, 64, 1408, 1600
    def make():
        p = S.Patch(48000, B, 2)
        ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
        p.set_field(ids["osc_a"], S.OSC_VAL, 0.3)
        p.configure_voices(V)
        return p
    whole = make().render_channels(T1 + T2, flags)
    p = make()
    p.keep_state(True)
    p.render_channels(T1, flags)
    data = p.save_srk()
    q = S.Patch.load_srk(data, 48000, B, 2)
    q.configure_voices(1)
    cont = q.render_channels(T2, flags)
    np.testing.assert_array_equal(bits(cont[:, :, 0]), bits(whole[:, T1:, 0]))
    r = make()                                  # without keep_state the file holds the stored (initial) state
    r.render_channels(T1, flags)
    fresh = S.Patch.load_srk(r.save_srk(), 48000, B, 2)
    fresh.configure_voices(1)
    np.testing.assert_array_equal(bits(fresh.render_channels(T2, flags)[:, :, 0]), bits(whole[:, :T2, 0]))


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3])
def test_keep_state_runs_the_rack_while_the_output_is_unwired(S, oracle, flags):
    """Nothing wired into the OutputModule yet (or the output muted): the render is silence, but under keep_state the planned modules
    keep ticking as in the reference's execute() — the phases, envelopes and sample counter a later patch cable finds are the
    reference's, not frozen ones."""
    V, B, T1, T2 = 66, 64, 1536, 1024   # (whole blocks: the oracle, like the reference, only ticks in units of buffer_size)
    def build(g):
        ids = S.build_p1(g, adsr="finite", lfo_val=-3.0)
        g.disconnect(ids["out"], 0)
        g.disconnect(ids["out"], 1)
        return ids
    def patch_in(g, ids):
        g.connect(ids["vca"], 0, ids["out"], 0)
        g.connect(ids["osc_a"], S.OSC_OUT_SAW, ids["out"], 1)
    p = S.Patch(48000, B, 2)
    ids = build(p)
    p.configure_voices(V)
    p.keep_state(True)
    fr, mix = p.render(T1, flags=flags)
    assert fr.shape[0] == 0 and not mix.any()
    patch_in(p, ids)
    b = p.render_channels(T2, flags)
    o = oracle.OraclePatch(48000, B, 2)
    oi = build(o)
    o.render(T1)
    patch_in(o, oi)
    rb = o.render(T2)
    for v in (0, V - 1):
        np.testing.assert_array_equal(bits(b[:, :, v]), bits(rb))
    assert np.abs(rb[0]).max() > 0.01 and abs(float(rb[1][0]) + 1.0) > 1e-3   # audible, and the saw did not start from phase 0


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [1, 3])
def test_keep_state_host_write_to_a_state_field_wins(S, flags):
    """Under keep_state a re-flatten starts from the running state — except where the host has just written a state field itself
    (a phase reset, an envelope forced to None): that value is the one the next render starts from."""
    V, B, T = 5, 64, 1024
    p = S.Patch(48000, B, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=-3.0)
    p.configure_voices(V)
    det, _ = S.p1_voice_params(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.keep_state(True)
    p.render_channels(T, flags)
    running = p.get_voice_field(ids["osc_a"], S.OSC_POS)
    assert (running != 0.25).all()
    p.set_field(ids["osc_a"], S.OSC_POS, 0.25)                       # all voices (replaces what an earlier carry left per voice)
    p.set_field(ids["vcf"], S.VCF_RES, 0.4)                          # plus an ordinary parameter edit
    got = p.get_voice_field(ids["osc_a"], S.OSC_POS)
    np.testing.assert_array_equal(got, np.full(V, 0.25))
    filt = p.get_voice_field(ids["vcf"], S.VCF_ST_B4)                # ... while everything else keeps running
    assert np.abs(filt).max() > 0
    p.render_channels(16, flags)
    pos = np.full(V, 0.25)
    for _ in range(16):
        pos = np.fmod(pos + 440.0 * np.power(2.0, det.astype(np.float64)) / 48000.0, 1.0)
    np.testing.assert_array_equal(p.get_voice_field(ids["osc_a"], S.OSC_POS), pos)
    per_voice = np.linspace(0.1, 0.5, V)
    p.set_voice_field(ids["osc_a"], S.OSC_POS, per_voice)            # the per-voice form
    p.set_field(ids["vcf"], S.VCF_RES, 0.45)
    np.testing.assert_array_equal(p.get_voice_field(ids["osc_a"], S.OSC_POS), per_voice)


@pytest.mark.gpu
def test_kernel_timer_is_off_until_asked_for(S):
    p = S.Patch(48000, 1024, 2)
    S.build_p1(p)
    p.configure_voices(64)
    lib = S.lib
    import ctypes as C
    p.render(2048)
    ms, n = C.c_double(), C.c_int()
    assert lib.srack_render_kernel_ms(p.h, C.byref(ms), C.byref(n), 0) == 0 and n.value == 0   # nothing was recorded; this call arms
    p.render(2048)
    assert lib.srack_render_kernel_ms(p.h, C.byref(ms), C.byref(n), 1) == 0 and n.value >= 1 and ms.value > 0
    assert lib.srack_render_kernel_ms(p.h, C.byref(ms), C.byref(n), -1) == 0 and n.value == 0  # read + disarm
    p.render(2048)
    assert lib.srack_render_kernel_ms(p.h, C.byref(ms), C.byref(n), -1) == 0 and n.value == 0


def _vibrato(g, S, loop):
    """P1 whose audio oscillator's pitch is modulated by a SAW (an approximated value reaches a pitch input): by a free-running LFO
    (feed-forward), or — loop — by the audio oscillator's own saw through a Multiply (a loop through the pitch input)."""
    ids = S.build_p1(g, adsr="finite", lfo_val=-3.0)
    depth = g.add_module(S.MOD_MATH)
    g.set_field(depth, S.MATH_OPERATION, S.MATH_MULTIPLY)
    g.set_field(depth, S.MATH_CONSTANT, 0.03)
    if loop:
        g.connect(ids["osc_a"], S.OSC_OUT_SAW, depth, 0)
    else:
        lfo = g.add_module(S.MOD_OSCILLATOR)
        g.set_field(lfo, S.OSC_VAL, -6.0)
        g.connect(lfo, S.OSC_OUT_SAW, depth, 0)
    g.connect(depth, 0, ids["osc_a"], 0)   # port 0: the 1 V/oct CV
    return ids


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [pytest.param(0, id="default"), pytest.param(2, id="interp"), pytest.param(34, id="special")])
@pytest.mark.parametrize("loop", [False, True], ids=["feed-forward", "loop"])
def test_a_saw_that_reaches_a_pitch(S, oracle, loop, flags):
    """A saw that modulates a pitch is INTEGRATED by the phase accumulator behind it: the default mode's f32 PolyBLEP (a biased 1e-7) left
    2e-4 on the carrier after one second.  Feed-forward, the producing oscillator alone gets the exact PolyBLEP (OSC_EXACT_BLEP) and the
    rest of the patch keeps the default arithmetic; with a loop through the pitch input the oscillator in the loop is evaluated exactly as a
    whole (approx.cpp: the loop's gain has no bound) — its saw the reference's to the bit — and the ladder behind it, which feeds nothing
    back, keeps its contracted form.  Either way the full second stays inside the contract."""
    V, T, B = 24, 48000, 64
    det, cut = S.p1_voice_params(V)
    o = oracle.OraclePatch(48000, B, 2)
    ids = _vibrato(o, S, loop)
    ref, _ = o.render_batch(V, T, [(ids["osc_a"], S.OSC_VAL, det), (ids["vcf"], S.VCF_FREQ, cut)], threads=8)
    p = S.Patch(48000, B, 2)
    _vibrato(p, S, loop)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    src = p.kernel_source(flags)
    assert "vcf_run<true>" in src and ("; exact osc 0]" in src.split("\n", 1)[0]) == loop   # the default flavour; in the loop: oscillator 0 exact as a whole
    fr = p.render_channels(T, flags)
    assert assert_close(fr, ref) < 5e-6
    assert np.abs(ref[0]).max() > 0.05
    if loop:   # ... and the oscillator alone (its saw straight to the output) is the reference's to the bit
        o2, p2 = oracle.OraclePatch(48000, B, 2), S.Patch(48000, B, 2)
        for g in (o2, p2):
            i2 = _vibrato(g, S, True)
            g.connect(i2["osc_a"], S.OSC_OUT_SAW, i2["out"], 0)
            g.connect(i2["osc_a"], S.OSC_OUT_SAW, i2["out"], 1)
        ref2, _ = o2.render_batch(V, T, [(ids["osc_a"], S.OSC_VAL, det)], threads=8)
        p2.configure_voices(V)
        p2.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
        np.testing.assert_array_equal(bits(p2.render_channels(T, flags)), bits(ref2))


@pytest.mark.parametrize("B", [1, 1024])
def test_exact_mode_frames_do_not_depend_on_the_sharding(S, B):
    """What lets an N-GPU render be diffed against one GPU: in the exact mode a voice's frames are bit-identical whatever shard it is rendered
    in — here config 4's patch (whose DEFAULT kernels vote per wave on what their 64 voices allow, so that a voice's last bits can depend
    on its neighbours: DESIGN section 5), global voices [16384, 24576) as a shard of 8 192 and inside a shard of 65 536."""
    T = 6000
    name = "cfg4" if B == 1 else "cfg4_b1024"

    def shard(first, n):
        Bs, build, overrides = S.bench_workload(name, n, first_voice=first)
        p = S.Patch(48000, Bs, 2)
        ids = build(p)
        p.configure_voices(n)
        for m, f, v in overrides(ids):
            p.set_voice_field(m, f, v)
        fr, mix = p.render(T, flags=S.RENDER_EXACT_OSC)
        return fr[0]

    big = shard(0, 65536)
    small = shard(16384, 8192)
    np.testing.assert_array_equal(bits(small), bits(big[:, 16384:24576]))
    assert np.abs(small).max() > 0.5


def test_read_backs_from_several_host_threads_at_once(S):
    """`srack_device_to_host` holds no lock across a copy since round 6 (a pool of bounce sets per device, helper threads for large copies):
    six host threads read overlapping windows of one rendered plane at the same time — sizes on both sides of the 8 MB bounce chunk and of
    the 64 MB helper-thread threshold — and every one of them gets the bytes a single-threaded read-back gets."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    V, T = 65536, 1280   # 320 MB of frames
    det, cut = S.p1_voice_params(V)
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p, adsr="finite", lfo_val=0.0)   # (a gate that opens within the first milliseconds)
    p.configure_voices(V)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    d_fr = C.c_void_p()
    assert S.lib.srack_device_alloc(C.byref(d_fr), T * V * 4) == 0
    try:
        p.render_raw(T, d_fr, None, 0, None)
        assert S.lib.srack_device_sync(None) == 0
        whole = np.empty(T * V, dtype=np.float32)
        assert S.lib.srack_device_to_host(whole.ctypes.data_as(C.c_void_p), d_fr, whole.nbytes, None) == 0
        assert np.abs(whole).max() > 1e-3 and (whole != 0).mean() > 0.5
        windows = [(0, 1 << 20), (3 << 20, 9 << 20), (1 << 20, 70 << 20), (5, 4099), (11 << 20, 80 << 20), (0, T * V * 4), (64 << 20, 200 << 20), (17, 33 << 20)]   # (byte offset, bytes)
        windows = [(off // 4 * 4, n // 4 * 4) for off, n in windows]

        def read(w):
            off, n = w
            out = np.empty(n // 4, dtype=np.float32)
            rc = S.lib.srack_device_to_host(out.ctypes.data_as(C.c_void_p), C.c_void_p(d_fr.value + off), n, None)
            return rc, out

        for _ in range(2):
            with ThreadPoolExecutor(6) as pool:
                got = list(pool.map(read, windows))
            for (off, n), (rc, out) in zip(windows, got):
                assert rc == 0
                np.testing.assert_array_equal(out, whole[off // 4:(off + n) // 4])
    finally:
        S.lib.srack_device_free(d_fr)
