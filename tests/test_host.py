"""Host-side logic of the product (no GPU): C-ABI surface, graph API, planner parity with the oracle, flattening."""
import ctypes
import random
import re
import os

import numpy as np
import pytest

import srack_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def S():
    return srack_pkg.load()


def test_library_exports_every_declared_symbol(S):
    hdr = open(os.path.join(ROOT, "include", "srack_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(srack_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(S.ABI_SYMBOLS)
    L = ctypes.CDLL(S.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert L.srack_abi_version() == 2


def test_module_defaults_match_reference_new(S):
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    q = S.Patch(44100, 64, 2)
    osc, vcf, adsr, vca, mix, math, out = [q.add_module(t) for t in (1, 2, 3, 4, 5, 6, 0)]
    assert [q.get_num_inputs(m) for m in (osc, vcf, adsr, vca, mix, math, out)] == [2, 2, 1, 2, 4, 2, 2]
    assert [q.get_num_outputs(m) for m in (osc, vcf, adsr, vca, mix, math, out)] == [3, 3, 1, 1, 1, 1, 0]
    assert q.get_field(osc, S.OSC_VAL) == 0.0 and q.get_field(osc, S.OSC_ANTIALIASING) == 1 and q.get_field(osc, S.OSC_SYNC_LAST) == 1
    assert q.get_field(vcf, S.VCF_FREQ) == float(np.float32(0.2)) and q.get_field(vcf, S.VCF_RES) == 0.5 and q.get_field(vcf, S.VCF_EXP_AMT) == 0.5
    assert [q.get_field(adsr, f) for f in (S.ADSR_A_SEC, S.ADSR_D_SEC, S.ADSR_S_VAL, S.ADSR_R_SEC)] == [0.0, 0.5, 0.25, 0.5]
    assert q.get_field(adsr, S.ADSR_MODE) == 4 and q.get_field(adsr, S.ADSR_SAMPLE_RATE) == 44100.0 and q.get_field(adsr, S.ADSR_GATE_LAST) == 1
    assert [q.get_field(mix, k) for k in range(4)] == [1.0] * 4
    assert q.get_field(math, S.MATH_CONSTANT) == 0.0 and q.get_field(math, S.MATH_OPERATION) == 0
    assert p.get_input(ids["vca"], 1) == (ids["adsr"], 0) and p.get_input(ids["osc_a"], 0) is None


def test_error_behaviour(S):
    p = S.Patch(48000, 64, 2)
    osc, out = p.add_module(1), p.add_module(0)
    with pytest.raises(S.SrackError) as e:  # Err(()) of set_input on a bad port
        p.connect(osc, 0, out, 2)
    assert e.value.code == S.ERR_PORT
    with pytest.raises(S.SrackError) as e:  # get_output(3) is Err(())
        p.connect(osc, 3, out, 0)
    assert e.value.code == S.ERR_PORT
    with pytest.raises(S.SrackError) as e:
        p.connect(7, 0, out, 0)
    assert e.value.code == S.ERR_INVALID
    with pytest.raises(S.SrackError) as e:  # the reference has thirteen module types (synth.rs:295-311)
        p.add_module(13)
    assert e.value.code == S.ERR_UNSUPPORTED
    for bad in ((0, 64, 2), (70000, 64, 2), (48000, 0, 2), (48000, 64, 0), (48000, 64, 9)):
        with pytest.raises(S.SrackError):
            S.Patch(*bad)
    q = S.Patch(48000, 64, 2)
    q.add_module(1)
    with pytest.raises(S.SrackError) as e:  # find_output() fails: no OutputModule
        q.plan()
    assert e.value.code == S.ERR_NO_OUTPUT
    # a module wired to itself deadlocks the reference; rejected at flatten time
    r = S.Patch(48000, 64, 2)
    mix, out = r.add_module(5), r.add_module(0)
    r.connect(mix, 0, mix, 0)
    r.connect(mix, 0, out, 0)
    r.configure_voices(4)
    with pytest.raises(S.SrackError) as e:
        r.info()
    assert e.value.code == S.ERR_SELF_LOOP
    # render before voices are configured
    s = S.Patch(48000, 64, 2)
    S.build_p1(s)
    with pytest.raises(S.SrackError) as e:
        s.render_raw(16)
    assert e.value.code == S.ERR_STATE


def _topo_graph(g):
    mods = [g.add_module(5) for _ in range(7)]
    out = g.add_module(0)
    free = {m: 0 for m in mods + [out]}

    def connect(src, sink):
        g.connect(src, 0, sink, free[sink])
        free[sink] += 1

    for a, b in ((0, 1), (1, 2), (2, 3)):
        connect(mods[a], mods[b])
    connect(mods[3], out)
    connect(mods[0], mods[4])
    connect(mods[4], mods[3])
    connect(mods[6], mods[4])
    connect(mods[5], mods[6])
    connect(mods[6], mods[5])
    return mods, out


def test_reference_topological_sort_on_the_product_planner(S, oracle):
    """synth::tests::topological_sort (synth.rs:537-613) against the product, and vs the oracle's order."""
    g = S.Patch(44100, 64, 2)
    mods, out = _topo_graph(g)
    o = oracle.OraclePatch(44100, 64, 2)
    _topo_graph(o)
    rng = random.Random(3)
    for _ in range(1000):
        lst = mods + [out]
        rng.shuffle(lst)
        plan = g.plan(output=out, all_modules=lst)
        idx = {m: i for i, m in enumerate(plan)}
        assert idx[mods[0]] < idx[mods[1]] < idx[mods[2]] < idx[mods[3]] < idx[out]
        assert idx[mods[0]] < idx[mods[4]] < idx[mods[3]]
        assert idx[mods[6]] < idx[mods[4]]
        assert idx[mods[5]] < idx[mods[6]]
        assert plan == o.plan(output=out, all_modules=lst)
        assert g.removed_edges() == o.removed_edges()


def test_planner_random_graphs_product_vs_oracle(S, oracle):
    rng = random.Random(11)
    n_in = {0: 2, 1: 2, 2: 2, 3: 1, 4: 2, 5: 4, 6: 2}
    n_out = {0: 0, 1: 3, 2: 3, 3: 1, 4: 1, 5: 1, 6: 1}
    for trial in range(300):
        n = rng.randint(2, 12)
        types = [rng.choice([1, 2, 3, 4, 5, 6]) for _ in range(n)]
        types.insert(rng.randint(0, n), 0)
        g, o = S.Patch(48000, 8, 2), oracle.OraclePatch(48000, 8, 2)
        for t in types:
            g.add_module(t)
            o.add_module(t)
        for sink, t in enumerate(types):
            for port in range(n_in[t]):
                if rng.random() < 0.6:
                    src = rng.randrange(len(types))
                    if src == sink or n_out[types[src]] == 0:
                        continue
                    sp = rng.randrange(n_out[types[src]])
                    g.connect(src, sp, sink, port)
                    o.connect(src, sp, sink, port)
        plan = g.plan()
        assert plan == o.plan(), trial
        assert g.removed_edges() == o.removed_edges(), trial
        pos = {m: i for i, m in enumerate(plan)}
        for (src, sp, sink, k) in g.delayed_edges():  # delayed <=> source scheduled after its sink
            assert pos[src] > pos[sink]
            assert g.get_input(sink, k) == (src, sp)


def test_flatten_descriptions(S):
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    p.configure_voices(100)
    det, cut = S.p1_voice_params(100)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    info = p.info()
    # uniform hoisting: LFO + ADSR carry no per-voice value => control program (3 ops: OSC, ADSR, track out);
    # the voice program keeps OSC_A, VCF, VCA, OUT (the VCA's CV is the control track, read in place) and matches the fused track kernel
    assert "voice[ops=4" in info and "fused=2" in info and "ctl[ops=3" in info and "tracks=1" in info
    assert p.planes() == (1, [0, 0])
    # one voice: nothing to share, the all-per-voice fused kernel
    q = S.Patch(48000, 1024, 2)
    S.build_p1(q)
    q.configure_voices(1)
    assert "voice[ops=6" in q.info() and "fused=1" in q.info() and "ctl[" not in q.info()
    # identical voices: everything is voice-invariant; the voice program only broadcasts the track
    q = S.Patch(48000, 1024, 2)
    S.build_p1(q)
    q.configure_voices(4096)
    # ... and a control program of five modules becomes five pipelined units, one module each
    assert "voice[ops=1" in q.info() and q.info().count("ctl[ops=2") == 5 and "tracks=5" in q.info()
    # feedback patch, per-voice beta: the whole loop is per voice; B = 1 => ring in LDS rows, tile of 1
    q = S.Patch(48000, 1, 2)
    ids2 = S.build_p2(q)
    q.configure_voices(64)
    q.set_voice_field(ids2["mul_fb"], S.MATH_CONSTANT, np.linspace(0.1, 0.4, 64))
    assert "rings=1(lds)" in q.info() and "tile=1" in q.info() and "ctl[" not in q.info()
    assert q.delayed_edges() == [(0, 0, 1, 0)]  # OSC_M.sine -> MUL_FB.in1
    q = S.Patch(48000, 1024, 2)
    ids2 = S.build_p2(q)
    q.configure_voices(64)
    q.set_voice_field(ids2["mul_fb"], S.MATH_CONSTANT, np.linspace(0.1, 0.4, 64))
    assert "rings=1(hbm)" in q.info() and "tile=32" in q.info()
    # per-voice values of a state field are returned before any render
    p.set_voice_field(ids["osc_a"], S.OSC_POS, np.linspace(0, 0.5, 100))
    np.testing.assert_array_equal(p.get_voice_field(ids["osc_a"], S.OSC_POS), np.linspace(0, 0.5, 100))
    np.testing.assert_array_equal(p.get_voice_field(ids["vcf"], S.VCF_FREQ), cut.astype(np.float64))
    # a module of the control program has one state shared by all voices
    np.testing.assert_array_equal(p.get_voice_field(ids["adsr"], S.ADSR_MODE), np.full(100, 4.0))


def test_no_gpu_render_fails_loudly(S):
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    p = S.Patch(48000, 64, 2)
    S.build_p1(p)
    p.configure_voices(64)
    with pytest.raises(S.SrackError) as e:
        p.render(16)
    assert e.value.code == S.ERR_DEVICE
    with pytest.raises(S.SrackError) as e:
        p.reserve(16)
    assert e.value.code == S.ERR_DEVICE


def test_sequencer_graph_api(S):
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p3(p)
    assert p.get_num_inputs(ids["grid"]) == 2 and p.get_num_outputs(ids["grid"]) == 3 and p.get_num_outputs(ids["pat"]) == 9
    assert p.get_field(ids["grid"], S.GRIDSEQ_STEPS_PER_OCTAVE) == 12 and p.get_field(ids["grid"], S.GRIDSEQ_STEP_LAST) == 1
    assert p.get_step(ids["grid"], 0, 1) == (S.STEP_HOLD, 3) and p.get_step(ids["grid"], 0, 3) == (S.STEP_NONE, 0)
    assert p.get_step(ids["pat"], 5, 2) == (S.STEP_HOLD, 0) and p.get_step(ids["pat"], 1, 0) == (S.STEP_ON, 0)
    with pytest.raises(S.SrackError):
        p.set_step(ids["grid"], 1, 0, 1, 0)      # a grid sequencer has one channel
    with pytest.raises(S.SrackError):
        p.set_step(ids["osc"], 0, 0, 1, 0)       # not a sequencer
    with pytest.raises(S.SrackError):
        p.set_field(ids["pat"], S.PATSEQ_LENGTH, 0)
    with pytest.raises(S.SrackError) as e:       # Err(()) of get_output(9)
        p.connect(ids["pat"], 9, ids["out"], 0)
    assert e.value.code == S.ERR_PORT
    # per-voice transpose only: clock, both sequencers and both envelopes are voice-invariant => control program
    p.configure_voices(128)
    p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, np.linspace(-2, 0, 128))
    info = p.info()
    # ... as five pipelined units (clock, grid, pattern, two envelopes); 4 tracks reach the voice program, 4 more carry
    # wires between units
    assert info.count("ctl[ops=") == 5 and "tracks=8" in info, info
    assert "fused=5" in info.split(" + ")[0]     # [transpose] -> VCO -> VCF(cv) -> VCA + a raw gate channel: the fused sequencer-driven chain


def test_sample_and_nonlinear_graph_api(S):
    p = S.Patch(44100, 256, 2)
    ids = S.build_p4(p)
    smp, sh = ids["smp"], ids["shaper"]
    assert (p.get_num_inputs(smp), p.get_num_outputs(smp), p.get_num_inputs(sh), p.get_num_outputs(sh)) == (2, 1, 2, 1)
    assert p.get_field(smp, S.SAMPLE_SAMPLE_RATE) == 44100.0 and p.get_field(smp, S.SAMPLE_WAVE_SAMPLE_RATE) == 44100.0
    assert p.get_field(smp, S.SAMPLE_WAVE_NEW) == 1 and p.get_field(smp, S.SAMPLE_GATE_LAST) == 1 and p.get_field(smp, S.SAMPLE_PLAYING) == 0
    w, sr = p.get_wave(smp)
    np.testing.assert_array_equal(w, S.p4_wave())
    q = S.Patch(48000, 64, 2)
    nl = q.add_module(S.MOD_NONLINEAR)
    fresh = q.add_module(S.MOD_SAMPLE)
    assert q.get_field(nl, S.NONLIN_CONSTANT) == 1.0                   # math.rs:194
    assert q.get_wave(fresh)[0].size == 0 and q.get_field(fresh, S.SAMPLE_WAVE_NEW) == 0   # WaveBox::default()
    with pytest.raises(S.SrackError):
        q.set_wave(nl, np.zeros(4, dtype=np.float32), 48000.0)         # not a SampleModule
    with pytest.raises(S.SrackError) as e:
        q.connect(nl, 1, fresh, 0)                                     # get_output(1) is Err(())
    assert e.value.code == S.ERR_PORT
    # per-voice exponent + depth: the clock and the LFO stay voice-invariant (control program), the sampler does not
    p.configure_voices(256)
    depth, expo = S.p4_voice_params(256)
    p.set_voice_field(ids["depth"], S.MATH_CONSTANT, depth)
    p.set_voice_field(sh, S.NONLIN_CONSTANT, expo)
    info = p.info()
    assert "ctl[ops=" in info and "tracks=2" in info, info
    with pytest.raises(S.SrackError) as e:
        p.set_voice_field(smp, S.SAMPLE_WAVE_NEW, np.zeros(256))
        p.info()
    assert e.value.code == S.ERR_UNSUPPORTED


def test_noise_module_graph_api(S):
    """oscillator.rs:308-393: no inputs, one output, no parameters; usable as a source like any other module."""
    p = S.Patch(48000, 64, 2)
    nz, vcf, out = p.add_module(S.MOD_NOISE), p.add_module(S.MOD_MOOG_FILTER), p.add_module(S.MOD_OUTPUT)
    assert (p.get_num_inputs(nz), p.get_num_outputs(nz)) == (0, 1)
    with pytest.raises(S.SrackError) as e:
        p.connect(vcf, 0, nz, 0)                 # set_input is Err(())
    assert e.value.code == S.ERR_PORT
    with pytest.raises(S.SrackError) as e:
        p.connect(nz, 1, vcf, 0)                 # get_output(1) is Err(())
    assert e.value.code == S.ERR_PORT
    with pytest.raises(S.SrackError):
        p.set_field(nz, 0, 1.0)                  # no fields
    p.connect(nz, 0, vcf, 0)
    p.connect(vcf, 0, out, 0)
    p.connect(nz, 0, out, 1)
    assert p.plan() == [nz, vcf, out]
    p.set_noise_seed(1234, first_voice=5)
    p.configure_voices(128)
    info = p.info()                              # every voice draws its own noise: nothing is hoisted to the control program
    assert "ctl[" not in info and "ops=4" in info, info


def test_mix_tree_lane_map_is_documented_formula():
    """(kept from the register-tree mix-down experiment, DESIGN section 4: the lane -> sample map of the reduction tree)"""
    L = np.arange(64)
    row, bank, hi = L >> 4, (L >> 2) & 3, (L >> 1) & 1
    slot = 16 * hi + 8 * (bank & 1) + 4 * (bank >> 1) + 2 * (row & 1) + (row >> 1)
    assert sorted(set(slot.tolist())) == list(range(32)) and (np.bincount(slot) == 2).all()


def test_freeverb_module_graph_api(S):
    """freeverb.rs:60-82, 135-206: two inputs (Left, Right), two outputs, the *_ctl defaults; parameters are f64."""
    p = S.Patch(48000, 64, 2)
    osc, fv, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_FREEVERB), p.add_module(S.MOD_OUTPUT)
    assert (p.get_num_inputs(fv), p.get_num_outputs(fv)) == (2, 2)
    assert [p.get_field(fv, f) for f in range(6)] == [0.5, 0.0, 1.0, 0.5, 0.5, 0.0]
    p.set_field(fv, S.FREEVERB_ROOM_SIZE, 0.1 + 0.2)                 # not representable in f32: kept as the f64 it is
    assert p.get_field(fv, S.FREEVERB_ROOM_SIZE) == 0.1 + 0.2
    for bad in ((osc, 0, fv, 2), (fv, 2, out, 0)):
        with pytest.raises(S.SrackError) as e:
            p.connect(*bad)
        assert e.value.code == S.ERR_PORT
    p.connect(osc, 2, fv, 0)
    p.connect(fv, 0, out, 0)
    p.connect(fv, 1, out, 1)
    assert p.plan() == [osc, fv, out]
    p.configure_voices(64)
    assert "ops=" in p.info()
    with pytest.raises(S.SrackError) as e:                             # the reverb's parameters are voice-invariant here
        p.set_voice_field(fv, S.FREEVERB_WET, np.linspace(0, 1, 64))
        p.info()
    assert e.value.code == S.ERR_UNSUPPORTED
    q = S.Patch(700, 64, 2)                                            # 225 * 700 / 44100 = 3 samples: below the supported line length
    a, b = q.add_module(S.MOD_FREEVERB), q.add_module(S.MOD_OUTPUT)
    q.connect(a, 0, b, 0)
    q.configure_voices(1)
    with pytest.raises(S.SrackError) as e:
        q.info()
    assert e.value.code == S.ERR_UNSUPPORTED


def test_save_srk_rejects_a_short_buffer(S):
    """A truncated MessagePack file can still parse up to the cut: the call fails instead of handing one back."""
    import ctypes as C
    p = S.Patch(48000, 64, 2)
    S.build_p1(p)
    n = C.c_size_t()
    assert S.lib.srack_patch_save_srk(p.h, None, 0, C.byref(n)) == S.OK and n.value > 100
    buf = C.create_string_buffer(n.value)
    need = n.value
    assert S.lib.srack_patch_save_srk(p.h, buf, need - 1, C.byref(n)) == S.ERR_INVALID
    assert n.value == need and b"too small" in S.lib.srack_last_error()
    assert S.lib.srack_patch_save_srk(p.h, buf, need, C.byref(n)) == S.OK
    assert buf.raw == p.save_srk()


def test_hostile_numbers_and_lists_do_not_crash(S):
    """NaN / huge values in integer-like fields, a negative list length: clamped or rejected, never undefined behaviour."""
    import ctypes as C
    p = S.Patch(48000, 64, 2)
    ids = S.build_p1(p)
    for bad in (float("nan"), 1e300, -1e300, float("inf")):
        p.set_field(ids["adsr"], S.ADSR_MODE, bad)
        assert p.get_field(ids["adsr"], S.ADSR_MODE) in (0.0, 2147483647.0, -2147483648.0)
    p.configure_voices(3)
    p.set_voice_field(ids["adsr"], S.ADSR_MODE, np.array([np.nan, 1e30, -1e30]))
    arr = (C.c_int * 2)(0, 1)
    out = (C.c_int * 8)()
    assert S.lib.srack_patch_plan_list(p.h, ids["out"], arr, -5, out, 8) == S.ERR_INVALID


# ---- the kernel generator of the general path (jit.cpp), as far as it goes without a GPU: source + compilation for gfx950 ----------
def _have_hiprtc():
    import ctypes
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            ctypes.CDLL(name)
            return True
        except OSError:
            pass
    return False


def test_specialised_kernel_source_of_p1_and_p3(S):
    p = S.Patch(48000, 1024, 2)
    ids = S.build_p1(p)
    p.configure_voices(128)
    det, cut = S.p1_voice_params(128)
    p.set_voice_field(ids["osc_a"], S.OSC_VAL, det)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut)
    src = p.kernel_source(S.RENDER_NO_FUSION)
    assert 'extern "C" __global__' in src and "srk_voice(KernelArgs a)" in src
    assert "fosc_saw(m0)" in src and "vcf_run<true>" in src and "emit_put<KOUT>" in src  # the carried-phase saw (fixed-point phase: nothing integrates it), the default-mode ladder
    assert "srk_ctl0" in src and "a.ctl_slots[blockIdx.x]" in src                               # the gate -> envelope unit rides along,
    assert "cosc_tile<false>(" in src and "adsr_seg_tile(" in src and "sample(lane);" in src     # evaluated across lanes: lane j = sample j of the tile
    assert "rowf(14)" in src and "a.ops[1].par_val[2]" in src                               # per-voice cutoff from its row, uniform exp_amt from the op list
    exact = p.kernel_source(S.RENDER_NO_FUSION | S.RENDER_EXACT_OSC)
    # exact mode: the tile-wise saw with the literal per-sample oscillator behind it, the literal ladder without a look at its (bounded)
    # input, the sample loop versioned on both preconditions; the gate LFO of the control unit leaves its PolyBLEP windows to osc_step
    assert "xsaw_tile(" in exact and "osc_step(" in exact and "vcf_run_bounded(" in exact and "if (m0_xs && m1_fin) {" in exact
    assert "cosc_tile<true>(" in exact and "adsr_seg_tile(" in exact                            # (the exact flavour of the same unit)
    # parameter VALUES are not part of the kernel: an edit does not ask for another compilation
    p.set_field(ids["vcf"], S.VCF_RES, 0.7)
    p.set_voice_field(ids["vcf"], S.VCF_FREQ, cut[::-1].copy())
    body = lambda text: text.split("\n", 1)[1]   # (the first line names the program, with the error bound of its parameter values: a comment for the reader)
    assert body(p.kernel_source(S.RENDER_NO_FUSION)) == body(src)
    q = S.Patch(48000, 1024, 2)
    qi = S.build_p3(q)
    q.configure_voices(70)
    q.set_voice_field(qi["transpose"], S.MATH_CONSTANT, np.linspace(-1, 0, 70).astype(np.float32))
    src3 = q.kernel_source(0)
    assert src3.count("static __device__ __forceinline__ void srk_ctl") == 5 and "steposc_step<0x20u>" in src3 and "emit_track_put" in src3
    assert "seq_advance" in src3 and "readlane_f32(trk" in src3
    # track values reach the sample function as arguments, fetched a group of samples ahead; a swept cutoff compares the frequency only
    assert "auto sample = [&](int i, float tk0, float tk1, float tk2, float tk3)" in src3 and "tg0[u] = trk0[t0 + (uint32_t)(i0 + u)]" in src3
    assert "vcf_res_settle(" in src3 and "vcf_coeffs_freq<true>" in src3 and "vcf_frequency_med3(" in src3 and "vca_step_uniform(" in src3
    exact3 = q.kernel_source(S.RENDER_EXACT_OSC)
    # exact mode: the sequenced saw tile-wise wherever its note track is flat over the tile, the literal ladder told so by a scalar;
    # the clock LFO's square leaves only its PolyBLEP windows to the f64 formulas
    assert "track_flat(trk0_v + t0, lane)" in exact3 and "xsaw_tile(m1_x, m1_xt, kMixPitch, kMixRows)" in exact3
    assert "vcf_run_bounded(m2, m2_fin, m1_flat" in exact3 and "vcf_coeffs_freq<false>" in exact3 and "vcf_frequency(" in exact3
    assert "cosc_tile<true>(" in exact3                                                         # the clock: a unit evaluated across lanes
    r = S.Patch(48000, 64, 2)   # the one module the generator leaves to the interpreter
    v, o = r.add_module(S.MOD_FREEVERB), r.add_module(S.MOD_OUTPUT)
    r.connect(v, 0, o, 0)
    r.configure_voices(4)
    with pytest.raises(S.SrackError) as e:
        r.kernel_source(S.RENDER_NO_UNIFORM_HOIST)
    assert e.value.code == S.ERR_UNSUPPORTED


def test_specialised_kernel_versions_oscillators_with_a_bounded_cv(S):
    """What render_fm_pair proves by hand the generator derives for any program: an oscillator whose pitch CV is a bounded value — a sine
    through a gain, here P2's two operators — gets the per-wave vote and a copy of the sample loop per class (jit.cpp, analyze)."""
    p = S.Patch(48000, 1, 2)
    ids = S.build_p2(p)
    p.configure_voices(128)
    beta, index = S.p2_voice_params(128)
    p.set_voice_field(ids["mul_fb"], S.MATH_CONSTANT, beta)
    p.set_voice_field(ids["mul_idx"], S.MATH_CONSTANT, index)
    # (the patch's own feedback loop runs through a pitch: since round 5 its default is the exact flavour — approx.cpp — and the fast kernels,
    # whose per-wave proofs these are, are what SRACK_RENDER_KEEP_DEFAULT renders)
    assert "; exact osc 0]" in p.kernel_source(S.RENDER_NO_FUSION).split("\n", 1)[0]   # (the modulator: exact as a whole; the carrier keeps the default forms)
    src = p.kernel_source(S.RENDER_NO_FUSION | S.RENDER_KEEP_DEFAULT)
    assert "approx[kept default: unbounded gain" in src.split("\n", 1)[0]
    # both operators: |cv| <= 1 x |gain|; the modulator's bound rests on the z^-1 ring's stored value being a sine's too
    assert "const float fm_b0 = (1.0f * __builtin_fabsf(m1_c));" in src and "const float fm_b1 = (1.0f * __builtin_fabsf(m4_c));" in src
    assert "fm_lane = fm_lane && __builtin_fabsf(ring6) <= 1.0f;" in src and "m2.pos >= 0.0 && m2.pos < 1.0" in src and "m5.pos >= 0.0 && m5.pos < 1.0" in src
    assert "auto fm_run = [&](auto fm_c0, auto fm_c1) {" in src and "| kFm0), m2, m2_k" in src and "| kFm1), m5, m5_k" in src
    assert src.count("fm_run(UC<") == 10  # nothing proved + 3 x 3 classes
    assert "m2_k.scale = (440.0 / m2_k.sr) * exp2(m2_k.val);" in src
    assert "osc_step((0x10du | kFm0)" in src and "osc_step((0x110du | kFm1)" in src  # OSC_CV_AUDIO_RATE: no compare with the last CV; the carrier's loose sine
    # exact mode proves nothing (bit-identical arithmetic), but an audio-rate CV is still not compared with the previous one
    exact = p.kernel_source(S.RENDER_NO_FUSION | S.RENDER_EXACT_OSC)
    assert "fm_run" not in exact and "osc_step(0x14du, m2" in exact
    # the app's block size: the ring lives in HBM, whose contents no launch-time vote can vouch for — the modulator keeps the literal forms,
    # and with it the carrier (a sine is only bounded by 1 while its phase provably stays in [0, 1))
    q = S.Patch(48000, 1024, 2)
    qi = S.build_p2(q)
    q.configure_voices(128)
    q.set_voice_field(qi["mul_fb"], S.MATH_CONSTANT, beta)
    q.set_voice_field(qi["mul_idx"], S.MATH_CONSTANT, index)
    ring = q.kernel_source(S.RENDER_NO_FUSION | S.RENDER_KEEP_DEFAULT)
    assert "fm_run" not in ring and "osc_step(0x10du, m2" in ring and "osc_step(0x110du, m5" in ring
    # an envelope on the index: the bound is the envelope's hull times the gains, its time constants join the vote
    e = S.Patch(48000, 1024, 2)
    lfo, env, om, vca, idx, oc, out = (e.add_module(t) for t in (S.MOD_OSCILLATOR, S.MOD_ADSR, S.MOD_OSCILLATOR, S.MOD_VCA, S.MOD_MATH, S.MOD_OSCILLATOR, S.MOD_OUTPUT))
    e.set_field(idx, S.MATH_OPERATION, 2)
    for a_, ap, b_, bp in ((lfo, 1, env, 0), (om, 0, vca, 0), (env, 0, vca, 1), (vca, 0, idx, 0), (idx, 0, oc, 0), (oc, 0, out, 0)):
        e.connect(a_, ap, b_, bp)
    e.configure_voices(64)
    for m_, f_ in ((idx, S.MATH_CONSTANT), (env, S.ADSR_S_VAL), (om, S.OSC_VAL)):
        e.set_voice_field(m_, f_, np.linspace(0.1, 0.9, 64).astype(np.float32))
    esrc = e.kernel_source(S.RENDER_NO_FUSION)
    assert "adsr_bound(m0, m0_k)) * __builtin_fabsf(m3_c));" in esrc and "fm_lane = fm_lane && adsr_tame(m0, m0_k);" in esrc and esrc.count("fm_run(UC<") == 4
    # a sequencer's note is stepwise and of any size: no vote, the carried-phase oscillator as before
    r = S.Patch(48000, 1024, 2)
    ri = S.build_p3(r)
    r.configure_voices(70)
    r.set_voice_field(ri["transpose"], S.MATH_CONSTANT, np.linspace(-1, 0, 70).astype(np.float32))
    assert "fm_run" not in r.kernel_source(0)


@pytest.mark.skipif(not _have_hiprtc(), reason="no libhiprtc on this host")
def test_specialised_kernels_compile_for_gfx950(S):
    """hiprtc cross-compiles without a GPU: the generated source of the BASELINE patches and of a few random ones (rings of every
    size class, sequencers and a sample player per voice) must at least be valid HIP for the library's target."""
    from tests.fuzz_patches import random_patch
    cases = []
    for B in (1, 7, 24, 1024):
        p = S.Patch(48000, B, 2)
        S.build_p2(p)
        p.configure_voices(64)
        cases.append((p, S.RENDER_NO_FUSION))
    p = S.Patch(48000, 1024, 2)
    S.build_p4(p)
    p.configure_voices(64)
    cases.append((p, S.RENDER_NO_UNIFORM_HOIST))
    for seed in (0, 3, 9):
        B, build, overrides = random_patch(seed)
        for flags in (3, 7):
            p = S.Patch(48000, B, 2)
            ids = build(p)
            p.configure_voices(70)
            for m, f, fn in overrides:
                p.set_voice_field(ids[m], f, fn(70))
            cases.append((p, flags))
    n = 0
    for p, flags in cases:
        try:
            p.kernel_compile(flags)
            n += 1
        except S.SrackError as e:
            assert e.code == S.ERR_UNSUPPORTED, str(e)[:2000]
    assert n >= 9


# ---- the cache of specialised kernels (jit.cpp): memory (bounded) over disk (persistent) over hiprtc -----------------------------------
_CACHE_PROBE = r"""
import json, os, sys
sys.path.insert(0, os.environ["SRACK_ROOT"])
import numpy as np
import srack_pkg
S = srack_pkg.load()
def chain(ops):   # oscillator -> one Math module per entry of `ops` -> output: the op sequence IS the program's structure
    p = S.Patch(48000, 64, 2)
    o = p.add_module(S.MOD_OSCILLATOR)
    prev = o
    for k in ops:
        m = p.add_module(S.MOD_MATH)
        p.set_field(m, S.MATH_OPERATION, k)
        p.connect(prev, 2 if prev == o else 0, m, 0)
        prev = m
    out = p.add_module(S.MOD_OUTPUT)
    p.connect(prev, 2 if prev == o else 0, out, 0)
    p.configure_voices(64)
    p.set_voice_field(o, S.OSC_VAL, np.linspace(-1, 1, 64).astype(np.float32))
    return p
for spec in json.loads(os.environ["SRACK_PROBE_SPECS"]):
    chain(spec).kernel_compile(S.RENDER_NO_FUSION)
print(json.dumps(S.kernel_cache_stats()))
"""


def _cache_probe(specs, cache_dir, extra_env=None, wait=True):
    import json, subprocess, sys
    env = dict(os.environ, SRACK_ROOT=ROOT, SRACK_PROBE_SPECS=json.dumps(specs), SRACK_KERNEL_CACHE_DIR=str(cache_dir), **(extra_env or {}))
    pr = subprocess.Popen([sys.executable, "-c", _CACHE_PROBE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if not wait:
        return pr
    out, err = pr.communicate(timeout=600)
    assert pr.returncode == 0, err[-3000:]
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.skipif(not _have_hiprtc(), reason="no libhiprtc on this host")
def test_kernel_cache_persists_across_processes(tmp_path):
    """The second start of a host finds its kernels on disk: no hiprtc call.  The key is the program's structure — parameter values
    and the voice count do not enter —, the architecture, the hiprtc version and the device headers this library embeds."""
    first = _cache_probe([[0], [1, 2]], tmp_path)
    assert first["compiled"] == 2 and first["disk_hits"] == 0 and first["directory"] == str(tmp_path) and first["compile_ms"] > 0
    assert len([f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]) == 2 and not [f for f in os.listdir(tmp_path) if f.endswith(".lock")]
    second = _cache_probe([[0], [1, 2], [0]], tmp_path)
    assert second["compiled"] == 0 and second["disk_hits"] == 2 and second["memory_hits"] == 1
    # a damaged file is not trusted: recompiled and rewritten
    victim = sorted(f for f in os.listdir(tmp_path) if f.endswith(".hsaco"))[0]
    with open(os.path.join(tmp_path, victim), "r+b") as f:
        f.seek(200)
        f.write(b"\x00garbage\x00")
    third = _cache_probe([[0], [1, 2]], tmp_path)
    assert third["compiled"] == 1 and third["disk_hits"] == 1
    off = _cache_probe([[0]], "off")
    assert off["compiled"] == 1 and off["directory"] == ""


@pytest.mark.skipif(not _have_hiprtc(), reason="no libhiprtc on this host")
def test_kernel_cache_is_bounded_in_memory(tmp_path):
    """A long-running host that keeps re-patching: the process keeps the last SRACK_KERNEL_CACHE_MAX structures, not all of them."""
    specs = [[a, b] for a in range(3) for b in range(3)] + [[0, 0]]   # nine structures, then the first again
    st = _cache_probe(specs, "off", {"SRACK_KERNEL_CACHE_MAX": "4"})
    assert st["resident_code_objects"] == 4 and st["code_evictions"] == 6 and st["compiled"] == 10  # [0, 0] had been evicted: compiled again
    st = _cache_probe(specs, tmp_path, {"SRACK_KERNEL_CACHE_MAX": "4"})
    assert st["resident_code_objects"] == 4 and st["compiled"] == 9 and st["disk_hits"] == 1          # ... or fetched from the disk level


@pytest.mark.skipif(not _have_hiprtc(), reason="no libhiprtc on this host")
def test_ranks_starting_together_compile_once(tmp_path):
    """Every rank of a multi-GPU job needs the same kernel at the same moment: one compiles (a per-kernel file lock), the others read."""
    import json
    procs = [_cache_probe([[2, 1, 0, 2]], tmp_path, wait=False) for _ in range(3)]
    stats = []
    for pr in procs:
        out, err = pr.communicate(timeout=600)
        assert pr.returncode == 0, err[-3000:]
        stats.append(json.loads(out.strip().splitlines()[-1]))
    assert sum(s["compiled"] for s in stats) == 1 and sum(s["disk_hits"] for s in stats) == 2


def test_nonlinear_takes_the_f32_power_only_where_nothing_integrates_it(S):
    """NONLIN_LOOSE (program.hpp): the flattener proves that a NonLinear module's output can reach neither a pitch input nor a threshold —
    P4's shaper feeds the OutputModule only — before the default mode's power goes through v_log_f32 / v_exp_f32; a shaper in front of an
    oscillator's CV, or of an envelope's gate, keeps the f64 power, and the exact mode never takes it."""
    import re

    def nonlin_flags(p, flags=0):
        src = p.kernel_source(flags | S.RENDER_NO_UNIFORM_HOIST)
        return [int(x, 16) for x in re.findall(r"nonlin_step\((0x[0-9a-f]+)u", src)]

    p = S.Patch(48000, 1024, 2)
    S.build_p4(p)
    p.configure_voices(8)
    assert [f & 0x200 for f in nonlin_flags(p)] == [0x200]
    assert [f & 0x300 for f in nonlin_flags(p, S.RENDER_EXACT_OSC)] == [0x100]            # NONLIN_EXACT
    for sink_type, sink_port in ((S.MOD_OSCILLATOR, 0), (S.MOD_ADSR, 0)):                  # a pitch CV; a gate (a threshold)
        q = S.Patch(48000, 1024, 2)
        osc, shaper, sink, out = q.add_module(S.MOD_OSCILLATOR), q.add_module(S.MOD_NONLINEAR), q.add_module(sink_type), q.add_module(S.MOD_OUTPUT)
        q.connect(osc, S.OSC_OUT_SINE, shaper, 0)
        q.connect(shaper, 0, sink, sink_port)
        q.connect(sink, 0, out, 0)
        q.connect(shaper, 0, out, 1)
        q.configure_voices(8)
        assert [f & 0x200 for f in nonlin_flags(q)] == [0], sink_type
    # ... nor on a feedback cycle (shaper -> mixer -> back into the shaper), nor in front of another shaper's base
    q = S.Patch(48000, 1024, 2)
    osc, mix, shaper, out = q.add_module(S.MOD_OSCILLATOR), q.add_module(S.MOD_MONO_MIXER), q.add_module(S.MOD_NONLINEAR), q.add_module(S.MOD_OUTPUT)
    q.connect(osc, S.OSC_OUT_SINE, mix, 0)
    q.connect(shaper, 0, mix, 1)
    q.connect(mix, 0, shaper, 0)
    q.connect(shaper, 0, out, 0)
    q.configure_voices(8)
    assert [f & 0x200 for f in nonlin_flags(q)] == [0]
    q = S.Patch(48000, 1024, 2)
    osc, first, second, out = q.add_module(S.MOD_OSCILLATOR), q.add_module(S.MOD_NONLINEAR), q.add_module(S.MOD_NONLINEAR), q.add_module(S.MOD_OUTPUT)
    q.connect(osc, S.OSC_OUT_SINE, first, 0)
    q.connect(first, 0, second, 0)
    q.connect(second, 0, out, 0)
    q.configure_voices(8)
    assert sorted(f & 0x200 for f in nonlin_flags(q)) == [0, 0x200]   # the first feeds a base: f64; the second feeds the output only: f32


# ---- the Rust binding's extern block against the C header, mechanically (scope row (f)3: no rustc in the image) -----------------------
def _c_prototypes():
    """name -> (return type, [parameter types]) of every function include/srack_hip.h declares, comments stripped"""
    import re
    text = open(os.path.join(ROOT, "include", "srack_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for ret, name, args in re.findall(r"\b(int|const char\s*\*|void)\s+(srack_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",")]:
            if a in ("void", ""):
                continue
            m = re.match(r"(.*?)(\w+)$", a)          # the last word is the parameter's name
            params.append(re.sub(r"\s+", " ", m.group(1)).strip().replace(" *", "*"))
        protos[name] = (re.sub(r"\s+", " ", ret).replace(" *", "*"), params)
    return protos


_C_TO_RUST = {
    "int": {"c_int"}, "uint32_t": {"u32"}, "uint64_t": {"u64"}, "size_t": {"usize"}, "double": {"f64"}, "float": {"f32"},
    "const char*": {"*const c_char"}, "char*": {"*mut c_char"}, "void*": {"*mut c_void", "*mut u8"}, "const void*": {"*const c_void", "*const u8"},
    "void**": {"*mut *mut c_void"}, "srack_patch*": {"*mut SrackPatch"}, "const srack_patch*": {"*const SrackPatch", "*mut SrackPatch"},
    "srack_patch**": {"*mut *mut SrackPatch"}, "float*": {"*mut f32"}, "const float*": {"*const f32"}, "double*": {"*mut f64"},
    "const double*": {"*const f64"}, "int*": {"*mut c_int"}, "const int*": {"*const c_int"}, "size_t*": {"*mut usize"},
    "srack_kernel_cache_info*": {"*mut SrackKernelCacheInfo"},
}


def _rust_externs(text):
    """[(name, [parameter types], return type)] of the `fn srack_*` declarations inside `extern "C" { ... }` blocks"""
    import re
    out = []
    for block in re.findall(r'extern "C" \{(.*?)\n\s*\}', text, flags=re.S):
        block = re.sub(r"//[^\n]*", "", block)
        for name, args, ret in re.findall(r"fn\s+(srack_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
            params = [re.sub(r"\s+", " ", a.split(":", 1)[1]).strip() for a in args.split(",") if ":" in a]
            out.append((name, params, (ret or "()").strip()))
    return out


@pytest.mark.parametrize("source", ["integration/rust/src/lib.rs", "INTEGRATION.md"])
def test_rust_extern_block_matches_the_header(source):
    """The Rust side cannot be compiled here (no rustc): at least every `extern "C"` declaration — the binding crate's and the glue shown
    in INTEGRATION.md — must name a function the header declares, with the same number of parameters and types that are the C ones'
    Rust spellings (c_int for int, *mut SrackPatch for srack_patch*, usize for size_t ...).  A signature that drifts is a crash at the
    first call; this is the check a `bindgen` run would be."""
    protos = _c_prototypes()
    assert len(protos) >= 56 and "srack_render" in protos
    decls = _rust_externs(open(os.path.join(ROOT, source)).read())
    assert len(decls) >= 15, source
    for name, params, ret in decls:
        assert name in protos, f"{source}: {name} is not in include/srack_hip.h"
        c_ret, c_params = protos[name]
        assert ret in _C_TO_RUST[c_ret], f"{source}: {name} returns {ret}, the header says {c_ret}"
        assert len(params) == len(c_params), f"{source}: {name} takes {len(params)} parameters, the header {len(c_params)}"
        for k, (r, c) in enumerate(zip(params, c_params)):
            assert r in _C_TO_RUST[c], f"{source}: {name} parameter {k} is {r}, the header says {c}"


# ---- round 4's last flattener rules, read off the generated source (no GPU) -----------------------------------------------------------
def test_pitch_cv_that_holds_and_pitch_cv_that_sweeps(S):
    """OSC_CV_AUDIO_RATE (program.hpp) is the flattener's: a pitch CV with an oscillator, a filter, noise or a sample player upstream sweeps
    (the polynomial 2^cv every sample); one with nothing upstream but envelopes, sequencers and arithmetic on them holds, and its oscillator
    recomputes the increment per held value with the reference's own 2^cv (modules.hip.h, osc_delta_cold; flatten.cpp, `sweeps`)."""
    import re

    def osc_flags(p):
        src = p.kernel_source(S.RENDER_NO_UNIFORM_HOIST | S.RENDER_NO_FUSION)
        return [int(x, 16) for x in re.findall(r"osc_step\(\(?(0x[0-9a-f]+)u", src)]

    AUDIO, HAS_CV = 0x100, 0x1
    for upstream, expect in ((S.MOD_ADSR, 0), (S.MOD_OSCILLATOR, AUDIO), (S.MOD_MOOG_FILTER, AUDIO)):
        p = S.Patch(48000, 64, 2)
        src_m, gain, osc, out = p.add_module(upstream), p.add_module(S.MOD_MATH), p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_OUTPUT)
        p.set_field(gain, S.MATH_OPERATION, S.MATH_MULTIPLY)
        p.set_field(gain, S.MATH_CONSTANT, 0.5)
        p.connect(src_m, 0, gain, 0)
        p.connect(gain, 0, osc, 0)
        p.connect(osc, S.OSC_OUT_SINE, out, 0)
        p.configure_voices(8)
        with_cv = [f for f in osc_flags(p) if f & HAS_CV]
        assert with_cv and all((f & AUDIO) == expect for f in with_cv), (upstream, [hex(f) for f in with_cv])


def test_cutoff_rules_of_the_flattener(S):
    """csrc/approx.cpp through the flattener: a cutoff that moves at audio rate (a square, noise) leaves its filter without a bound — the
    literal ladder, and the exact PolyBLEP for the square and for the saw on the audio input; an envelope on the cutoff — P3's sweep — changes
    nothing."""
    import re
    EXACT_BLEP = 1 << 13
    for cv_source in (S.MOD_OSCILLATOR, S.MOD_NOISE, S.MOD_ADSR):
        p = S.Patch(48000, 64, 2)
        audio, cv, vcf, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(cv_source), p.add_module(S.MOD_MOOG_FILTER), p.add_module(S.MOD_OUTPUT)
        p.connect(audio, S.OSC_OUT_SAW, vcf, 0)
        p.connect(cv, S.OSC_OUT_SQUARE if cv_source == S.MOD_OSCILLATOR else 0, vcf, 1)
        p.connect(vcf, 0, out, 0)
        p.configure_voices(8)
        src = p.kernel_source(S.RENDER_NO_UNIFORM_HOIST | S.RENDER_NO_FUSION)
        literal_forms = [int(x, 16) for x in re.findall(r"osc_step\(\(?(0x[0-9a-f]+)u", src)]   # (the fast constant-pitch forms have other names)
        assert ("vcf_run<true>" in src) == (cv_source == S.MOD_ADSR), cv_source
        if cv_source == S.MOD_ADSR:
            assert "fosc_saw" in src and not literal_forms
        else:
            assert "fosc_saw" not in src and literal_forms and all(f & EXACT_BLEP for f in literal_forms)


def test_exp2_fast10_coefficients_in_the_header():
    """modules.hip.h's polynomial 2^f for sweeping pitch CVs (exp2_fast10), read out of the header and evaluated in f64 by its own scheme against
    mpmath: 5e-16 over |f| <= 1/2 (tools/exp2_coeffs.py made the coefficients; the degree-8 form it replaced outside the proved FM loops
    was 1.1e-12, a phase drift wherever such a CV sat still — tests/test_gpu_fuzz.py, the one-second cases)."""
    import re
    mp = pytest.importorskip("mpmath")
    text = open(os.path.join(ROOT, "s-rack_amd", "csrc", "modules.hip.h")).read()
    body = text[text.index("SRK_DEV double exp2_fast10(double x)"):]
    body = body[:body.index("return kReduce")]
    hx = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p[+-]\d+", body)]
    assert len(hx) == 10   # c1, c3, c2, c5, c4, c7, c6, c9, c8, c10 in the order the fmas name them; c0 = 1.0
    c1, c3, c2, c5, c4, c7, c6, c9, c8, c10 = hx
    f = np.linspace(-0.5, 0.5, 4001)
    f2 = f * f
    a01, a23, a45, a67, a89 = c1 * f + 1.0, c3 * f + c2, c5 * f + c4, c7 * f + c6, c9 * f + c8
    f4 = f2 * f2
    b0, b1, b2 = a23 * f2 + a01, a67 * f2 + a45, c10 * f2 + a89
    p = b2 * (f4 * f4) + (b1 * f4 + b0)
    mp.mp.dps = 40
    worst = max(abs(mp.mpf(float(pi)) / mp.power(2, mp.mpf(float(fi))) - 1) for pi, fi in zip(p, f))
    assert worst < 6e-16, float(worst)


def test_exp2_fast9_coefficients_in_the_header():
    """The degree-9 form the proved classes take (exp2_fast9, OSC_CV_SERIES9), the same way: 2e-14 over |f| <= 1/2 — and what it is there for:
    the error of 2^cv = (2^(cv / 4))^4 summed over a minute of a sine through config 4's index, weighted by the increment — the phase the
    carrier drifts by — is below 5e-10 cycles (degree 8: 1.2e-8; profiles/r06_horizon.json saw that one as 1.2e-7 on the output)."""
    import re
    mp = pytest.importorskip("mpmath")
    text = open(os.path.join(ROOT, "s-rack_amd", "csrc", "modules.hip.h")).read()
    body = text[text.index("SRK_DEV double exp2_fast9(double x)"):]
    body = body[:body.index("return kReduce")]
    hx = [float.fromhex(h) for h in re.findall(r"0x1\.[0-9a-f]+p[+-]\d+", body)]
    assert len(hx) == 10   # c1, c0, c3, c2, c5, c4, c7, c6, c9, c8 in the order the fmas name them
    c1, c0, c3, c2, c5, c4, c7, c6, c9, c8 = hx

    def series(f):
        f2 = f * f
        a01, a23, a45, a67, a89 = c1 * f + c0, c3 * f + c2, c5 * f + c4, c7 * f + c6, c9 * f + c8
        f4 = f2 * f2
        b0, b1 = a23 * f2 + a01, a67 * f2 + a45
        return (a89 * f4 + b1) * f4 + b0

    f = np.linspace(-0.5, 0.5, 4001)
    mp.mp.dps = 40
    worst = max(abs(mp.mpf(float(pi)) / mp.power(2, mp.mpf(float(fi))) - 1) for pi, fi in zip(series(f), f))
    assert worst < 2e-14, float(worst)
    t = np.arange(2_880_000)
    for index in (0.5, 1.0, 1.5):
        cv = np.sin(2 * np.pi * 0.0123 * t + 0.3).astype(np.float32) * np.float32(index)
        p = series((cv * np.float32(0.25)).astype(np.float64))
        p = p * p
        p = p * p
        drift = float(((440.0 / 48000.0) * (p.astype(np.longdouble) - np.exp2(cv.astype(np.longdouble)))).sum())
        assert abs(drift) < 5e-10, (index, drift)


def test_cycle_rules_of_the_flattener(S):
    """approx.cpp, the cycles: an Add <-> Subtract pair is an integrator — neither its gain nor its VALUES have a bound: the exact flavour; the
    same pair with a ladder in it (whose lowpass is clamped: bounded values, unbounded gain): the saw that feeds it gets the exact PolyBLEP (its
    epsilon times that gain), the ladder the literal form, and as a constant-pitch saw has an exact form of its own the patch stays in the
    default flavour — and an oscillator whose PITCH moves in front of the loop (2^cv by polynomial) is evaluated exactly as a whole, with the
    LFO that moves it; the same saw into no cycle keeps the fast form."""
    import re

    def source(wire_cycle, filter_on_cycle, vibrato=False):
        p = S.Patch(48000, 16, 2)
        osc, add, sub, out = p.add_module(S.MOD_OSCILLATOR), p.add_module(S.MOD_MATH), p.add_module(S.MOD_MATH), p.add_module(S.MOD_OUTPUT)
        p.set_field(add, S.MATH_OPERATION, S.MATH_ADD)
        p.set_field(sub, S.MATH_OPERATION, S.MATH_SUBTRACT)
        p.connect(osc, S.OSC_OUT_SAW, add, 1)
        p.connect(add, 0, sub, 0)
        p.connect(add, 0, out, 0)
        if vibrato:
            lfo = p.add_module(S.MOD_OSCILLATOR)
            p.connect(lfo, S.OSC_OUT_SINE, osc, 0)
        if wire_cycle:
            if filter_on_cycle:
                vcf = p.add_module(S.MOD_MOOG_FILTER)
                p.connect(sub, 0, vcf, 0)
                p.connect(vcf, 0, add, 0)
            else:
                p.connect(sub, 0, add, 0)
        p.configure_voices(8)
        return p.kernel_source(S.RENDER_NO_UNIFORM_HOIST | S.RENDER_NO_FUSION)

    EXACT, EXACT_BLEP = 1 << 6, 1 << 13   # program.hpp: OSC_EXACT, OSC_EXACT_BLEP
    flags_of = lambda src: [int(x, 16) for x in re.findall(r"osc_step\(\(?(0x[0-9a-f]+)u", src)]
    plain = source(False, False)
    assert "fosc_saw" in plain and not flags_of(plain) and "approx[bound" in plain
    fed = source(True, False)
    assert "approx[exact: unbounded values" in fed and "fosc_saw" not in fed
    ladder = source(True, True)
    assert "vcf_run<true>" not in ladder and "fosc_saw" not in ladder and [f & (EXACT | EXACT_BLEP) for f in flags_of(ladder)] == [EXACT_BLEP]
    moving = source(True, True, vibrato=True)
    assert "; exact osc 0,4]" in moving.split("\n", 1)[0] and "fosc_saw" not in moving
    assert all(f & EXACT for f in flags_of(moving)) and len(flags_of(moving)) == 2   # the oscillator and the LFO that moves its pitch: exact as a whole
