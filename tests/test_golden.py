"""The C oracle against the committed golden vectors (tests/golden/, made by make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    audio = z["audio"]
    assert hashlib.sha256(audio.tobytes()).hexdigest() == str(z["sha256"])
    return z, audio


@pytest.mark.parametrize("adsr", ["default", "finite"])
def test_cfg1_p1(W, oracle, adsr):
    z, audio = load(f"cfg1_p1_{adsr}.npz")
    g = oracle.OraclePatch(48000, int(z["buffer_size"]), 2)
    W.build_p1(g, adsr=adsr)
    out = g.render(48000)
    np.testing.assert_array_equal(out[0].view(np.uint32), audio.view(np.uint32))
    np.testing.assert_array_equal(out[1].view(np.uint32), audio.view(np.uint32))
    # the gate opens at sample 13964 (LFO square first > 0): silence before, sound after
    assert not audio[:13964].any() and audio[13964:].any()


@pytest.mark.parametrize("B", [1, 1024])
def test_cfg4_p2(W, oracle, B):
    z, audio = load(f"cfg4_p2_b{B}.npz")
    g = oracle.OraclePatch(48000, B, 2)
    W.build_p2(g, beta=float(z["beta"]), index=float(z["index"]))
    np.testing.assert_array_equal(g.render(len(audio))[0].view(np.uint32), audio.view(np.uint32))


def test_cfg3_voices(W, oracle):
    z, audio = load("cfg3_p1_voices8.npz")
    det, cut = W.p1_voice_params(8)
    np.testing.assert_array_equal(det, z["detune"])
    np.testing.assert_array_equal(cut, z["cutoff"])
    g = oracle.OraclePatch(48000, 1024, 2)
    ids = W.build_p1(g, lfo_val=float(z["lfo_val"]))
    frames, _ = g.render_batch(8, audio.shape[0], [(ids["osc_a"], W.OSC_VAL, det), (ids["vcf"], W.VCF_FREQ, cut)], threads=4)
    np.testing.assert_array_equal(frames[0].view(np.uint32), audio.view(np.uint32))


def test_p3_sequencers(W, oracle):
    z, audio = load("p3_sequencers.npz")
    g = oracle.OraclePatch(48000, int(z["buffer_size"]), 2)
    W.build_p3(g)
    np.testing.assert_array_equal(g.render(audio.shape[0]).T.view(np.uint32), audio.view(np.uint32))
