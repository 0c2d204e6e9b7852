"""Tick sessions (render.hip, TickSession): a host that calls srack_render once per block, as the reference's audio callback calls
`execute(&plan)` once per buffer_size frames (src/main.rs:59-63), gets the control program computed AHEAD across calls.  The contract:
a session changes no bit — of any render, of any state read back between ticks — whatever the host does between ticks (another
length, one long call, frames only / mix only, a state read-back, a parameter edit under keep_state).

Each scenario of tests/tick_driver.py runs in two processes, SRACK_TICK=2 (the default) and SRACK_TICK=0 (every call starts the
control program afresh: the library as it was before sessions existed), and the two .npz files must agree bit for bit; the renders
are also held to the CPU oracle at the 1e-5 bar, so that "both wrong alike" does not pass.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import srack_pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tick_driver  # noqa: E402


def run(scenario, flags, tick, tmp_path):
    out = os.path.join(tmp_path, f"{scenario}_{flags}_{tick}.npz")
    env = dict(os.environ, SRACK_TICK=str(tick))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tick_driver.py"), scenario, str(flags), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32 if a.dtype == np.float32 else np.uint64)


# flags: 0 default, 1 exact oscillators; 32 = SRACK_RENDER_SPECIALIZE (the generated kernel with the control units as its first blocks:
# by default it is reserved for 4096 voices and more)
@pytest.mark.parametrize("scenario,flags", [("p1", 0), ("p1", 1), ("p1", 32), ("p1", 33), ("p1_wide", 0), ("identical", 32), ("identical", 33),
                                            ("p3", 32), ("p3", 33), ("keep", 32), ("keep", 33),
                                            ("p1_multi", 0), ("p1_multi", 1), ("p1_multi", 32), ("identical_multi", 32), ("identical_multi", 33),
                                            ("p3_multi", 32), ("p3_multi", 33), ("keep_multi", 32)])
def test_a_tick_session_changes_no_bit(oracle, tmp_path, scenario, flags):
    S = srack_pkg.load()
    on, off = run(scenario, flags, 2, tmp_path), run(scenario, flags, 0, tmp_path)   # SRACK_TICK=2: sessions of one or several chunks per call (the default)
    assert sorted(on.files) == sorted(off.files)
    for k in on.files:
        if k == "infos":
            continue
        assert on[k].shape == off[k].shape, k
        np.testing.assert_array_equal(bits(on[k]), bits(off[k]), err_msg=k)
    # the kernels the scenario is about: the control program rides on the voice launches (the flagship's block 0 / the units of a
    # specialised kernel) — otherwise there is no session to test
    info = str(on["infos"][0])
    assert ("kernel=render_voice_chain_track" in info) or ("kernel=render_specialized" in info and "tracks=" in info), info
    # ... and against the oracle: one render of the script's whole length (up to the edit, where there is one), cut at the calls
    V = tick_driver.voices_of(scenario)
    script = tick_driver.SCRIPTS[scenario]
    upto = next((i for i, op in enumerate(script) if op[0] == "edit"), len(script))
    total = sum(op[1] for op in script[:upto] if op[0] == "render")
    o = oracle.OraclePatch(48000, 1024, 2)
    ids, over = tick_driver.make(S, o, scenario, V)
    ref, _ = o.render_batch(V, total, over, threads=8)
    p = S.Patch(48000, 1024, 2)
    tick_driver.make(S, p, scenario, V)
    p.configure_voices(V)
    _, planes = p.planes()
    n = checked = 0
    for step, op in enumerate(script[:upto]):
        if op[0] != "render":
            continue
        if f"fr{step}" in on.files:
            got = on[f"fr{step}"]
            for c, pl in enumerate(planes):
                if pl >= 0:
                    want = ref[c][n:n + op[1]]
                    err = np.abs(got[pl].astype(np.float64) - want) / np.maximum(np.abs(want), 1.0)
                    assert err.max() <= 1e-5, (step, c, err.max())
                    checked += 1
        n += op[1]
    assert checked > 0 and np.abs(ref).max() > 0.05


def test_two_patches_on_two_streams(tmp_path):
    """Two patches ticking on two non-default streams, their calls interleaved and unsynchronised (each with a session of its own),
    one of them changing stream mid-way: bit for bit what each renders alone on the null stream."""
    out = os.path.join(tmp_path, "streams.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_driver.py"), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    for k in range(2):
        assert np.abs(d[f"alone_fr{k}"]).max() > 0.05
        np.testing.assert_array_equal(bits(d[f"both_fr{k}"]), bits(d[f"alone_fr{k}"]), err_msg=f"frames of patch {k}")
        np.testing.assert_array_equal(bits(d[f"both_mx{k}"]), bits(d[f"alone_mx{k}"]), err_msg=f"mix of patch {k}")
    assert "kernel=render_voice_chain_track" in str(d["info0"]) and "kernel=render_specialized" in str(d["info1"])
    # ... and a session whose stream the host destroyed: carried on after a state read-back / on another stream, still the same bits
    for tag in ("gone_read", "gone_move"):
        np.testing.assert_array_equal(bits(d[tag + "_fr"]), bits(d["alone_fr0"]), err_msg=tag)
        np.testing.assert_array_equal(bits(d[tag + "_mx"]), bits(d["alone_mx0"]), err_msg=tag)
    assert ((d["gone_read_pos"] >= 0) & (d["gone_read_pos"] < 1)).all()
