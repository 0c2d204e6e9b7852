"""Kernels generated for patches with several output planes (default mode): the planes share ONE LDS transpose tile for the mix-down —
plane j owns 32 / P' rows of it (P' = planes rounded up to a power of two) and all rows are summed in one pass every 32 / P' samples
(wave.hip.h: emit_put_rows / emit_rows_flush) — instead of a tile each, which held a CU to two waves per SIMD from two planes on.
Checked here: 2, 3 and 4 planes, a ragged last wave (70 voices), ragged tiles and sub-tiles (T = 1003), frames against the oracle,
the mix against the f64 sum of the frames, and the mix BIT FOR BIT against the tile-per-plane form (SRACK_TILE_PER_PLANE=1: a row's sum
is formed by the same reads and the same additions either way)."""
import os

import numpy as np
import pytest

import srack_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    S = srack_pkg.load()
    assert S.device_count() > 0, "no GPU visible: the render path has no CPU fallback"
    return S


def build(g, n_planes, S):
    """n_planes oscillators (distinct pitches, per-voice detune on the first), the first through a ladder filter; channel c <- plane c."""
    out = None
    oscs = [g.add_module(S.MOD_OSCILLATOR) for _ in range(n_planes)]
    vcf = g.add_module(S.MOD_MOOG_FILTER)
    out = g.add_module(S.MOD_OUTPUT)
    for k, o in enumerate(oscs):
        g.set_field(o, S.OSC_VAL, 0.13 * k - 0.2)
    g.connect(oscs[0], S.OSC_OUT_SAW, vcf, 0)
    g.connect(vcf, 0, out, 0)
    for k in range(1, n_planes):
        g.connect(oscs[k], (S.OSC_OUT_SINE, S.OSC_OUT_SQUARE, S.OSC_OUT_SAW)[k % 3], out, k)
    return oscs


@pytest.mark.parametrize("n_planes", [2, 3, 4])   # (the generator takes up to four distinct planes)
def test_planes_share_one_mix_tile(S, oracle, n_planes):
    V, T = 70, 1003
    flags = S.RENDER_SPECIALIZE
    vals = [np.linspace(-0.3, 0.3, V).astype(np.float32) + np.float32(0.11 * k) for k in range(n_planes)]

    def render():
        p = S.Patch(48000, 256, n_planes)
        oscs = build(p, n_planes, S)
        p.configure_voices(V)
        for o, v in zip(oscs, vals):
            p.set_voice_field(o, S.OSC_VAL, v)
        assert p.planes()[0] == n_planes
        fr, mx = p.render(T, frames=True, mix=True, flags=flags)
        assert "kernel=render_specialized" in p.info(), p.info()
        return p, fr, mx

    os.environ.pop("SRACK_TILE_PER_PLANE", None)
    p, fr, mx = render()
    src = p.kernel_source(flags)
    assert "emit_rows_flush<%d>" % (16 if n_planes == 2 else 8) in src and "mix_tile[1 * kMixTile]" in src
    os.environ["SRACK_TILE_PER_PLANE"] = "1"
    try:
        p1, fr1, mx1 = render()
        assert "emit_rows_flush" not in p1.kernel_source(flags)
    finally:
        del os.environ["SRACK_TILE_PER_PLANE"]
    np.testing.assert_array_equal(fr.view(np.uint32), fr1.view(np.uint32))
    np.testing.assert_array_equal(mx.view(np.uint32), mx1.view(np.uint32))   # bit for bit: the same sums
    o = oracle.OraclePatch(48000, 256, n_planes)
    oscs = build(o, n_planes, S)
    ref, _ = o.render_batch(V, T, [(m, S.OSC_VAL, v) for m, v in zip(oscs, vals)], threads=4)
    for c in range(n_planes):
        err = np.abs(fr[c].astype(np.float64) - ref[c]) / np.maximum(np.abs(ref[c]), 1.0)
        assert err.max() <= 1e-5, (c, err.max())
        own = fr[c].astype(np.float64).sum(axis=1)
        scale = np.abs(fr[c].astype(np.float64)).sum(axis=1)
        assert (np.abs(mx[c] - own) <= 1e-5 * np.maximum(scale, 1.0)).all(), c
        assert np.abs(mx[c]).max() > 1.0
