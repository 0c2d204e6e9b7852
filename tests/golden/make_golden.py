"""Generates tests/golden/*.npz — run from the repo root: python tests/golden/make_golden.py

The reference (Rust) cannot be built or imported in this image and ships no golden audio, so the
vectors come from the C oracle (oracle/srack_oracle.c) and are written ONLY if the independent
NumPy restatement (oracle/srack_numpy.py) reproduces them bit for bit.  Each fixture holds the
full rendered channel 0 as float32 plus a sha256 of its bytes.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import srack_pkg  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.npgraph import NumpyGraph  # noqa: E402

W = srack_pkg.load_workloads()
HERE = os.path.dirname(os.path.abspath(__file__))


def both(build, n, B, **kw):
    g = O.OraclePatch(48000, B, 2)
    build(g, **kw)
    ng = NumpyGraph(48000, B, 2)
    build(ng, **kw)
    a, b = g.render(n), ng.render(n)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "C oracle != NumPy restatement"
    assert np.array_equal(a[0], a[1])
    return a[0]


def save(name, audio, **meta):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:  # `make_golden.py p4_sample_nonlinear.npz`: only that file
        return
    sha = hashlib.sha256(audio.tobytes()).hexdigest()
    np.savez_compressed(os.path.join(HERE, name), audio=audio, sha256=np.array(sha), **{k: np.array(v) for k, v in meta.items()})
    print(name, audio.shape, sha[:16], float(np.abs(audio).max()))


if __name__ == "__main__":
    O.build()
    # cfg1: 1 voice, P1, 1 s @ 48 kHz, B=1024 (47 ticks, truncated), both ADSR variants
    save("cfg1_p1_default.npz", both(W.build_p1, 48000, 1024, adsr="default"), buffer_size=1024, adsr="default")
    save("cfg1_p1_finite.npz", both(W.build_p1, 48000, 1024, adsr="finite"), buffer_size=1024, adsr="finite")
    # cfg4 shape: P2 FM with feedback, z^-1 (B=1) and the app-default delay line (B=1024); 0.25 s
    save("cfg4_p2_b1.npz", both(W.build_p2, 12000, 1, beta=0.3, index=1.0), buffer_size=1, beta=0.3, index=1.0)
    save("cfg4_p2_b1024.npz", both(W.build_p2, 12000, 1024, beta=0.3, index=1.0), buffer_size=1024, beta=0.3, index=1.0)
    # scope table (f) rank 1: sequencer-driven patch P3, both channels (audio, raw pattern gate), 0.25 s
    g = O.OraclePatch(48000, 1024, 2)
    W.build_p3(g)
    ng = NumpyGraph(48000, 1024, 2)
    W.build_p3(ng)
    a, b = g.render(12000), ng.render(12000)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "C oracle != NumPy restatement (P3)"
    save("p3_sequencers.npz", np.ascontiguousarray(a.T), buffer_size=1024)
    # scope table (f) rank 4: sample player + waveshaper P4, both channels (shaped, raw sample), 0.25 s
    g = O.OraclePatch(48000, 1024, 2)
    W.build_p4(g)
    ng = NumpyGraph(48000, 1024, 2)
    W.build_p4(ng)
    a, b = g.render(12000), ng.render(12000)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "C oracle != NumPy restatement (P4)"
    save("p4_sample_nonlinear.npz", np.ascontiguousarray(a.T), buffer_size=1024)
    # cfg3 shape: 8 voices of P1 with the per-voice detune/cutoff draw, 0.5 s
    det, cut = W.p1_voice_params(8)
    voices = []
    for v in range(8):
        def build(g, v=v):
            ids = W.build_p1(g, lfo_val=-4.0)
            g.set_field(ids["osc_a"], W.OSC_VAL, det[v])
            g.set_field(ids["vcf"], W.VCF_FREQ, cut[v])
        voices.append(both(build, 24000, 1024))
    save("cfg3_p1_voices8.npz", np.stack(voices, axis=1), detune=det, cutoff=cut, lfo_val=-4.0)
