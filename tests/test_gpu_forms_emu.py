"""The CPU emulation of the default mode's forms (tests/cpp/forms_emu.c) against the kernels themselves.

tools/cpu_soak.py's evidence for the error bound — hundreds of thousands of fuzz patches rendered by the oracle with csrc/approx.cpp's chosen
forms put into its modules — is evidence about forms_emu.c; it is evidence about the KERNELS only while the two agree.  Round 6 found they had not:
the kernels render a constant-pitch saw / square through carried-phase forms (cosc_saw: the second PolyBLEP window as (next phase / dt)^2) and a
voice program's saw with its phase in 2^-64 fixed point (fosc_saw: t = pos / dt from the phase's upper 32 bits), the emulation restated osc_step
only — 1.0e-6 on the GPU where the emulation (and the bound) said 2.4e-7.  Both forms are restated now, the bound's epsilon is measured on them
(tools/blep_calib.py), and this test holds the two together: the same fuzz patches, the same decisions, rendered by the GPU's default modes and by
the emulation — bit for bit on nearly every patch, never further apart than one f32 ulp (what is NOT restated: the fixed-point accumulator's own
2^-64 steps — the emulation takes the phase's upper word from the oracle's f64 phase — and the polynomial 2^cv of oscillators whose pitch moves,
3e-16 against the libm's pow), and every GPU render within its own bound."""
import os
import sys

import pytest

import srack_pkg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from tests.test_approx import probe  # noqa: E402,F401  (the fixture that builds tests/cpp/approx_probe)


@pytest.mark.parametrize("noise,first,last", [(False, 900000, 900060), (True, 900000, 900030)])
def test_the_emulation_is_the_kernels(noise, first, last, probe):
    import emu_vs_gpu
    from tests import forms_emu
    S = srack_pkg.load()
    forms_emu.lib()
    rows = []
    for seed in range(first, last):
        rows += emu_vs_gpu.compare(seed, noise, 64, 3000, [0, 4], S, probe) or []
    assert len(rows) >= (40 if not noise else 16), len(rows)       # (about a third of the fuzzer's patches take a form at all)
    for r in rows:
        where = f"seed {r['seed']} noise {noise} flags {r['flags']}: {r['info']}"
        assert r["gpu_err"] <= 1e-5 and r["gpu_err"] <= r["bound"] * 1.05 + 3.6e-7, (where, r["gpu_err"], r["bound"])   # the bound's own claim (+ 3 f32 ulps)
        assert r["gpu_minus_emu"] <= 1.2e-7 and r["bit_equal"] >= 0.99, (where, r["gpu_minus_emu"], r["bit_equal"])
    assert sum(r["bit_equal"] == 1.0 for r in rows) >= 0.9 * len(rows)
