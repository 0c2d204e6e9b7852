"""tests/cpp/forms_emu.c (CPU only): the oracle with the GPU default mode's cheaper forms in single modules, the instrument of tools/cpu_soak.py.
Pinned here: with no forms it IS the oracle; on the benchmarked patches its error against the oracle is the GPU's own measured error
(profiles/r05_horizon.json, same 64 voices, first second); round 5's five GPU soak finds leave the band with every form taken and stay inside with
csrc/approx.cpp's decisions; a small soak of the bound itself."""
import json
import os
import sys

import numpy as np
import pytest

import srack_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cpu_soak  # noqa: E402
from test_approx import probe  # noqa: E402,F401  (the fixture that builds tests/cpp/approx_probe)


def test_without_forms_it_is_the_oracle(oracle):
    from tests import forms_emu
    S = srack_pkg.load()
    B, build, overrides = S.bench_workload("p3", 8)
    a, b = oracle.OraclePatch(48000, B, 2), forms_emu.EmuPatch(48000, B, 2)
    ids = build(a)
    build(b)
    for m in range(b.num_modules()):
        if b.L.emu_set_forms(b.h, m, 0) < 0:
            raise AssertionError(m)
    ra, _ = a.render_batch(8, 5000, overrides(ids), threads=2)
    rb, _ = b.render_batch(8, 5000, overrides(ids), threads=2)
    assert np.array_equal(ra.view(np.uint32), rb.view(np.uint32)) and np.abs(ra).max() > 0.1


@pytest.mark.parametrize("name", ["cfg3", "cfg3_poly", "cfg4", "p3", "p4"])
def test_emulated_error_is_the_gpus_measured_error(name, probe):
    """Same patch, same 64 voices, same second as tools/horizon.py rendered on the GPU: the emulation's error against the oracle and the
    kernels' agree to a few percent (config 3: 3.576e-7 both) — the forms' errors are what the emulation injects, not a stand-in for them."""
    S = srack_pkg.load()
    rows = json.load(open(os.path.join(ROOT, "profiles", "r05_horizon.json")))["rows"]
    gpu = max(r["max_rel_err_per_second"][0] for r in rows if r["workload"] == name and r["flags"] in (0, 32))
    V, T = 64, 48000
    B, build, overrides = S.bench_workload(name, V)
    g = cpu_soak.Both(48000, B, 2)
    ids = build(g)
    ov = overrides(ids)
    for m, f, vals in ov:
        g.rec.override(m, f, vals)
    plan = g.rec.run(probe)
    assert g.b.apply_plan(g.types, plan)
    ref, _ = g.a.render_batch(V, T, ov, threads=8)
    emu, _ = g.b.render_batch(V, T, ov, threads=8)
    err = float((np.abs(emu.astype(np.float64) - ref) / np.maximum(np.abs(ref), 1.0)).max())
    assert 0.7 * gpu <= err <= 1.3 * gpu and err <= plan["bound"] * 1.01, (name, err, gpu, plan["bound"])


@pytest.mark.parametrize("seed,noise,more,gpu_found", [(66697, False, False, 4.5e-5), (72223, False, False, 5.3e-5), (104123, False, True, 5.2e-4),
                                                       (105055, False, True, 6.9e-5), (123042, True, False, 1.01e-4)])
def test_round_fives_gpu_soak_finds_reproduce(seed, noise, more, gpu_found, probe, monkeypatch):
    """The five patches round 5's GPU soaks found outside the band (200 voices x 6 000 samples): with every form taken the emulation leaves the
    band on each — on 123042, whose decisions then WERE every form, by the GPU's own 1.01e-4 —, with csrc/approx.cpp's decisions it stays
    inside.  (The other four were found under decisions that denied some forms already: the emulation with everything taken errs more.)"""
    if more:
        monkeypatch.setenv("FUZZ_MORE_OV", "1")
    _, e_all, _, masks, _, n, _ = cpu_soak.one((seed, noise, 200, 6000, True))
    assert n > 0 and masks and e_all >= gpu_found * 0.9, (e_all, gpu_found)
    if seed == 123042:
        assert e_all == pytest.approx(gpu_found, rel=0.02)
    _, e_plan, _, masks, _, _, _ = cpu_soak.one((seed, noise, 200, 6000, False))
    assert masks and e_plan <= 1e-5


@pytest.mark.parametrize("noise", [False, True])
def test_a_small_soak_of_the_bound(noise, probe):
    """120 fuzz patches per family, 16 voices x 3 000 samples, the forms the bound chose: inside the contract (tools/cpu_soak.py runs the large ones:
    notes/r05.md R5.8)."""
    worst, rendered = 0.0, 0
    for seed in range(900000, 900120):
        _, e, _, masks, note, n, bound = cpu_soak.one((seed, noise, 16, 3000, False))
        assert masks and e <= 1e-5 and e <= bound + 3.6e-7, (seed, e, note)
        worst, rendered = max(worst, e), rendered + (n > 0)
    assert rendered > 30 and worst > 0.0
