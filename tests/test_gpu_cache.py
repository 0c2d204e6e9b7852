"""The cache of run-time specialised kernels on a real device (-m gpu): bounded in memory while a host keeps re-patching, persistent on
disk across process starts (jit.cpp "the kernel cache"; include/srack_hip.h srack_kernel_cache_*).  The CPU suite covers the compile
and disk levels (tests/test_host.py); here modules are really loaded, launched and unloaded."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_MANY = r"""
import json, os, sys
sys.path.insert(0, os.environ["SRACK_ROOT"])
import numpy as np
import srack_pkg
from oracle import oracle as O
S = srack_pkg.load()
O.build()
V, T, N = 64, 96, int(os.environ["SRACK_N_STRUCTURES"])
val = np.linspace(-1, 1, V).astype(np.float32)
def chain(g, ops):   # oscillator saw -> one Math module per digit (add / subtract / multiply by a constant) -> output
    o = g.add_module(S.MOD_OSCILLATOR)
    prev = o
    for j, k in enumerate(ops):
        m = g.add_module(S.MOD_MATH)
        g.set_field(m, S.MATH_OPERATION, k)
        g.set_field(m, S.MATH_CONSTANT, 0.25 + 0.125 * j)
        g.connect(prev, 2 if prev == o else 0, m, 0)
        prev = m
    out = g.add_module(S.MOD_OUTPUT)
    g.connect(prev, 2 if prev == o else 0, out, 0)
    return o
notes, checked = set(), 0
for i in range(N):
    ops = [(i // 3 ** j) % 3 for j in range(5)]   # 243 distinct op sequences = 243 program structures
    p = S.Patch(48000, 64, 2)
    o = chain(p, ops)
    p.configure_voices(V)
    p.set_voice_field(o, S.OSC_VAL, val)
    fr = p.render_channels(T, S.RENDER_NO_FUSION | S.RENDER_SPECIALIZE)
    info = p.info()
    assert "kernel=render_specialized" in info, info
    notes.add(info.split("jit=")[1].split("(")[0].split(" ")[0])
    if i % 25 == 0 or i == N - 1:
        g = O.OraclePatch(48000, 64, 2)
        og = chain(g, ops)
        ref, _ = g.render_batch(V, T, [(og, S.OSC_VAL, val)])
        assert np.abs(fr.astype(np.float64) - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), (i, ops)
        checked += 1
    del p
st = S.kernel_cache_stats()
st["notes"], st["checked"] = sorted(notes), checked
print(json.dumps(st))
"""

_P3 = r"""
import json, os, sys
sys.path.insert(0, os.environ["SRACK_ROOT"])
import numpy as np
import srack_pkg
S = srack_pkg.load()
V = 4096   # from here up the general path specialises by default
p = S.Patch(48000, 1024, 2)
ids = S.build_p3(p)
p.configure_voices(V)
p.set_voice_field(ids["transpose"], S.MATH_CONSTANT, np.linspace(-2, 0.5, V).astype(np.float32))
p.set_voice_field(ids["vcf"], S.VCF_FREQ, np.linspace(0.05, 0.4, V).astype(np.float32))
fr, mix = p.render(2048)
st = S.kernel_cache_stats()
st["info"] = p.info()
st["checksum"] = float(np.abs(fr).sum())
print(json.dumps(st))
"""


def run(script, cache_dir, **env):
    e = dict(os.environ, SRACK_ROOT=ROOT, SRACK_KERNEL_CACHE_DIR=str(cache_dir), **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", script], env=e, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_hundred_program_structures_do_not_accumulate():
    """A host that keeps re-patching: 200 distinct structures rendered one after the other with room for 16 — every one is compiled,
    loaded, launched (a sample of them checked against the oracle) and, once sixteen newer ones have come, forgotten and unloaded."""
    st = run(_MANY, "off", SRACK_KERNEL_CACHE_MAX=16, SRACK_N_STRUCTURES=200)
    assert st["compiled"] == 200 and st["modules_loaded"] == 200 and st["notes"] == ["compiled"] and st["checked"] == 9
    assert st["resident_modules"] == 16 and st["module_evictions"] == 184
    assert st["resident_code_objects"] == 16 and st["code_evictions"] == 184


def test_second_start_of_a_host_compiles_nothing(tmp_path):
    """P3 — five control units and a voice program, ~1.5 s of hiprtc — on the first start; from the disk cache on the second."""
    first = run(_P3, tmp_path)
    assert first["compiled"] == 1 and "jit=compiled(" in first["info"] and first["info"].endswith("kernel=render_specialized")
    second = run(_P3, tmp_path)
    assert second["compiled"] == 0 and second["disk_hits"] == 1 and "jit=disk-cache" in second["info"]
    assert second["checksum"] == first["checksum"] and first["checksum"] > 0  # the same kernel, to the bit


_BASELINE = r"""
import json, os, sys
if os.environ.get("SRACK_IMPORT_TORCH_FIRST") == "1":
    import torch   # bench.py's order: the process then runs on the HIP runtime (and hiprtc) of torch's wheel, not the image's
sys.path.insert(0, os.environ["SRACK_ROOT"])
import numpy as np
import srack_pkg
S = srack_pkg.load()
V, out = 4096, {}
def poly(ids):
    return S.p1_poly_overrides(ids, S.p1_poly_voice_params(V))
def fm(ids):
    beta, index = S.p2_voice_params(V)
    return [(ids["mul_fb"], S.MATH_CONSTANT, beta), (ids["mul_idx"], S.MATH_CONSTANT, index)]
for name, B, build_patch, per_voice, flags in (("cfg2", 1024, S.build_p1, lambda ids: [], 0), ("cfg3_poly", 1024, S.build_p1, poly, 0),
                                              ("cfg4_fast", 1, S.build_p2, fm, S.RENDER_KEEP_DEFAULT)):
    p = S.Patch(48000, B, 2)
    ids = build_patch(p)
    p.configure_voices(V)
    for m, f, v in per_voice(ids):
        p.set_voice_field(m, f, v)
    fr, mix = p.render(1024, flags=flags)
    out[name] = p.info()
st = S.kernel_cache_stats()
st["info"] = out
print(json.dumps(st))
"""


@pytest.mark.parametrize("torch_first", [False, True])
def test_kernels_of_the_baseline_configurations_come_prebuilt(torch_first):
    """`__graft_entry__.build()` pre-compiles the kernels the BASELINE configurations' renders use (config 2, cfg3_poly, config 4's fast kernels) into
    the in-tree disk cache that ships with the library — once per HIP runtime a process may run on: the image's, and the one torch's
    wheel brings when torch is imported first (bench.py's order; round 3's driver line said `jit=compiled` because only the former had
    been built).  A FRESH process of either kind renders them without compiling anything."""
    e = dict(os.environ, SRACK_ROOT=ROOT, SRACK_IMPORT_TORCH_FIRST="1" if torch_first else "0")
    e.pop("SRACK_KERNEL_CACHE_DIR", None)   # the default resolution: next to the library
    r = subprocess.run([sys.executable, "-c", _BASELINE], env=e, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-4000:]
    st = json.loads(r.stdout.strip().splitlines()[-1])
    assert st["directory"].endswith(".srack_kernel_cache"), st["directory"]
    for name, info in st["info"].items():
        assert "kernel=render_specialized" in info and "jit=disk-cache" in info, (name, info)
    assert st["compiled"] == 0, st
