"""The CPU emulation of the default mode's forms (tests/cpp/forms_emu.c) against the kernels themselves, patch by patch (GPU box).
For each fuzz patch: the oracle, the emulation with csrc/approx.cpp's decisions (tests/cpp/approx_probe), the GPU's default modes — and how the
last two differ: share of samples equal to the bit, max |gpu - emu| / max(|ref|, 1), each one's error against the oracle, the patch's bound.
usage: emu_vs_gpu.py <first> <last> [noise] [--vt V,T] [--flags 0,2]      environment: FUZZ_MORE_OV / FUZZ_NONLIN as for tools/cpu_soak.py"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def compare(seed, noise, V, T, flags_list, S, probe):
    import cpu_soak
    from tests.fuzz_patches import random_patch
    B, build, overrides = random_patch(seed, noise)
    g = cpu_soak.Both(48000, B, 2)
    ids = build(g)
    ov = [(ids[m], f, fn(V)) for m, f, fn in overrides]
    for m, f, vals in ov:
        g.rec.override(m, f, vals)
    plan = g.rec.run(probe)
    if plan["exact_patch"]:
        return None
    forms = g.b.apply_plan(g.types, plan, per_voice={m for m, _, _ in ov})
    if not forms:
        return None
    ref, _ = g.a.render_batch(V, T, ov, threads=8)
    emu, _ = g.b.render_batch(V, T, ov, threads=8)
    r64 = ref.astype(np.float64)
    den = np.maximum(np.abs(r64), 1.0)
    fin = np.isfinite(r64)
    rows = []
    for flags in flags_list:
        p = S.Patch(48000, B, 2)
        build(p)
        p.configure_voices(V)
        for m, f, vals in ov:
            p.set_voice_field(m, f, vals)
        if flags & 32:
            try:
                p.kernel_source(flags)
            except S.SrackError:
                continue
        fr = p.render_channels(T, flags)
        ok = fin & np.isfinite(fr) & np.isfinite(emu)
        same = float(((fr.view(np.uint32) == emu.view(np.uint32)) | ~ok).mean())
        d_ge = float(np.where(ok, np.abs(fr.astype(np.float64) - emu) / den, 0.0).max())
        e_g = float(np.where(ok, np.abs(fr.astype(np.float64) - r64) / den, 0.0).max())
        e_e = float(np.where(ok, np.abs(emu.astype(np.float64) - r64) / den, 0.0).max())
        rows.append(dict(seed=seed, noise=noise, flags=flags, forms=len(forms), bit_equal=same, gpu_minus_emu=d_ge, gpu_err=e_g, emu_err=e_e, bound=plan["bound"], info=p.info()))
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int); ap.add_argument("last", type=int); ap.add_argument("noise", nargs="?")
    ap.add_argument("--vt", default="64,6000"); ap.add_argument("--flags", default="0,2"); ap.add_argument("--json")
    a = ap.parse_args()
    V, T = (int(x) for x in a.vt.split(","))
    import srack_pkg
    from tests import forms_emu
    from tests.test_approx import _build, CSRC
    S = srack_pkg.load()
    forms_emu.lib()
    probe = _build(os.path.join(ROOT, "tests", "cpp", "approx_probe"),
                   [os.path.join(ROOT, "tests", "cpp", "approx_probe.cpp"), os.path.join(CSRC, "graph.cpp"), os.path.join(CSRC, "approx.cpp"), os.path.join(CSRC, "approx.hpp"),
                    os.path.join(CSRC, "graph.hpp"), os.path.join(CSRC, "flatten.hpp"), os.path.join(ROOT, "include", "srack_hip.h")], ["-std=c++17", "-Wall"])
    out = []
    for seed in range(a.first, a.last):
        rows = compare(seed, bool(a.noise), V, T, [int(x) for x in a.flags.split(",")], S, probe)
        for r in rows or []:
            out.append(r)
            print("seed %d flags %2d forms %d: bit-equal %.5f  |gpu-emu| %.2e  gpu err %.2e  emu err %.2e  bound %.1e" % (r["seed"], r["flags"], r["forms"], r["bit_equal"], r["gpu_minus_emu"], r["gpu_err"], r["emu_err"], r["bound"])
                  + ("   " + r["info"] if os.environ.get("EMU_INFO") or r["gpu_err"] > r["bound"] * 1.05 + 3.6e-7 else ""), flush=True)
    if a.json:
        os.makedirs(os.path.dirname(os.path.abspath(a.json)), exist_ok=True)
        json.dump(out, open(a.json, "w"), indent=1)
