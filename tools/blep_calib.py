"""kEpsBlep (csrc/approx.cpp), measured: the f32 PolyBLEP forms of the kernels — restated in tests/cpp/forms_emu.c, which tools/emu_vs_gpu.py holds
to the kernels bit for bit — against the reference's f64 PolyBLEP on a raw oscillator port, max |a - b| / max(|b|, 1) over many pitches.
  carried   one port read, constant pitch: cosc_saw / cosc_square (and, up to one conversion, the flagship's fosc_saw)
  stepwise  both ports read (or a pitch CV): osc_step's poly_blep_sel
usage: blep_calib.py [voices] [samples]     (CPU only)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import srack_pkg
from oracle import oracle
from tests import forms_emu

W = srack_pkg.load_workloads()
V = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
T = int(sys.argv[2]) if len(sys.argv) > 2 else 48000
rng = np.random.default_rng(6)
vals = rng.uniform(-9.0, 4.7, V).astype(np.float32)     # 0.86 Hz ... 11.4 kHz: every increment below 1/4
pos0 = rng.uniform(0.0, 1.0, V)
worst = {}
for name, ports, word in (("carried saw", [2], 1 | 8), ("carried square", [1], 1 | 8), ("stepwise saw + square", [2, 1], 1), ("fixed-point saw", [2], 1 | 8 | 16)):
    outs = []
    for emu in (False, True):
        g = forms_emu.EmuPatch(48000, 1024, 2) if emu else oracle.OraclePatch(48000, 1024, 2)
        osc = g.add_module(W.MOD_OSCILLATOR)
        out = g.add_module(W.MOD_OUTPUT)
        for c, p in enumerate(ports):
            g.connect(osc, p, out, c)
        if emu:
            g.set_forms(osc, word)
        fr, _ = g.render_batch(V, T, [(osc, W.OSC_VAL, vals), (osc, W.OSC_POS, pos0)], threads=8)
        outs.append(fr.astype(np.float64))
    err = np.abs(outs[1] - outs[0]) / np.maximum(np.abs(outs[0]), 1.0)
    per_voice = err.max(axis=(0, 1)) if err.ndim == 3 else err.max(axis=0)
    k = int(np.argmax(per_voice))
    if name == "fixed-point saw":   # its window term is 2^-31 / dt: report the error in that unit, and what is left above it
        dt = 440.0 * 2.0 ** vals.astype(np.float64) / 48000.0
        pv = err.max(axis=(0, 1)) if err.ndim == 3 else err.max(axis=0)
        print(f"{name:24s} max over voices of error / (2^-31 / dt): {float((pv / (2.0 ** -31 / dt)).max()):.3f};  max of error - 2^-31 / dt: {float((pv - 2.0 ** -31 / dt).max()):.3e}")
        continue
    worst[name] = float(err.max())
    print(f"{name:24s} max {err.max():.3e}  (voice {k}: val {vals[k]:+.3f} = {440 * 2.0 ** float(vals[k]):.1f} Hz)  share of samples that differ {float((err > 0).mean()):.4f}  |ref| max {np.abs(outs[0]).max():.3f}")
print("kEpsBlep must cover", max(worst.values()))
